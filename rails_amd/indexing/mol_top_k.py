from ..topk_modules import MoLBruteForceTopK, MoLTopKModule  # reference: rails/indexing/mol_top_k.py

__all__ = ["MoLTopKModule", "MoLBruteForceTopK"]
