from ..topk_modules import MoLAvgTopK, MoLBruteForceTopK, MoLTopKModule  # reference: rails/indexing/mol_top_k.py

__all__ = ["MoLTopKModule", "MoLBruteForceTopK", "MoLAvgTopK"]
