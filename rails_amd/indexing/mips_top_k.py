from ..topk_modules import MIPSBruteForceTopK, MIPSTopKModule  # reference: rails/indexing/mips_top_k.py

__all__ = ["MIPSTopKModule", "MIPSBruteForceTopK"]
