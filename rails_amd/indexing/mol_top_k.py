from ..topk_modules import (MoLAvgTopK, MoLBruteForceTopK, MoLCombTopK, MoLNaiveTopK,  # reference: rails/indexing/mol_top_k.py
                            MoLTopKModule)

__all__ = ["MoLTopKModule", "MoLBruteForceTopK", "MoLNaiveTopK", "MoLAvgTopK", "MoLCombTopK"]
