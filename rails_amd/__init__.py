"""rails_amd: MI355X-native Mixture-of-Logits retrieval behind the rails.similarities / rails.indexing API.

Importing the package does not load the HIP library; the first kernel call does (rails_amd._lib.load()),
and fails loudly if librails_amd.so has not been built.
"""
from .factory import create_mol_interaction_module
from .hstu import HSTU
from .mol_module import DotProductSimilarity, GeGLU, MoLSimilarity, SimilarityModule, SwiGLU
from .topk_modules import (CandidateIndex, MIPSBruteForceTopK, MoLAvgTopK, MoLBruteForceTopK, MoLCombTopK, MoLNaiveTopK,
                           TopKModule, get_top_k_module)

__all__ = [
    "create_mol_interaction_module", "MoLSimilarity", "DotProductSimilarity", "SimilarityModule", "CandidateIndex", "MIPSBruteForceTopK",
    "MoLBruteForceTopK", "MoLAvgTopK", "MoLNaiveTopK", "MoLCombTopK", "TopKModule", "get_top_k_module", "HSTU", "GeGLU", "SwiGLU",
]
