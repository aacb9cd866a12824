# reference: rails/indexing/candidate_index.py (TopKModule) and indexing/candidate_index.py (CandidateIndex)
from ..topk_modules import CandidateIndex, TopKModule

__all__ = ["TopKModule", "CandidateIndex"]
