from ..topk_modules import get_top_k_module  # reference: indexing/utils_rails.py

__all__ = ["get_top_k_module"]
