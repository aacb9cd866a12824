from ..mol_module import DotProductSimilarity  # reference: rails/similarities/dot_product_similarity_fn.py

__all__ = ["DotProductSimilarity"]
