from ..mol_module import SimilarityModule  # reference: rails/similarities/module.py

__all__ = ["SimilarityModule"]
