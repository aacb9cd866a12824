from ...mol_module import MoLEmbeddingsFn  # reference: rails/similarities/mol/embeddings_fn.py

__all__ = ["MoLEmbeddingsFn"]
