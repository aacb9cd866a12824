from ...mol_module import RecoMoLQueryEmbeddingsFn  # reference: rails/similarities/mol/query_embeddings_fns.py

__all__ = ["RecoMoLQueryEmbeddingsFn"]
