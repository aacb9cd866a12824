from ...mol_module import RecoMoLItemEmbeddingsFn  # reference: rails/similarities/mol/item_embeddings_fns.py

__all__ = ["RecoMoLItemEmbeddingsFn"]
