from ...mol_module import MoLGatingFn, MoLSimilarity, SoftmaxDropoutCombiner  # reference: rails/similarities/mol/similarity_fn.py

__all__ = ["MoLSimilarity", "MoLGatingFn", "SoftmaxDropoutCombiner"]
