"""Import-path mirror of the reference's modeling/sequential/output_postprocessors.py."""
from rails_amd.hstu import L2NormEmbeddingPostprocessor, LayerNormEmbeddingPostprocessor  # noqa: F401
