"""Import-path mirror of the reference's modeling/sequential/input_features_preprocessors.py."""
from rails_amd.hstu import LearnablePositionalEmbeddingInputFeaturesPreprocessor  # noqa: F401
