"""Import-path mirror of the reference's modeling/sequential/embedding_modules.py."""
from rails_amd.hstu import LocalEmbeddingModule  # noqa: F401
