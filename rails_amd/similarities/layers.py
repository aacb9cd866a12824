from ..mol_module import GeGLU, SwiGLU  # reference: rails/similarities/layers.py

__all__ = ["GeGLU", "SwiGLU"]
