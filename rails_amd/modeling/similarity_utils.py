from ..factory import create_mol_interaction_module, init_mlp_xavier_weights_zero_bias  # reference: modeling/similarity_utils.py

__all__ = ["create_mol_interaction_module", "init_mlp_xavier_weights_zero_bias"]
