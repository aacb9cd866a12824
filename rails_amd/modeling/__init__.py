"""Mirror of the reference package `modeling` (MoL factory only)."""
