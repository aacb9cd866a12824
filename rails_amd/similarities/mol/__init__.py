"""Mirror of the reference package `rails.similarities.mol`."""
