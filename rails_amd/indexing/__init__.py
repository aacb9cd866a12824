"""Mirror of the reference package `rails.indexing` (+ top-level `indexing`)."""
