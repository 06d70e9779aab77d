"""Mirror of the pieces of lib/helpers the hot path's caller needs (optimizer)."""
