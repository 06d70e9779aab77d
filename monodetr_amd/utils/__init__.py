"""Mirror of the reference's top-level ``utils`` package (only what the model path uses)."""
