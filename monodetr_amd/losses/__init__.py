"""Mirror of lib/losses (only the function the model path calls)."""
