"""Mirror of lib/models/monodetr (the model package of the reference)."""
