"""Mirror of lib/models/monodetr/ops: the multi-scale deformable attention operator package."""
