"""Host side of the input pipeline -- mirror of ``lib/datasets``.  File decoding,
label / calibration parsing, the random augmentation decisions and the target encoding stay on CPU workers; every
per-pixel operation of the reference's ``__getitem__`` runs on the device (``monodetr_amd/kitti_prep_ext.py``)."""
