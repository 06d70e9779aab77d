"""Builds tests/native/host_kernels.cpp with g++ (once per session) and exposes it through ctypes with
the C ABI's own signatures -- a CPU stand-in for the kernel ARITHMETIC, used only by tests."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "host_kernels.cpp")
OUT = os.path.join(HERE, "native", "_build", "libhost_kernels.so")

_lib = None


def lib():
    global _lib
    if _lib is None:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        deps = [SRC] + [os.path.join(HERE, "..", "monodetr_amd", "csrc", f) for f in os.listdir(os.path.join(HERE, "..", "monodetr_amd", "csrc")) if f.endswith("_math.h")]
        if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-o", OUT])
        from monodetr_amd import _capi
        L = ctypes.CDLL(OUT)
        for name in ("mdetr_adamw_step", "mdetr_pair_losses_workspace_bytes", "mdetr_pair_losses_forward",
                     "mdetr_pair_losses_backward", "mdetr_ddn_loss_forward", "mdetr_ddn_loss_backward", "mdetr_lsa_forward_fused", "mdetr_msda_prologue_forward",
                     "mdetr_msda_prologue_backward", "mdetr_kitti_preprocess"):
            res, args = _capi.SIGNATURES[name]
            getattr(L, name).restype, getattr(L, name).argtypes = res, args
        _lib = L
    return _lib
