"""Builds the repository's own .hip sources -- capi.hip and the kernels that have not met a GPU yet -- with g++
against tests/native/hipshim (a minimal HIP-on-the-CPU: fibers for the threads of a block, barriers, wave shuffles)
and exposes the resulting library through ctypes with the C ABI's signatures.  Unlike tests/native_host.py (which
re-implements the launch loops around the shared *_math.h arithmetic) this runs the REAL kernels, launchers and
C-ABI entry points: grid / block geometry, index arithmetic, LDS reductions, last-block finalisation, workspace
handling.  Test infrastructure only."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "monodetr_amd", "csrc")
SHIM = os.path.join(HERE, "native", "hipshim")
OUT = os.path.join(HERE, "native", "_build", "libemul.so")
KERNELS = ["capi", "pair_losses", "ddn_loss", "adamw", "msda_prologue", "kitti_prep", "colsum", "token_gemm", "msda", "msda_tiled", "lsa", "rotate_iou", "kitti_stats", "add_ln"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        srcs = [os.path.join(CSRC, k + ".hip") for k in KERNELS]
        extra = [os.path.join(SHIM, "runtime.cpp"), os.path.join(SHIM, "stubs.cpp")]
        deps = srcs + extra + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
            [os.path.join(SHIM, "hip", f) for f in os.listdir(os.path.join(SHIM, "hip"))] + [os.path.join(SHIM, "mdetr_wave.h")] + [os.path.join(ROOT, "include", "monodetr_amd.h")]
        if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
            cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-I", SHIM, "-I", os.path.join(ROOT, "include"),
                   "-I", CSRC, "-o", OUT]
            for s in srcs:
                cmd += ["-x", "c++", s]
            cmd += ["-x", "none"] + extra
            subprocess.check_call(cmd)
        from monodetr_amd import _capi
        L = ctypes.CDLL(OUT)
        for name, (res, args) in _capi.SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib
