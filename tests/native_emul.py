"""Builds the repository's own .hip sources -- capi.hip and every kernel file -- with g++
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
OUT = os.path.join(HERE, "native", "_build", "libemul_asan.so" if os.environ.get("MDETR_EMUL_ASAN") == "1" else "libemul.so")
KERNELS = ["capi", "bias_act", "decimate", "conv3x3", "conv_taps", "conv_stem", "conv_wgrad", "pair_losses", "ddn_loss", "adamw", "msda_prologue", "kitti_prep", "colsum", "group_norm", "small_wgrad", "tgemm", "sgemm", "head_tail", "twgrad", "wfold", "msda", "msda_tiled", "msda_fused", "msda_cpu", "lsa", "rotate_iou", "kitti_stats", "add_ln", "attn"]

_libs = {}


def lib(defines=()):
    """The emulated library; ``defines`` (e.g. ("MDETR_ATTN_STAGE_REMAP=1",)) selects a compile-time variant, built
    into its own file."""
    key = tuple(defines)
    if key not in _libs:
        out = OUT if not key else OUT.replace(".so", "_" + "_".join(d.replace("=", "-") for d in key) + ".so")
        _libs[key] = _build(out, key)
    return _libs[key]


SANITIZE = os.environ.get("MDETR_EMUL_ASAN") == "1"       # AddressSanitizer build: run the interpreter with LD_PRELOAD=libasan.so
_SAN = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g"] if SANITIZE else []


def _compile(job):
    src, obj, defines = job
    tmp = "%s.%d.tmp" % (obj, os.getpid())
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-ffp-contract=off"] + _SAN + [ "-I", SHIM, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-c", "-o", tmp] + ["-D" + d for d in defines] + (["-x", "c++"] if src.endswith(".hip") else []) + [src]
    subprocess.check_call(cmd)
    os.replace(tmp, obj)                                   # (a reader never sees a half-written object)


def _build(OUT, defines):
    """One object per source (compiled in parallel, cached by modification time), then one link.  A compile-time
    variant recompiles only the sources that mention one of its macros."""
    from concurrent.futures import ThreadPoolExecutor
    import fcntl
    bdir = os.path.dirname(OUT)
    os.makedirs(bdir, exist_ok=True)
    # several test processes (pytest -n) may find the library stale at the same moment: one builds, the others wait for the lock
    # and then find everything up to date ("file too short" from a half-written libemul.so otherwise)
    lock = open(os.path.join(bdir, ".build.lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        return _build_locked(OUT, defines, bdir)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(OUT, defines, bdir):
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, k + ".hip") for k in KERNELS] + [os.path.join(SHIM, "runtime.cpp")]
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(SHIM, "hip", f) for f in os.listdir(os.path.join(SHIM, "hip"))] + [os.path.join(SHIM, "mdetr_wave.h"),
                                                                                         os.path.join(ROOT, "include", "monodetr_amd.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    macros = [d.split("=")[0] for d in defines]
    jobs, objs = [], []
    for src in srcs:
        affected = [d for d, m in zip(defines, macros) if m in open(src).read()]
        tag = "".join("_" + d.replace("=", "-") for d in affected)
        obj = os.path.join(bdir, ("asan_" if SANITIZE else "emul_") + os.path.basename(src).replace(".", "_") + tag + ".o")
        objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append((src, obj, affected))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as pool:
            list(pool.map(_compile, jobs))
    if jobs or not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(o) for o in objs):
        tmp = "%s.%d.tmp" % (OUT, os.getpid())
        subprocess.check_call(["g++", "-shared", "-o", tmp] + _SAN + objs)
        os.replace(tmp, OUT)
    from monodetr_amd import _capi
    L = ctypes.CDLL(OUT)
    for name, (res, args) in _capi.SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    return L
