"""ctypes binding of libmonodetr_amd.so (C ABI declared in include/monodetr_amd.h).

This is plumbing: it hands raw device pointers, sizes, the device ordinal and the current HIP
stream to the C ABI and turns negative return codes into ``RuntimeError`` (the reference's ATen
wrapper raises through AT_ASSERTM / AT_ERROR, ops/src/cuda/ms_deform_attn_cuda.cu:28-52).

There is NO fallback: if the shared library is missing or fails to load, importing symbols from
here raises.  Build it with ``python -m monodetr_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MDETR_LIB_PATH: developer switch for timing two builds of the library in one GPU call; the product never sets it)
LIB_PATH = os.environ.get("MDETR_LIB_PATH") or os.path.join(_HERE, "libmonodetr_amd.so")

MDETR_F32, MDETR_F64, MDETR_BF16 = 0, 1, 2
ABI_VERSION = 12

_c_int, _c_vp = ctypes.c_int, ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol of include/monodetr_amd.h
SIGNATURES = {
    "mdetr_abi_version": (_c_int, []),
    "mdetr_last_error": (ctypes.c_char_p, []),
    "mdetr_msda_variant": (_c_int, [_c_int] * 5),
    "mdetr_msda_forward": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int] * 7 + [_c_int, _c_vp]),
    "mdetr_msda_backward": (_c_int, [_c_int] + [_c_vp] * 9 + [_c_int] * 7 + [_c_int, _c_vp]),
    "mdetr_msda_forward_cpu": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int] * 7),
    "mdetr_msda_backward_cpu": (_c_int, [_c_int] + [_c_vp] * 9 + [_c_int] * 7),
    "mdetr_msda_backward_ex": (_c_int, [_c_int] + [_c_vp] * 9 + [_c_int] * 7 + [_c_vp, _c_vp, _c_vp, ctypes.c_int64] + [_c_int, _c_vp]),
    "mdetr_msda_backward_workspace_bytes": (ctypes.c_int64, [_c_int, _c_vp, _c_vp] + [_c_int] * 7),
    "mdetr_msda_indices": (_c_int, [_c_int] + [_c_vp] * 3 + [_c_int] * 5 + [_c_int, _c_vp]),
    "mdetr_attn_forward": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int] * 4 + [ctypes.c_int64] * 3 + [_c_int] * 3
                           + [ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "mdetr_attn_backward": (_c_int, [_c_int] + [_c_vp] * 11 + [_c_int] * 4 + [ctypes.c_int64] * 3 + [_c_int] * 3
                            + [ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "mdetr_lsa_forward": (_c_int, [_c_vp] * 3 + [_c_int] * 5 + [ctypes.c_int64] * 3 + [_c_int, _c_vp]),
    "mdetr_ddn_loss_forward": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_float] * 5 + [_c_vp, _c_vp, _c_int, _c_vp]),
    "mdetr_ddn_loss_backward": (_c_int, [_c_vp] * 4 + [_c_int] * 5 + [ctypes.c_int64] * 4 + [ctypes.c_float] * 5 + [_c_vp, _c_vp, _c_int, _c_vp]),
    "mdetr_pair_losses_workspace_bytes": (ctypes.c_int64, [_c_int, _c_int]),
    "mdetr_pair_losses_forward": (_c_int, [_c_vp] * 14 + [_c_int] * 6 + [ctypes.c_float, ctypes.c_float] + [_c_vp] * 4 + [_c_int, _c_vp]),
    "mdetr_pair_losses_backward": (_c_int, [_c_vp] * 13 + [_c_int] * 6 + [ctypes.c_float, ctypes.c_float] + [_c_vp] * 8 + [_c_int, _c_vp]),
    "mdetr_adamw_step": (_c_int, [_c_int] + [_c_vp] * 5 + [ctypes.c_int64] * 2 + [ctypes.c_float] * 5 + [_c_vp, _c_int, _c_vp]),
    "mdetr_adamw_step_gathered": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_int] + [_c_vp] * 5 + [_c_int, _c_vp, _c_vp, ctypes.c_int64] + [ctypes.c_float] * 5
                                  + [_c_vp, _c_vp, ctypes.c_float, _c_vp, _c_int, _c_vp]),
    "mdetr_adamw_step_counted": (_c_int, [_c_int] + [_c_vp] * 5 + [ctypes.c_int64] * 2 + [ctypes.c_float] * 4 + [_c_vp, ctypes.c_float, _c_vp, _c_int, _c_vp]),
    "mdetr_tgemm": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_int, _c_int] + [ctypes.c_int64] * 4 + [_c_int, ctypes.c_float, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "mdetr_sgemm_workspace_bytes": (ctypes.c_int64, [_c_int, _c_vp, _c_int]),
    "mdetr_sgemm_grouped": (_c_int, [_c_int, _c_vp, _c_int, _c_vp, ctypes.c_int64, _c_int, _c_vp]),
    "mdetr_tgemm_masked": (_c_int, [_c_vp] * 5 + [ctypes.c_int64, _c_int, _c_int] + [ctypes.c_int64] * 5 + [_c_int, _c_vp]),
    "mdetr_column_sum_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, _c_int]),
    "mdetr_column_sum": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_int, ctypes.c_int64, _c_int, _c_vp]),
    "mdetr_box_refine": (_c_int, [_c_vp] * 3 + [ctypes.c_int64, _c_int, _c_int, _c_vp]),
    "mdetr_head_tail_forward": (_c_int, [_c_vp] * 10 + [_c_int] * 6 + [_c_int, _c_vp]),
    "mdetr_head_tail_backward": (_c_int, [_c_vp] * 13 + [_c_int] * 6 + [_c_int, _c_vp]),
    "mdetr_chunk_sums": (_c_int, [_c_vp, _c_int, _c_int, _c_vp]),
    "mdetr_column_sum_to": (_c_int, [_c_int, _c_vp, _c_vp, _c_int, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_int, ctypes.c_int64, _c_int, _c_vp]),
    "mdetr_add_layernorm_forward": (_c_int, [_c_int, _c_int] + [_c_vp] * 7 + [ctypes.c_int64, _c_int, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "mdetr_add_layernorm_partial_rows": (ctypes.c_int64, [ctypes.c_int64]),
    "mdetr_add_layernorm_backward": (_c_int, [_c_int, _c_int] + [_c_vp] * 7 + [ctypes.c_int64, _c_int, ctypes.c_float, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "mdetr_conv3x3_forward": (_c_int, [_c_vp] * 4 + [_c_int] * 6 + [_c_int, _c_vp]),
    "mdetr_conv3x3_plan": (_c_int, [_c_int] * 4),
    "mdetr_conv3x3_masked": (_c_int, [_c_vp] * 5 + [_c_int] * 6 + [_c_int, _c_vp]),
    "mdetr_conv_taps": (_c_int, [_c_vp] * 5 + [_c_int, _c_int, _c_vp]),
    "mdetr_conv_taps_split": (_c_int, [_c_vp] * 4 + [ctypes.c_int64, _c_vp, _c_int, _c_int, _c_vp]),
    "mdetr_conv_dgrad_s2": (_c_int, [_c_vp] * 3 + [_c_int] * 8 + [_c_int, _c_vp]),
    "mdetr_conv_stem": (_c_int, [_c_vp] * 4 + [_c_int] * 3 + [_c_int, _c_vp]),
    "mdetr_conv_wgrad_chunks": (_c_int, [_c_int] * 9),
    "mdetr_conv_wgrad": (_c_int, [_c_vp] * 3 + [ctypes.c_int64] + [_c_int] * 9 + [_c_int, _c_vp]),
    "mdetr_token_wgrad_chunks": (_c_int, [ctypes.c_int64, _c_int, _c_int]),
    "mdetr_token_wgrad": (_c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mdetr_small_wgrad_workspace_bytes": (ctypes.c_int64, [ctypes.c_int64, _c_int, _c_int]),
    "mdetr_small_wgrad": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_int, _c_vp, ctypes.c_int64, ctypes.c_int64, _c_int, _c_int, ctypes.c_int64, ctypes.c_int64, _c_int, _c_vp]),
    "mdetr_group_norm_workspace_bytes": (ctypes.c_int64, [_c_int, ctypes.c_int64, _c_int, _c_int]),
    "mdetr_group_norm_forward": (_c_int, [_c_int, _c_int] + [_c_vp] * 6 + [ctypes.c_int64, _c_int, ctypes.c_int64, _c_int, _c_int, ctypes.c_float, _c_int, _c_int, _c_vp]),
    "mdetr_group_norm_backward": (_c_int, [_c_int, _c_int] + [_c_vp] * 8 + [ctypes.c_int64, _c_int, ctypes.c_int64, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mdetr_bias_act_forward": (_c_int, [_c_int, _c_int] + [_c_vp] * 4 + [ctypes.c_int64, _c_int, _c_int, ctypes.c_float, ctypes.c_uint64, _c_vp, _c_int, _c_vp]),
    "mdetr_bias_act_backward": (_c_int, [_c_int] + [_c_vp] * 3 + [ctypes.c_int64, _c_int, ctypes.c_float, _c_int, _c_vp]),
    "mdetr_rotate_iou_eval": (_c_int, [_c_vp] * 5 + [_c_int, ctypes.c_int64, _c_int, _c_vp, _c_int, _c_vp]),
    "mdetr_box3d_overlap_eval": (_c_int, [_c_vp] * 5 + [_c_int, ctypes.c_int64, _c_int, _c_vp, _c_int, _c_vp]),
    "mdetr_kitti_pr_curve": (_c_int, [_c_vp] * 10 + [_c_int, _c_int, ctypes.c_double, ctypes.c_int64, _c_int, _c_int, _c_vp, _c_vp, _c_vp]),
    "mdetr_kitti_preprocess": (_c_int, [_c_vp, _c_vp, _c_int, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_vp, _c_int, _c_vp]),
    "mdetr_msda_forward_bf16": (_c_int, [_c_vp] * 6 + [_c_int] * 7 + [_c_int, _c_vp]),
    "mdetr_msda_backward_bf16": (_c_int, [_c_vp] * 9 + [_c_int] * 7 + [_c_vp, _c_vp, _c_vp, ctypes.c_int64, _c_int, _c_vp]),
    "mdetr_msda_prologue_forward": (_c_int, [_c_int, _c_int] + [_c_vp] * 6 + [_c_int] * 6 + [ctypes.c_int64] * 3 + [_c_int, _c_vp]),
    "mdetr_gather_flat": (_c_int, [_c_vp, _c_int] + [_c_vp] * 6 + [_c_int, _c_int, _c_vp]),
    "mdetr_fold_weights": (_c_int, [_c_int] + [_c_vp] * 7 + [_c_int, _c_vp]),
    "mdetr_unfold_grads": (_c_int, [_c_int] + [_c_vp] * 6 + [_c_int, _c_vp]),
    "mdetr_maxpool3x3s2_bf16": (_c_int, [_c_vp, _c_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp]),
    "mdetr_decimate2": (_c_int, [_c_int, _c_vp, _c_vp, _c_int, _c_int, _c_int, ctypes.c_int64, _c_int, _c_vp]),
    "mdetr_msda_prologue_forward_packed": (_c_int, [_c_int, _c_int] + [_c_vp] * 5 + [_c_int] * 6 + [ctypes.c_int64] * 3 + [_c_int, _c_vp]),
    "mdetr_msda_prologue_backward_packed": (_c_int, [_c_int, _c_int] + [_c_vp] * 8 + [_c_int] * 6 + [ctypes.c_int64] * 3 + [_c_int, _c_vp]),
    "mdetr_msda_prologue_backward": (_c_int, [_c_int, _c_int] + [_c_vp] * 9 + [_c_int] * 6 + [ctypes.c_int64] * 3 + [_c_int, _c_vp]),
    "mdetr_lsa_forward_fused": (_c_int, [_c_vp] * 6 + [_c_int] * 6 + [ctypes.c_float] * 5 + [_c_int, _c_vp]),
    "mdetr_profile_enable": (_c_int, [_c_int]),
    "mdetr_profile_read": (_c_int, [_c_vp, _c_int]),
    "mdetr_profile_read_work": (_c_int, [_c_vp, _c_int]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """The loaded library (loads on first use). Raises if it is not built -- no fallback."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "monodetr_amd: %s is missing -- the HIP extension is not built. Run "
                        "`python -m monodetr_amd.build` (needs hipcc). There is no CPU/PyTorch fallback." % LIB_PATH)
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)          # AttributeError if the .so is stale
                    fn.restype, fn.argtypes = res, args
                got = handle.mdetr_abi_version()
                if got != ABI_VERSION:
                    raise RuntimeError("monodetr_amd: ABI version mismatch (lib %d, python %d); rebuild" % (got, ABI_VERSION))
                _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().mdetr_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def dtype_code(t):
    import torch
    if t.dtype == torch.float32:
        return MDETR_F32
    if t.dtype == torch.float64:
        return MDETR_F64
    # same wording as AT_DISPATCH_FLOATING_TYPES (ms_deform_attn_cuda.cu:64,134)
    raise RuntimeError('"ms_deform_attn" not implemented for \'%s\'' % str(t.dtype).replace("torch.", ""))


def profile_enable(on):
    check(lib().mdetr_profile_enable(1 if on else 0), "mdetr_profile_enable")


def profile_read(cap=64):
    """[(kind, Lq, launches, total_ms)] of the kernel launches recorded since profile_enable(True);
    kind 0 = msda forward, 1 = msda backward."""
    buf = (ctypes.c_double * (4 * cap))()
    n = lib().mdetr_profile_read(ctypes.cast(buf, ctypes.c_void_p), cap)
    if n < 0:
        check(n, "mdetr_profile_read")
    return [(int(buf[4 * i]), int(buf[4 * i + 1]), int(buf[4 * i + 2]), float(buf[4 * i + 3])) for i in range(n)]


def profile_read_work(cap=256):
    """[(kind, key, launches, total_ms, mflop, kbytes)]: `profile_read` plus the useful work the launches declared."""
    buf = (ctypes.c_double * (6 * cap))()
    n = lib().mdetr_profile_read_work(ctypes.cast(buf, ctypes.c_void_p), cap)
    if n < 0:
        check(n, "mdetr_profile_read_work")
    return [(int(buf[6 * i]), int(buf[6 * i + 1]), int(buf[6 * i + 2]), float(buf[6 * i + 3]), float(buf[6 * i + 4]), float(buf[6 * i + 5])) for i in range(n)]
