"""ctypes binding of libieagan_hip.so (the C ABI declared in include/ieagan_hip.h).

There is deliberately NO fallback: if the shared object is missing or a kernel launcher reports an
error, a RuntimeError is raised.  The product path never routes through PyTorch eager convolutions
or through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libieagan_hip.so")
CONV_FP8_NOSCALE = 8     # with CONV_FP8 (benchmarks): the non-scaled K = 32 fp8 MFMA
CONV_FP8 = 4             # ieagan_conv_desc.flags bit: e4m3 MFMA operands in the forward C = 64 / 128 3x3 launches (configs[4])
CONV_NO_LDS_WEIGHTS = 2  # ieagan_conv_desc.flags bit: C = 64 / 128 3x3 layers through conv3x3_halo instead of conv3x3_lds
CONV_FORCE_GATHER = 1   # ieagan_conv_desc.flags bit (tests): route a 3x3 layer through the gather kernel
B1_OCC2, B1_OCC3, B1_TP32 = 1, 2, 4  # ieagan_conv1x1_bwd_desc.flags bits (benchmarks)
BWD_NO_REDUCE = 16      # ieagan_conv1x1_bwd / ieagan_conv3x3_bwd: the caller folds the dW slabs (ieagan_wgrad_reduce)
PROLOGUE_BWD_SLOTS = 64  # include/ieagan_hip.h: IEAGAN_PROLOGUE_BWD_SLOTS
AUG_SLOTS = 128         # include/ieagan_hip.h: IEAGAN_AUG_SLOTS (per-image partial-sum slots of the DiffAugment entry points)
BNB_REPL = 8            # replicas of the per-image accumulators of a BatchNorm-backward dgrad launch (common.h)
STAT_REPL = 32          # replicas of every (sum, sumsq) statistics buffer (common.h)
SN_FIELDS = 16          # int64 fields per row of the spectral-norm layer table (sn.hip)

vp, fp, lp, ip = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p   # all device pointers travel as void*


class SrcDesc(C.Structure):
    _fields_ = [("x", vp), ("Cx", C.c_int), ("Hs", C.c_int), ("Ws", C.c_int), ("rs", C.c_int),
                ("scale", vp), ("shift", vp), ("aff_nstride", C.c_int), ("relu", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("taps", C.c_int), ("Kpad", C.c_int), ("src", SrcDesc), ("w", vp), ("bias", vp),
                ("ra", vp), ("Cra", C.c_int), ("Ca", C.c_int), ("ra_rs", C.c_int), ("ra_scale", C.c_float), ("rb", vp),
                ("Crb", C.c_int), ("mask", vp), ("out", vp), ("stats", vp), ("n_per_event", C.c_int), ("flags", C.c_int),
                ("bnb_scale", vp), ("bnb_shift", vp), ("bnb_nstride", C.c_int), ("bnb_relu", C.c_int), ("stats_slots", C.c_int)]


class WgradDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("taps", C.c_int), ("Kpad", C.c_int), ("src", SrcDesc), ("g", vp), ("Cg", C.c_int),
                ("dw", vp), ("tiles_per_block", C.c_int), ("pad_rows", C.c_int), ("partials", vp), ("colsum", vp)]


class Conv1x1BwdDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("Kpad", C.c_int),
                ("Kpad2", C.c_int), ("src", SrcDesc), ("g", vp), ("Cg", C.c_int), ("y", vp), ("dstat", vp), ("n_per_event", C.c_int),
                ("geff_out", vp), ("w_bwd", vp), ("lg", vp), ("lC", C.c_int), ("lCa", C.c_int), ("lmode", C.c_int), ("dx", vp),
                ("out_mode", C.c_int), ("bn_acc", vp), ("dw", vp), ("partials", vp), ("colsum", vp), ("flags", C.c_int), ("bn_slots", C.c_int)]


class Conv3x3BwdDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("Kpad", C.c_int), ("src", SrcDesc), ("g", vp), ("Cg", C.c_int),
                ("y", vp), ("dstat", vp), ("n_per_event", C.c_int), ("w_bwd", vp), ("dx", vp), ("bn_acc", vp), ("dw", vp), ("partials", vp),
                ("colsum", vp), ("flags", C.c_int), ("bn_slots", C.c_int)]


class ReduceItem(C.Structure):
    _fields_ = [("partials", vp), ("dw", vp), ("S", C.c_int), ("Cout", C.c_int), ("Kpad", C.c_int), ("K", C.c_int)]


class DStemDesc(C.Structure):
    _fields_ = [("img", vp), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("w_in", vp), ("b_in", vp), ("w1", vp), ("b1", vp), ("wsc", vp),
                ("bsc", vp), ("h1", vp), ("p0", vp), ("sc", vp), ("dh1", vp), ("dp0", vp), ("w1_bwd", vp), ("dw_in", vp), ("db_in", vp),
                ("dw1", vp), ("db1", vp)]


class ProfRec(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_long), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double), ("bytes_min", C.c_double)]


ABI_VERSION = 11           # include/ieagan_hip.h: IEAGAN_ABI_VERSION
i, f, l = C.c_int, C.c_float, C.c_long
_SIGS = {
    "ieagan_abi_version": [],
    "ieagan_prof_enable": [i],
    "ieagan_prof_reset": [],
    "ieagan_prof_collect": [C.POINTER(ProfRec), i],
    "ieagan_conv_forward": [C.POINTER(ConvDesc), vp],
    "ieagan_conv_wgrad": [C.POINTER(WgradDesc), i, vp],
    "ieagan_conv_wgrad_workspace": [C.POINTER(WgradDesc), i],
    "ieagan_conv1x1_bwd": [C.POINTER(Conv1x1BwdDesc), vp],
    "ieagan_conv1x1_bwd_workspace": [C.POINTER(Conv1x1BwdDesc)],
    "ieagan_conv1x1_bwd_supported": [i, i, i, i],
    "ieagan_conv3x3_bwd": [C.POINTER(Conv3x3BwdDesc), vp],
    "ieagan_conv3x3_bwd_workspace": [C.POINTER(Conv3x3BwdDesc)],
    "ieagan_conv3x3_bwd_supported": [i, i, i, i, i, i, i],
    "ieagan_conv3x3_bwd_slots": [C.POINTER(Conv3x3BwdDesc)],
    "ieagan_conv1x1_bwd_slots": [C.POINTER(Conv1x1BwdDesc)],
    "ieagan_wgrad_reduce": [vp, vp, i, i, i, i, vp],
    "ieagan_wgrad_reduce_batched": [C.POINTER(ReduceItem), i, vp],
    "ieagan_conv_stats_slots": [C.POINTER(ConvDesc)],
    "ieagan_d_stem_fwd": [C.POINTER(DStemDesc), vp],
    "ieagan_d_stem_bwd": [C.POINTER(DStemDesc), vp],
    "ieagan_effgrad": [vp, vp, vp, vp, vp, l, i, i, vp],
    "ieagan_prologue_bwd": [vp, vp, i, vp, vp, i, i, i, vp, vp, vp, i, i, i, i, vp, i, i, i, i, vp],
    "ieagan_bn_finalize_fwd": [vp, f, vp, vp, i, i, f, f, i, vp, vp, vp, vp, vp, i, i, i, i, vp, vp],
    "ieagan_bn_finalize_bwd": [vp, vp, vp, i, i, vp, f, i, vp, vp, i, vp, i, i, i, i, vp, vp],
    "ieagan_bn_finalize_fwd_scratch": [i, i, i],
    "ieagan_bn_finalize_bwd_scratch": [i, i, i, i],
    "ieagan_res_bwd": [vp, i, vp, i, i, i, i, i, i, vp],
    "ieagan_nchw_to_nhwc": [vp, vp, vp, i, i, i, i, vp],
    "ieagan_nhwc_to_nchw": [vp, vp, i, i, i, vp],
    "ieagan_channel_stats": [vp, vp, l, i, vp],
    "ieagan_conv_1toC": [vp, vp, vp, vp, vp, i, i, i, i, i, vp],
    "ieagan_conv_1toC_bnb": [vp, vp, vp, vp, vp, vp, i, i, vp, vp, vp, i, i, i, i, i, i, vp],
    "ieagan_conv_1toC_bnb_slots": [i, i, i, i],
    "ieagan_conv_Cto1": [vp, vp, vp, i, i, vp, vp, vp, i, i, i, i, i, i, vp],
    "ieagan_wgrad_c1": [vp, vp, vp, vp, vp, i, i, vp, i, i, i, i, i, vp],
    "ieagan_sn_backward_batched": [vp, vp, i, vp, vp, vp, vp, vp],
    "ieagan_sn_forward": [vp, vp, i, vp, i, vp, vp, vp, vp, f, i, vp],
    "ieagan_sn_backward": [vp, vp, i, i, i, i, i, i, vp, vp, vp, i, vp, vp, i, vp],
    "ieagan_sn_backward_stack": [vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp],
    "ieagan_nl_attention_fwd": [vp, vp, vp, vp, vp, i, i, i, i, i, vp],
    "ieagan_nl_attention_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp],
    "ieagan_rrm_attention_fwd": [vp, vp, vp, i, i, i, i, vp],
    "ieagan_rrm_attention_bwd": [vp, vp, vp, vp, i, i, i, i, vp],
    "ieagan_slin_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, f, vp],
    "ieagan_slin_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp],
    "ieagan_ln_fwd": [vp, vp, vp, vp, vp, vp, i, i, f, i, vp],
    "ieagan_ln_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp],
    "ieagan_embed_norm_fwd": [vp, vp, vp, vp, i, i, vp],
    "ieagan_embed_norm_bwd": [vp, vp, vp, vp, vp, i, i, vp],
    "ieagan_loss_block": [vp, vp, vp, vp, vp, C.POINTER(C.c_float), f, vp, vp, vp, vp, vp, i, i, vp],
    "ieagan_loss_block_events": [vp, vp, vp, vp, vp, C.POINTER(C.c_float), f, vp, vp, vp, vp, vp, i, i, i, vp],
    "ieagan_relu_sum_pool": [vp, vp, i, i, i, vp],
    "ieagan_relu_sum_pool_bwd": [vp, vp, vp, i, i, i, vp],
    "ieagan_diffaug_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, vp],
    "ieagan_diffaug_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, vp],
    "ieagan_cr_diffaug": [vp, vp, vp, vp, vp, i, i, i, vp],
    "ieagan_event_ingest": [vp, vp, vp, i, i, i, i, f, vp],
    "ieagan_adam_step": [vp, vp, vp, vp, l, vp, vp],
    "ieagan_ema_update": [vp, vp, l, vp, vp],
    "ieagan_maxpool2_fwd": [vp, vp, vp, i, i, i, i, vp],
    "ieagan_maxpool2_bwd": [vp, vp, vp, i, i, i, i, vp],
    "ieagan_gamma_residual_fwd": [vp, vp, vp, vp, l, vp],
    "ieagan_gamma_residual_bwd": [vp, vp, vp, vp, vp, l, vp],
    "ieagan_ortho_ksplit": [],
    "ieagan_ortho_grad": [vp, vp, vp, vp, i, vp, i, vp, l, f, vp],
    "ieagan_selftest_tr_read": [vp, vp, vp],
}
EXPORTS = ["ieagan_last_error"] + list(_SIGS)

_lib = None


def lib():
    """Load the shared object once (after torch, so that it binds to torch's HIP runtime)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(iea-gan_amd/csrc/build.sh).  There is no PyTorch/CPU fallback for the MI355X path.")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.ieagan_abi_version.restype = C.c_int
        if _lib.ieagan_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {_lib.ieagan_abi_version()}, the Python binding expects {ABI_VERSION}: "
                               "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        _lib.ieagan_last_error.restype = C.c_char_p
        for name, sig in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = sig
            fn.restype = C.c_long if name.endswith(("_workspace", "_scratch")) else C.c_int
    return _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("the IEA-GAN MI355X path needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback -- use oracle/ only as a test checker")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib().ieagan_last_error().decode()}")


# ---- small helpers shared by the op layer ----------------------------------------------------------
def src_desc(x, Cx, Hs, Ws, rs=0, scale=None, shift=None, nstride=0, relu=False) -> SrcDesc:
    return SrcDesc(ptr(x), Cx, Hs, Ws, rs, ptr(scale), ptr(shift), nstride, int(bool(relu)))


def prof_enable(on) -> None:
    call("ieagan_prof_enable", int(on))


def prof_collect() -> list:
    buf = (ProfRec * 512)()
    n = lib().ieagan_prof_collect(buf, 512)
    return [dict(name=buf[k].name.decode(), launches=buf[k].launches, ms=buf[k].ms, flops=buf[k].flops,
                 bytes=buf[k].bytes, bytes_min=buf[k].bytes_min) for k in range(n)]
