"""ctypes binding of the C ABI declared in `include/eqxvision_amd.h`.

The shared library is built in-tree by `eqxvision_amd.build.build()` (hipcc, gfx950) into
`eqxvision_amd/csrc/libeqxvision_amd.so`.  There is NO fallback: if the library is missing or
a call fails, a `RuntimeError` is raised -- the product never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import _trace

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EQV_LIB") or os.path.join(_HERE, "csrc", "libeqxvision_amd.so")   # EQV_LIB: tools only (debug build)

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU_TANH = 0, 1, 2
ACT_HARD_SWISH, ACT_HARD_SIGMOID, ACT_SIGMOID, ACT_SILU = 3, 4, 5, 6       # element-wise entries and the depthwise conv only
ABI_VERSION = 2

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

# name -> argtypes (restype is int unless listed in _RESTYPES)
PROTOTYPES = {
    "mv_abi_version": [],
    "mv_last_error": [],
    "mv_last_kernel": [],
    "mv_set_flag": [C.c_char_p, _i],
    "mv_get_flag": [C.c_char_p],
    "mv_flags_epoch": [],
    "mv_device_status": [_i, _vp],
    "mv_comm_unique_id": [_vp, C.c_size_t],
    "mv_comm_init": [_i, _i, _vp],
    "mv_comm_size": [],
    "mv_comm_rank": [],
    "mv_allgather": [_vp, _vp, C.c_size_t, _vp],
    "mv_allreduce_sum_f32": [_vp, C.c_size_t, _vp],
    "mv_swin_window_attn_dropout_fwd": [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "mv_dropout_windows_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp],
    "mv_drop_path_noise": [_vp, _vp, _i, _i, _i, _f, _i, _vp],
    "mv_prng_split": [_vp, _vp, _i64, _i, _i, _vp],
    "mv_dropout_fwd": [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _i, _vp],
    "mv_channel_moments2_fwd": [_vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "mv_bn_ema_fold1_fwd": [_vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp],
    "mv_bn_mean_fwd": [_vp, _vp, _f, _vp, _i, _vp],
    "mv_bn_ema_fold_fwd": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp],
    "mv_se_scale_supported": [_i, _i, _i],
    "mv_se_scale_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _vp],
    "mv_channel_moments_ws": [_i],
    "mv_channel_moments_supported": [_i64, _i, _i],
    "mv_channel_moments_fwd": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "mv_comm_destroy": [],
    "mv_conv2d_nhwc_fwd": [_vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 14 + [_i, _i, _i, _vp],
    "mv_conv2d_nchw_fwd": [_vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_i, _i, _i, _i, _i, _vp, _vp],
    "mv_linear_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp],
    "mv_linear_split_supported": [_i64, _i, _i, _i],
    "mv_linear_split_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp],
    "mv_channel_scale_nhwc_fwd": [_vp, _vp, _vp, _i, _i64, _i, _i, _vp],
    "mv_dwconv2d_supported": [_i] * 7,
    "mv_dwconv2d_nhwc_fwd": [_vp, _vp, _vp, _vp, _vp] + [_i] * 12 + [_i, _i, _i, _vp],
    "mv_conv2d_grouped64_supported": [_i] * 7,
    "mv_conv2d_grouped64_window": [_i, _i],
    "mv_conv2d_nhwc_grouped64_fwd": [_vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 14 + [_i, _i, _i, _vp],
    "mv_ln_linear_supported": [_i64, _i, _i, _i, _i],
    "mv_ln_linear_fwd": [_vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _i, _i, _vp],
    "mv_ln_mlp_supported": [_i64, _i, _i, _i],
    "mv_ln_mlp_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp],
    "mv_swin_block_attn_supported": [_i] * 7,
    "mv_swin_block_attn_fwd": [_vp] * 7 + [_i] * 9 + [_f, _i, _vp],
    "mv_conv2d_nchw_f32out_supported": [_i] * 11,
    "mv_conv2d_nchw_f32out_fwd": [_vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_i, _i, _i, _i, _vp, _vp],
    "mv_fc_stream_supported": [_i64, _i, _i, _i, _i],
    "mv_fc_stream_workspace": [_i64, _i, _i],
    "mv_set_scratch": [_vp, _i64, _vp],
    "mv_splitk_scratch_bytes": [_i64, _i64, _i64],
    "mv_fc_stream_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp],
    "mv_ln_mlp_stream_supported": [_i64, _i, _i, _i],
    "mv_ln_mlp_stream_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _vp],
    "mv_conv2d_nchw_split_fwd": [_vp, _vp, _vp, _vp, _vp, _vp] + [_i] * 11 + [_i, _i, _i, _vp],
    "mv_resize_bilinear_nhwc_fwd": [_vp, _vp] + [_i] * 6 + [_i, _i, _i, _vp],
    "mv_copy_rows": [_vp, _vp, _i64, _i64, _i64, _i64, _vp],
    "mv_maxpool2d_nhwc_fwd": [_vp, _vp] + [_i] * 10 + [_i, _vp],
    "mv_adaptive_avgpool2d_nhwc_fwd": [_vp, _vp] + [_i] * 6 + [_i, _i, _vp],
    "mv_layernorm_fwd": [_vp, _vp, _vp, _vp, _i64, _i, _i64, _f, _i, _i, _vp],
    "mv_mha_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "mv_stem_conv_pool_supported": [_i] * 14 + [_i64],
    "mv_stem_conv_pool_fwd": [_vp, _vp, _vp, _vp, _vp] + [_i] * 17 + [_vp],
    "mv_conv1x1_chain_supported": [_i64, _i, _i, _i, _i],
    "mv_conv1x1_chain_fwd": [_vp] * 10 + [_i64, _i, _i, _i, _i, _vp],
    "mv_bottleneck_tail_supported": [_i] * 5,
    "mv_bottleneck_tail_fwd": [_vp] * 9 + [_i] * 6 + [_vp],
    "mv_conv1x1_dual_supported": [_i64, _i, _i, _i, _i],
    "mv_conv1x1_dual_fwd": [_vp] * 6 + [_i] * 11 + [_vp],
    "mv_conv1x1_chain_res_supported": [_i] * 8,
    "mv_conv1x1_chain_res_fwd": [_vp] * 6 + [_i] * 8 + [_vp],
    "mv_conv1x1_chain_rc_supported": [_i64, _i, _i, _i, _i],
    "mv_conv1x1_chain_rc_fwd": [_vp] * 7 + [_i64, _i, _i, _i, _i, _vp],
    "mv_conv1x1_chain_rc0_fwd": [_vp] * 5 + [_i64, _i, _i, _i, _i, _vp],
    "mv_conv1x1_chain_sub_supported": [_i] * 7,
    "mv_conv1x1_chain_sub_fwd": [_vp] * 10 + [_i] * 7 + [_vp],
    "mv_conv1x1_dual_chain_supported": [_i64, _i, _i, _i, _i, _i],
    "mv_conv1x1_dual_chain_fwd": [_vp] * 10 + [_i64, _i, _i, _i, _i, _i, _vp],
    "mv_linear_heads_supported": [_i64, _i, _i, _i, _i, _i],
    "mv_linear_lnout_supported": [_i64, _i, _i, _i],
    "mv_linear_lnout_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "mv_linear_lnin_supported": [_i64, _i, _i, _i, _i, _i],
    "mv_linear_lnin_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _i, _i, _i, _vp],
    "mv_linear_heads_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp],
    "mv_mha_heads_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "mv_mha_dropout_fwd": [_vp, _i, _vp, _vp, _vp, _f, _i, _i, _i, _i, _f, _i, _vp],
    "mv_swin_window_attn_fwd": [_vp, _vp, _vp] + [_i] * 9 + [_i, _vp],
    "mv_patch_merge_gather_nhwc": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "mv_patch4_ln_supported": [_i] * 5,
    "mv_patch4_ln_fwd": [_vp] * 7 + [_i] * 5 + [_f, _i, _vp],
    "mv_patch_merge_ln_supported": [_i] * 4,
    "mv_patch_merge_ln_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "mv_vit_cls_pos_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "mv_eltwise_fwd": [_vp, _vp, _i64, _i, _i, _vp],
    "mv_add_fwd": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "mv_channel_affine_fwd": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "mv_channel_affine_res_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "mv_nchw_to_nhwc": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mv_nhwc_to_nchw": [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "mv_cast": [_vp, _vp, _i64, _i, _i, _vp],
    "mv_conv2d_dgrad_nhwc_f32": [_vp, _vp, _vp] + [_i] * 14 + [_vp],
    "mv_conv2d_wgrad_nhwc_f32": [_vp, _vp, _vp] + [_i] * 14 + [_vp],
    "mv_channel_scale_bwd_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mv_act_bwd_f32": [_vp, _vp, _vp, _i64, _i, _vp],
    "mv_maxpool2d_bwd_nhwc_f32": [_vp, _vp, _vp] + [_i] * 10 + [_vp],
    "mv_avgpool_global_bwd_nhwc_f32": [_vp, _vp, _i, _i, _i, _vp],
    "mv_colsum_f32": [_vp, _vp, _vp, _i64, _i, _vp],
    "mv_bn_dgamma_f32": [_vp, _vp, _vp, _vp, _f, _vp, _i, _vp],
    "mv_bn_train_dz_coef_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _i, _vp],
    "mv_layernorm_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp],
    "mv_softmax_bwd_f32": [_vp, _vp, _vp, _i64, _i, _f, _vp],
    "mv_mha_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp],
    "mv_softmax_xent_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "mv_adam_step_f32": [_vp, _vp, _vp, _vp, _i64] + [_f] * 6 + [_vp],
    "mv_transpose2d_f32": [_vp, _vp, _i, _i, _i64, _vp],
    "mv_swin_window_attn_bwd_f32": [_vp] * 5 + [_i] * 9 + [_vp],
    "mv_scatter_rows_sum_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "mv_patch_merge_gather_bwd_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "mv_graph_begin_capture": [_vp],
    "mv_graph_end_capture": [_vp, C.POINTER(_vp)],
    "mv_graph_launch": [_vp, _vp],
    "mv_graph_destroy": [_vp],
    "mv_event_create": [C.POINTER(_vp)],
    "mv_event_record": [_vp, _vp],
    "mv_event_elapsed_ms": [_vp, _vp, C.POINTER(_f)],
    "mv_event_destroy": [_vp],
}
_RESTYPES = {"mv_last_error": C.c_char_p, "mv_last_kernel": C.c_char_p, "mv_fc_stream_workspace": _i64,
             "mv_splitk_scratch_bytes": _i64}

_lib = None
_lock = threading.Lock()


class MVError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises RuntimeError if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MVError(
                f"eqxvision_amd HIP library not built: {LIB_PATH} is missing. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback."
            )
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(lib, name)  # AttributeError here == ABI mismatch, loud by design
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _i)
        if lib.mv_abi_version() != ABI_VERSION:
            raise MVError(f"ABI mismatch: library {lib.mv_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


# When a forward is being recorded by `eqxvision_amd.filter_jit`, every kernel-enqueueing call is
# appended here as (bound C function, argument tuple) so it can be replayed / graph-captured.
_tls = threading.local()        # .rec: the launch list being recorded on THIS thread (one trace per thread at a time)
_NOT_RECORDED = ("mv_set_flag", "mv_graph_", "mv_event_", "mv_comm_")
COMM_ID_BYTES = 128


def set_recording(rec):
    old = getattr(_tls, "rec", None)
    _tls.rec = rec
    _tls.pending_scratch = None        # a hand-over never outlives the recording it was made in
    if rec is not None:
        _tls.not_replayable = None
    if _lib is not None:               # ... on the library's side of the ABI either (advisor, round 4)
        _lib.mv_set_scratch(None, 0, None)
    return old


def mark_not_replayable(reason: str):
    """Called by code whose results are HOST values or whose work is not a pure list of launches (filter_value_and_grad: the loss is a
    Python float, gradients are fetched after a synchronize): a recording in progress on this thread must not be replayed."""
    if getattr(_tls, "rec", None) is not None:
        _tls.not_replayable = reason


def not_replayable():
    return getattr(_tls, "not_replayable", None)


_grad_guard = None      # set by eqxvision_amd.grad: refuses launches of un-differentiable ops inside filter_value_and_grad


def call(name, *args):
    lib = load()
    if _grad_guard is not None:
        _grad_guard(name)
    fn = getattr(lib, name)
    if _trace.enabled:
        _trace.push(name)
        try:
            rc = fn(*args)
        finally:
            _trace.pop()
    else:
        rc = fn(*args)
    if rc != 0:
        msg = lib.mv_last_error()
        raise MVError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")
    rec = getattr(_tls, "rec", None)
    if rec is not None and not name.startswith(_NOT_RECORDED):
        if name == "mv_set_scratch":                   # not a launch: it travels with the launch it was set for
            _tls.pending_scratch = args[:2]
            return rc
        ps = getattr(_tls, "pending_scratch", None)
        if ps is not None:
            _tls.pending_scratch = None
            fn = _with_scratch(lib.mv_set_scratch, ps, fn)
        rec.append((fn, args, name))
    return rc


def _with_scratch(setter, scratch, fn):
    """A recorded launch that was handed scratch memory: every replay hands it over again, on the stream it is replayed on."""
    ptr, nbytes = scratch

    def launch(*a):
        setter(ptr, nbytes, a[-1])
        return fn(*a)
    return launch


def last_kernel() -> str:
    s = load().mv_last_kernel()
    return s.decode() if s else ""


def set_flag(name: str, value: int):
    call("mv_set_flag", name.encode(), int(value))


def check_device_status(clear: bool = True) -> int:
    """Reads (and clears) the library's device status word; raises if a kernel recorded a broken run-time protocol (a split-K block
    that gave up on its partner: scratch shared between concurrent launches, or not zeroed).  Synchronises the device."""
    v = C.c_uint(0)
    rc = load().mv_device_status(1 if clear else 0, C.byref(v))
    if rc != 0:
        msg = load().mv_last_error()
        raise MVError(f"mv_device_status failed (rc={rc}): {msg.decode() if msg else ''}")
    if v.value:
        raise MVError(f"device status 0x{v.value:x}: a split-K tile was finished without its partner's partial sums -- the scratch of "
                      "mv_set_scratch was shared by concurrent launches or not zero-initialised; results of that launch are wrong")
    return 0


def get_flag(name: str) -> int:
    return load().mv_get_flag(name.encode())
