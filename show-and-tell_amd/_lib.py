"""ctypes binding of libsat_hip.so (C ABI: include/sat_hip.h).

The library is the product: there is NO CPU or eager-PyTorch fallback.  If it is missing or a kernel call
fails, a RuntimeError is raised -- nothing is computed another way.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAT_LIB") or os.path.join(_HERE, "libsat_hip.so")    # SAT_LIB: A/B another build of the same ABI

ABI_VERSION = 18
SAT_F32, SAT_BF16 = 0, 1
OP_IMAGE_PREP, OP_CONV, OP_BN_FINALIZE, OP_BN_RELU, OP_BN_ADD_RELU, OP_BN_RELU_MAXPOOL, OP_AVGPOOL = 1, 2, 3, 4, 5, 6, 7

OP_BN_EVAL_BATCH = 8
OP_MAXPOOL2 = 9
OP_MAXPOOL3S2, OP_AVGPOOL3 = 10, 11
OP_GRAM, OP_GRAM_COV, OP_GEMM_BF16_NT, OP_BN_FROM_GRAM = 12, 13, 14, 15
CONV_PADW = 2
CONV_GROUP_TABLE = 4
CONV_IN_RESIDUAL = 8

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class SatBnEvalItem(C.Structure):
    """mirror of `struct sat_bn_eval_item` (include/sat_hip.h)"""
    _fields_ = [("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp),
                ("scale_out", _vp), ("shift_out", _vp), ("C", C.c_int32), ("reserved", C.c_int32)]


class SatBnRunningItem(C.Structure):
    """mirror of `struct sat_bn_running_item` (include/sat_hip.h)"""
    _fields_ = [("running_mean", _vp), ("running_var", _vp), ("batch_mean", _vp), ("batch_var", _vp),
                ("C", C.c_int32), ("reserved", C.c_int32)]


class SatOp(C.Structure):
    """mirror of `struct sat_op` (include/sat_hip.h)"""
    _fields_ = [
        ("kind", C.c_int32), ("dtype", C.c_int32),
        ("in0", _vp), ("in1", _vp), ("out", _vp), ("w", _vp),
        ("scale0", _vp), ("shift0", _vp), ("scale1", _vp), ("shift1", _vp),
        ("stat_partial", _vp), ("gamma", _vp), ("beta", _vp),
        ("running_mean", _vp), ("running_var", _vp), ("scale_out", _vp), ("shift_out", _vp),
        ("N", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("training", C.c_int32), ("tiles_m", C.c_int32),
        ("sN", C.c_int64), ("sH", C.c_int64), ("sW", C.c_int64), ("count", C.c_int64),
        ("momentum", C.c_float), ("eps", C.c_float),
        ("variant", C.c_int32), ("flags", C.c_int32),
        ("stat_acc", _vp), ("stat_acc1", _vp), ("gamma1", _vp), ("beta1", _vp),
        ("running_mean1", _vp), ("running_var1", _vp), ("w_packed", _vp),
        ("reserved1", C.c_int32 * 2),
        ("pad_w", C.c_int32), ("groups", C.c_int32), ("ldc", C.c_int64), ("out1", C.c_void_p),
    ]


# name -> (restype, argtypes): every symbol include/sat_hip.h declares
SIGNATURES = {
    "sat_version": (_i, []),
    "sat_error_string": (C.c_char_p, [_i]),
    "sat_gemm_f32": (_i, [_i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _vp]),
    "sat_run_ops": (_i, [C.POINTER(SatOp), _i, _vp]),
    "sat_run_ops_parity": (_i, [C.POINTER(SatOp), _i, _i, _vp]),
    "sat_graph_create": (_i, [C.POINTER(SatOp), _i, _i, C.POINTER(_vp)]),
    "sat_graph_launch": (_i, [_vp, _vp]),
    "sat_graph_destroy": (_i, [_vp]),
    "sat_conv_bn_relu_fwd": (_i, [C.POINTER(SatOp), C.POINTER(SatOp), C.POINTER(SatOp), _vp]),
    "sat_conv_tiles_m": (_i, [_i64]),
    "sat_counter_add": (_i, [_vp, _i, _i64, _vp]),
    "sat_conv_variant_signature": (_i, [_i]),
    "sat_conv_variant_family": (_i, [_i]),
    "sat_gram_rows_per_slab": (_i, [_i64, _i]),
    "sat_gram_slabs": (_i, [_i64, _i]),
    "sat_gram_slab_floats": (_i64, [_i64, _i]),
    "sat_conv_num_variants": (_i, []),
    "sat_conv_default_variant": (_i, [C.POINTER(SatOp), _i]),
    "sat_conv_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sat_conv_autotune": (_i, [C.POINTER(SatOp), _i, _i, _vp, _i64, _vp]),
    "sat_conv_autotune_topk": (_i, [C.POINTER(SatOp), _i, _i, _vp, _i64, _vp, _i, C.POINTER(C.c_int32)]),
    "sat_run_ops_timed": (_i, [C.POINTER(SatOp), _i, _i, _vp, C.POINTER(C.c_float)]),
    "sat_validate_ids": (_i, [_vp, _i64, _i, _i, _i64, _i64, _vp, _vp]),
    "sat_fc_bn1d_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_fc_bn1d_ws_bytes": (_i64, [_i, _i, _i]),
    "sat_fc_bn1d_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_embed_concat_fwd": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sat_embed_concat_bwd": (_i, [_vp, _vp, _i64, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sat_pack_targets": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp]),
    "sat_lstm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_lstm_mixed_ws_bytes": (_i64, [_i, _i, _i]),
    "sat_lstm_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "sat_lstm_bwd_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _i, _i,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "sat_lstm_fwd_ws_bytes": (_i64, [_i, _i]),
    "sat_lstm_fwd_status_offset": (_i64, [_i, _i]),
    "sat_lstm_persist_enable": (_i, [_i]),
    "sat_lstm_bwd_ws_bytes": (_i64, [_i, _i]),
    "sat_lstm_bwd_ws_bytes_full": (_i64, [_i, _i, _i, _i]),
    "sat_lstm_bwd_status_offset": (_i64, [_i, _i, _i, _i]),
    "sat_lstm_bwd_ws_bytes_max": (_i64, [_i, _i, _i, _i]),
    "sat_lstm_ws_release": (_i, [_vp]),
    "sat_lstm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _i, _i,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_vocab_logits_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp]),
    "sat_ce_rows": (_i, [_vp, _i64, _vp, _i, _i, _f, _i, _vp, _vp, _vp]),
    "sat_vocab_ce_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _i64, _vp, _vp, _vp]),
    "sat_vocab_bf16_ws_bytes": (_i64, [_i, _i, _i]),
    "sat_vocab_ce_fwd_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "sat_vocab_ce_bwd_bf16": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_gemm_bf16_nt": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i, _i, _i, _i, _i64, _vp]),
    "sat_transpose_f32_bf16": (_i, [_vp, _i64, _i, _i, _vp, _i64, _vp]),
    "sat_vocab_ce_bwd": (_i, [_vp, _i64, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_vocab_ce_bwd_ws_bytes": (_i64, [_i, _i, _i]),
    "sat_gemm_f32_splitk": (_i, [_i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i64, _vp]),
    "sat_gemm_f32x3_packed_bytes": (_i64, [_i, _i]),
    "sat_gemm_f32x3_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "sat_gemm_f32x3": (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "sat_sum_slabs_f32": (_i, [_vp, _i, _i64, _i64, _vp, _vp]),
    "sat_skinny_gemm_f32": (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _i64, _vp]),
    "sat_skinny_gemm_ws_bytes": (_i64, [_i, _i, _i]),
    "sat_skinny_gemm2_f32": (_i, [_vp, _i64, _vp, _i64, _i, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _i64, _vp, _i64, _vp]),
    "sat_vocab_argmax": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i64, _vp, _i64, _vp]),
    "sat_vocab_argmax_ws_bytes": (_i64, [_i, _i]),
    "sat_lstm_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sat_embed_rows": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "sat_attention_fwd": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _i, _i, _vp, _vp, _i64, _vp, _i64, _vp]),
    "sat_attention_bwd": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_attention_ws_bytes": (_i64, [_i, _i]),
    "sat_bn_running_apply": (_i, [_vp, _i, _f, _vp]),
    "sat_pad_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sat_maxpool2_bwd_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sat_bcast_add_f32": (_i, [_vp, _i, _i, _i, _f, _vp, _vp]),
    "sat_lstmcell_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sat_rows_copy": (_i, [_vp, _i64, _vp, _i64, _i64, _i, _i, _vp, _i64, _vp]),
    "sat_rows_sum": (_i, [_vp, _i64, _i, _i, _vp, _i, _vp]),
    "sat_rows_add": (_i, [_vp, _i64, _vp, _i64, _i, _i, _vp, _i64, _vp]),
    "sat_pack_tokens": (_i, [_vp, _i64, _vp, _i, _i, _i, _vp, _vp]),
    "sat_scatter_rows_add": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sat_lstmcell_bwd_point": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sat_collate_captions": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sat_gather_rows_f32": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "sat_beam_step": (_i, [_vp, _i64, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sat_beam_step_ws_bytes": (_i64, [_i, _i]),
    "sat_beam_gather_rows": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sat_beam_backtrack": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sat_kept_tokens": (_i, [_vp, _i64, _i, _i, _i64, _vp, _vp]),
    "sat_beam_decode_ws_bytes": (_i64, [_i, _i, _i, _i, _i, _i, _i]),
    "sat_beam_decode": (_i, [_vp, _vp, C.POINTER(_vp), _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _i64, _vp]),
    "sat_greedy_decode": (_i, [_vp, _vp, C.POINTER(_vp), _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "sat_clamp_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp]),
    "sat_clamp_adam_step_guarded": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "sat_step_fault_flag": (_i, [C.POINTER(_vp), _i, _vp, _vp, _vp]),
    "sat_colsum_f32": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "sat_cast_f32_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "sat_cast_bf16_f32": (_i, [_vp, _vp, _i64, _vp]),
}

_lib = None


def open_library(path):
    """dlopen one build of the library and bind every declared symbol (tests bind the -DSAT_TESTHOOKS build this way)"""
    if not os.path.exists(path):
        raise RuntimeError(
            "show-and-tell_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.sat_version() != ABI_VERSION:
        raise RuntimeError("%s: ABI version mismatch" % path)
    return lib


def load():
    """dlopen libsat_hip.so once; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        _lib = open_library(LIB_PATH)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().sat_error_string(rc)
        raise RuntimeError("libsat_hip %s failed: [%d] %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream():
    """torch's current HIP stream as a raw hipStream_t: every kernel is launched on it."""
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("show-and-tell_amd: %s must live on the MI355X (got a %s tensor); the HIP path has no CPU fallback"
                           % (name, t.device))


def counter_add(t, value=1):
    """t += value for an int64 device tensor (BatchNorm's `num_batches_tracked`), through the library: no torch operator computes
    anything on the product path"""
    check(load().sat_counter_add(t.data_ptr(), t.numel(), value, stream()), "sat_counter_add")
