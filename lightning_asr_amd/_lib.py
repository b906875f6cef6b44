"""ctypes binding of liblasr.so (include/lasr.h).  There is no CPU fallback: if the HIP library is
missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LASR_LIB_PATH", os.path.join(_HERE, "liblasr.so"))   # override: A/B of two builds

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SWISH = 0, 1, 2
LEN_LEAD = 1 << 30          # LASR_LEN_LEAD: a row of samples that starts with one lead-in sample (crop after pre-emphasis)
VARIANT = {"plain": 0, "context": 1, "context_se": 2}

_p, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t


class Dropout(C.Structure):
    """lasr_dropout: device step counter, seed, unit index, p"""
    _fields_ = [("step", C.c_void_p), ("seed", C.c_uint64), ("unit", C.c_uint32), ("p", C.c_float)]


class WaveSrc(C.Structure):
    """lasr_wave_src: samples (f32 or int16 PCM) + explicit dither noise or (seed, device step counter) for generated noise"""
    _fields_ = [("wave", C.c_void_p), ("wave_dtype", C.c_int32), ("dither", C.c_void_p), ("dither_seed", C.c_uint64),
                ("dither_step", C.c_void_p), ("pitch", C.c_int64)]


WAVE_F32, WAVE_PCM16 = 0, 1


class ModelConfig(C.Structure):
    _fields_ = [("variant", C.c_int32), ("n_class", C.c_int32), ("in_c", C.c_int32),
                ("mask", C.c_int32), ("act", C.c_int32), ("dtype", C.c_int32)]


# name -> (restype, argtypes); every symbol include/lasr.h declares
SIGNATURES = {
    "lasr_version": (_i32, []),
    "lasr_roctx_range_push": (_i32, [C.c_char_p]),
    "lasr_roctx_range_pop": (_i32, []),
    "lasr_roctx_enabled": (_i32, []),
    "lasr_last_error": (C.c_char_p, []),
    "lasr_mel_num_frames": (_i64, [_i64]),
    "lasr_mel_workspace_bytes": (_sz, [_i64, _i64]),
    "lasr_mel_fwd": (_i32, [_p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "lasr_mel_fwd_src": (_i32, [_p, _p, _p, _i64, _i64, _i32, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "lasr_dither_noise": (_i32, [C.c_uint64, _p, _i64, _i64, _p, _p]),
    "lasr_spec_augment": (_i32, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "lasr_bct_to_btc": (_i32, [_p, _p, _i32, _i64, _i64, _i64, _p]),
    "lasr_btc_to_bct": (_i32, [_p, _i32, _p, _i64, _i64, _i64, _p]),
    "lasr_mask_lengths": (_i32, [_p, _i64, _i64, _p, _p]),
    "lasr_dwconv_fwd": (_i32, [_p, _p, _p, _p, _i32, _i64, _i64, _i64, _i32, _i32, _i32, _p]),
    "lasr_dwconv_wgrad_workspace_bytes": (_sz, [_i64, _i64, _i64, _i32]),
    "lasr_dwconv_wgrad": (_i32, [_p, _p, _p, _i32, _i64, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "lasr_gemm_workspace_bytes": (_sz, [_i64, _i64, _i32, _i32]),
    "lasr_gemm": (_i32, [_p, _p, _p, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _p, _p, _p, _i64, _p, _i32, _p, _sz, _p]),
    "lasr_gemm_batch_workspace_bytes": (_sz, [_p, _i32, _i32]),
    "lasr_gemm_batch": (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _sz, _p]),
    "lasr_bn_finalize": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _f32, _f32, _i32, _p]),
    "lasr_bn_eval_coef_many": (_i32, [_p, _i32, _f32, _p]),
    "lasr_bn_act_fwd": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i64, _i64, _i32, _p]),
    "lasr_bn_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lasr_bn_act_bwd_stats": (_i32, [_p] * 11 + [_i32, _i64, _i64, _i64, _i32, _p, _sz, _p]),
    "lasr_bn_act_bwd_apply": (_i32, [_p] * 20 + [_i32, _i64, _i64, _i64, _i32, _p, _sz, _p]),
    "lasr_bn_bwd_apply_workspace_bytes": (_sz, [_i64]),
    "lasr_reduce_many": (_i32, [_p, _i32, _p]),
    "lasr_gemm_batch_split_partials": (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _p, _sz, _p, _p, _p]),
    "lasr_gemm_multi_split_partials": (_i32, [_p, _i32, _i32, _p, _p, _p]),
    "lasr_dwconv_wgrad_partials": (_i32, [_p, _p, _i32, _i64, _i64, _i64, _i32, _i32, _p, _sz, _p, _p]),
    "lasr_dwconv_bwd_fused": (_i32, [_p, _p, _p, _p, _p, _i32, _i64, _i64, _i64, _i32, _p, _sz, _p, _p]),
    "lasr_gemm_batch_partials": (_i32, [_p, _i32, _i32, _i32, _i32, _i32, _p, _sz, _p, _p, _p]),
    "lasr_bn_finalize_partials": (_i32, [_p, _i32, _i64, _i64, C.c_float, C.c_float, _p]),
    "lasr_seqsum": (_i32, [_p, _i32, _i64, _i64, _i64, _p, _p]),
    "lasr_se_fwd": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p]),
    "lasr_se_bwd_workspace_bytes": (_sz, [_i64, _i64]),
    "lasr_se_bwd": (_i32, [_p] * 10 + [_i32, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _sz, _p]),
    "lasr_bilstm_saved_bytes": (_sz, [_i64, _i64]),
    "lasr_bilstm_fwd": (_i32, [_p] * 9 + [_i64, _i64, _p, _i32, _i64, _i64, _p, _p]),
    "lasr_bilstm_bwd_workspace_bytes": (_sz, [_i64]),
    "lasr_bilstm_bwd": (_i32, [_p, _i32, _i64, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lasr_copy_cols": (_i32, [_p, _i32, _i64, _i64, _p, _i32, _i64, _i64, _i64, _i64, _i32, _p]),
    "lasr_log_softmax": (_i32, [_p, _p, _p, _i64, _i64, _p]),
    "lasr_log_softmax_bwd": (_i32, [_p, _p, _p, _i64, _i64, _p]),
    "lasr_ctc_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lasr_ctc_loss": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _sz, _p]),
    "lasr_ctc_loss_mel": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _sz,
                                 _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _i32, _p, _p, _p, _sz, _p]),
    "lasr_greedy_decode": (_i32, [_p, _p, _i64, _i64, _i32, _p, _p, _p]),
    "lasr_novograd_workspace_bytes": (_sz, [_i64, _i64]),
    "lasr_novograd_step": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _p, _f32, _f32, _f32, _f32, _f32, _p, _sz, _p]),
    "lasr_novograd_step_keep": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _p, _f32, _f32, _f32, _f32, _f32, _p, _sz, _p]),
    "lasr_cast_f32_to_bf16": (_i32, [_p, _p, _i64, _p]),
    "lasr_cast_pad_f32_to_bf16": (_i32, [_p, _p, _i64, _i64, _i64, _p]),
    "lasr_gemm_ld": (_i32, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i64, _i64, _i64, _i32, _i32, _p, _i32, _p, _sz, _p]),
    "lasr_fold_bn_weights_many": (_i32, [_p, _i32, _p]),
    "lasr_gemm_dual": (_i32, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _p, _i64, _i32, _p]),
    "lasr_colsum_workspace_bytes": (_sz, [_i64, _i64]),
    "lasr_colsum_f32": (_i32, [_p, _p, _i64, _i64, _p, _sz, _p]),
    "lasr_scale_sum_f32": (_i32, [_p, _i64, _f32, _p, _p]),
    "lasr_edit_distance": (_i64, [_p, _i64, _p, _i64]),
    "lasr_prof_enable": (_i32, [_i32]),
    "lasr_prof_collect": (_i32, [_p, _p, _p, _p]),
    "lasr_prof_overhead_ms": (_i32, [_p, _i32, _p]),
    "lasr_model_create": (_i32, [C.POINTER(ModelConfig), C.POINTER(_p)]),
    "lasr_model_destroy": (None, [_p]),
    "lasr_model_set_prefetch": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _i32, _p, _p, _p, _sz]),
    "lasr_model_set_prefetch_src": (_i32, [_p, _p, _p, _p, _i64, _i64, _i32, _p, _i32, _p, _p, _p, _sz]),
    "lasr_model_clear_prefetch": (_i32, [_p]),
    "lasr_model_tensor_info": (_i64, [_p, _i64, C.c_char_p, _sz, C.POINTER(_i64), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int32), C.POINTER(_i64)]),
    "lasr_model_param_elems": (_i64, [_p]),
    "lasr_model_buffer_elems": (_i64, [_p]),
    "lasr_model_out_frames": (_i64, [_p, _i64]),
    "lasr_model_workspace_bytes": (_sz, [_p, _i64, _i64, _i64]),
    "lasr_model_tap": (_i64, [_p, C.c_char_p, _i64, _i64, _i64, C.POINTER(_i64)]),
    "lasr_model_forward": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    "lasr_model_backward": (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _p, _p, _sz, _p]),
    "lasr_model_loss_backward": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lasr_model_num_units": (_i64, [_p]),
    "lasr_model_unit_info": (_i32, [_p, _i64, C.c_char_p, _sz]),
    "lasr_model_loss_backward_partial": (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _sz, _i64, _p]),
    "lasr_model_backward_continue": (_i32, [_p, _p, _p, _i64, _i64, _p, _p, _sz, _i64, _p]),
    "lasr_mask_lengths_step": (_i32, [_p, _i64, _i64, _p, _p, _p]),
    "lasr_dropout_mask": (_i32, [_p, _i64, _p, _p]),
    "lasr_bn_act_fwd_drop": (_i32, [_p, _p, _p, _p, _p, _p, _i32, _i64, _i64, _i64, _i32, _p, _p]),
    "lasr_bn_act_bwd_stats_drop": (_i32, [_p] * 11 + [_i32, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "lasr_bn_act_bwd_apply_drop": (_i32, [_p] * 20 + [_i32, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "lasr_se_bwd_drop": (_i32, [_p] * 10 + [_i32, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p, _sz, _p]),
    "lasr_bn_se_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "lasr_bn_se_bwd": (_i32, [_p] * 26 + [_i32, _i64, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "lasr_model_set_dropout": (_i32, [_p, _f32, C.c_uint64, _p]),
    "lasr_lr_schedule_state_bytes": (_sz, []),
    "lasr_lr_schedule_init": (_i32, [_p, _sz, _i64, C.c_double, C.c_double, C.c_double, _i64, C.c_double, _i64, _i64, _i64, _i64]),
    "lasr_lr_schedule_step": (_i32, [_p, _p, _p]),
    "lasr_gemm_rowstat": (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _p, _p, _p, _p]),
    "lasr_gemm_rowstat_bytes": (_sz, [_i64, _i64]),
    "lasr_ctc_lean_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "lasr_ctc_loss_lean": (_i32, [_p, _i64, _p, _p, _i32, _p, _p, _p, _i64, _i64, _i64, _i64, _i32, _p, _p, _p, _p, _p, _p, _sz, _p]),
    "lasr_edit_distance_batch": (_i32, [_p, _p, _i64, _p, _p, _i64, _i64, _i32, _p, _p, _p, _p]),
    "lasr_step_metrics": (_i32, [_p, _p, _p, _i64, _p, _p]),
    "lasr_wav_info": (_i32, [C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lasr_wav_read_batch": (_i32, [_p, _i64, _p, C.c_double, _p, _i64, C.POINTER(_i64), _p, C.c_int32, _i32, _i32]),
    "lasr_comm_timing": (_i32, [_p, _i32]),
    "lasr_comm_timing_collect": (_i32, [_p, _i32, _p, _p, C.POINTER(_i32), _p, C.POINTER(_i32)]),
    "lasr_comm_unique_id": (_i32, [_p, _sz]),
    "lasr_comm_init": (_i32, [C.POINTER(_p), _p, _sz, _i32, _i32, _i32]),
    "lasr_comm_destroy": (_i32, [_p]),
    "lasr_comm_world": (_i32, [_p]),
    "lasr_comm_rank": (_i32, [_p]),
    "lasr_comm_allreduce": (_i32, [_p, _p, _i64, _p]),
    "lasr_comm_allreduce_ranges": (_i32, [_p, _p, _p, _p, _i32, _p]),
    "lasr_comm_broadcast": (_i32, [_p, _p, _i64, _i32, _p]),
    "lasr_comm_wait": (_i32, [_p, _p]),
}

_lib = None


class LasrError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load liblasr.so (built in-tree by ``make`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LasrError("liblasr.so not found at %s: build it with `make` (hipcc --offload-arch=gfx950); "
                        "there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().lasr_last_error()
        raise LasrError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))


def call(name: str, *args) -> None:
    """Call an int-returning entry point and raise on a non-zero status."""
    check(getattr(load(), name)(*args), name)
