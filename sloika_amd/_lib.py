"""ctypes binding of libsloika_amd.so (the C ABI declared in include/sloika_amd.h).

There is NO CPU fallback: if the HIP library is missing or no GPU is visible, every compute entry point
raises.  (The library itself loads fine on a CPU-only box, which is what the symbol-export test uses.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
#: SLOIKA_AMD_LIB: another build of the same C ABI (tools/build_diag_lib.sh: the diagnostic build the scripts under tools/ use)
LIB_PATH = os.environ.get("SLOIKA_AMD_LIB") or os.path.join(_HERE, "_build", "libsloika_amd.so")

SLK_OK = 0
SLK_ERR_INVALID_ARG = -1
SLK_ERR_UNSUPPORTED = -2
SLK_ERR_LAUNCH = -3
SLK_ERR_WORKSPACE = -4
SLK_ERR_NO_DEVICE = -5

POST_RAW, POST_PLAIN, POST_LOG, POST_LN = 0, 1, 2, 3

_vp, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/sloika_amd.h one to one
PROTOTYPES = {
    "slk_abi_version": (_i, []),
    "slk_error_string": (C.c_char_p, [_i]),
    "slk_device_count": (_i, []),
    "slk_clock_probe": (_i, [_vp, _i, _vp]),
    "slk_selftest_mfma4_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "slk_med_mad_normalise_f32": (_i, [_vp, _i, _i, _vp, _l, _l, _vp, _vp, _vp]),
    "slk_med_mad_normalise_ragged_f32": (_i, [_vp, _i, _l, _vp, _vp, _l, _l, _vp, _vp, _vp]),
    "slk_window_std_f32": (_i, [_vp, _i, _i, _vp, _vp]),
    "slk_conv1d_out_len": (_i, [_i, _i, _i, _i, _i]),
    "slk_conv1d_f32": (_i, [_vp, _l, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_window_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "slk_gemm_bias_act_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _l, _i, _i, _i, _vp]),
    "slk_linear_softmax_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "slk_linear_rowstats_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _l, _i, _i, _vp, _vp]),
    "slk_pack_reads_f32": (_i, [_vp, _vp, _vp, _i, _vp, _l, _vp]),
    "slk_reads_nonfinite_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "slk_open_pore_trim_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "slk_split_f16x2_f32": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "slk_linear_rowstats_f16x3": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _vp, _vp]),
    "slk_pack_bf16x3_bytes": (_sz, [_i, _i]),
    "slk_pack_bf16x3_f32": (_i, [_vp, _i, _i, _vp, _vp]),
    "slk_gemm_bias_act_bf16x6": (_i, [_vp, _l, _vp, _vp, _vp, _l, _l, _i, _i, _i, _vp]),
    "slk_gemm_dact_bf16x6": (_i, [_vp, _l, _vp, _vp, _l, _i, _vp, _l, _l, _i, _i, _vp]),
    "slk_gemm_bias_act_f16x3": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _vp]),
    "slk_softmax_from_stats_f32": (_i, [_vp, _l, _vp, _vp, _l, _l, _i, _vp]),
    "slk_softmax_rows_f32": (_i, [_vp, _l, _i, _vp]),
    "slk_softmax_rowstats_f32": (_i, [_vp, _l, _i, _vp, _vp]),
    "slk_viterbi_kmer_logits_f32": (_i, [_vp, _l, _vp, _i, _i, _i, _i, _f, _f, _vp, _sz, _vp, _vp, _vp, _vp]),
    "slk_viterbi_kmer_logits_ragged_f32": (_i, [_vp, _l, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "slk_softmax_viterbi_pack_bytes": (_sz, [_i, _i, _i]),
    "slk_softmax_viterbi_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "slk_softmax_viterbi_pack_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "slk_softmax_viterbi_f32": (_i, [_vp, _l, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "slk_gru_recurrent_ragged_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_gru_scan16_f32": (_i, [_vp, _l, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_log_post_logits_f32": (_i, [_vp, _l, _vp, _vp, _sz, _i, _f, _vp]),
    "slk_gru_recurrent_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_gru_recurrent_f32_ex": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_gru_workspace_bytes": (_sz, [_i, _i, _i]),
    "slk_gru_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slk_paths_to_bases": (_i, [_vp, _l, _vp, _i, _i, _i, _i, C.c_ulonglong, _vp, _l, _vp, _vp]),
    "slk_gru_bar16_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "slk_lstm_recurrent_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_lstm_recurrent_ragged_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_lstm_workspace_bytes": (_sz, [_i, _i, _i]),
    "slk_lstm_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "slk_viterbi_kmer_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "slk_viterbi_kmer_f32": (_i, [_vp, _i, _i, _i, _i, _f, _i, _f, _vp, _sz, _vp, _vp, _vp, _vp]),
    "slk_log_post_f32": (_i, [_vp, _vp, _sz, _i, _f, _vp]),
    "slk_prepare_post_f32": (_i, [_vp, _vp, _sz, _f, _vp]),
    "slk_argmax_decode_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "slk_slip_update_f32": (_i, [_vp, _i, _f, _vp, _vp, _vp]),
    "slk_map_to_sequence_workspace_bytes": (_sz, [_i, _i]),
    "slk_map_to_sequence_f32": (_i, [_vp, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "slk_map_to_sequence_batch_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "slk_kmer_labels_i32": (_i, [_vp, _l, _i, _i, C.c_char_p, _i, _i, _vp, _vp, _vp]),
    "slk_raw_chunk_labels_workspace_bytes": (_sz, [_l]),
    "slk_raw_chunk_labels_i32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _l, _i, _i, _vp, _sz, _vp, _vp]),
    "slk_raw_chunk_labels_interp_i32": (_i, [_vp, _vp, _vp, _l, _i, _i, _l, _vp, _l, _i, C.c_char_p, _i, _l, _i, _vp, _i, _vp, _vp,
                                             _vp, _vp]),
    "slk_lstm_fused16_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_lstm_scan16_f32": (_i, [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_activation_f32": (_i, [_vp, _vp, _sz, _i, _vp]),
    "slk_train_pack_xh_f32": (_i, [_vp, _l, _vp, _l, _vp, _i, _i, _i, _i, _i, _vp]),
    "slk_train_pack_xrh_f32": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "slk_gru_backward_f32": (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_gru_backward16_f32": (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_gru_backward16_dx_f32": (_i, [_vp, _l, _vp, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _l, _i,
                                       _vp]),
    "slk_lstm_gates_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "slk_lstm_backward_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_lstm_backward16_f32": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "slk_softmax_xent_grad_f32": (_i, [_vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "slk_linear_xent_grad_f16x3": (_i, [_vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "slk_reduce_sum_f32": (_i, [_vp, _sz, _i, _vp, _vp]),
    "slk_reduce_rows_sum_f32": (_i, [_vp, _i, _sz, _vp, _vp, _vp]),
    "slk_gemm_tn_workspace_bytes": (_sz, [_l, _i, _i]),
    "slk_gemm_tn_f32": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _i, _i, _vp, _vp, _sz, _vp]),
    "slk_gemm_tn_bf16x6_f32": (_i, [_vp, _l, _vp, _l, _vp, _l, _l, _i, _i, _vp, _vp, _sz, _vp]),
    "slk_gemm_tn_multi_workspace_bytes": (_sz, [_l, _i, _vp, _vp]),
    "slk_gemm_tn_multi_bf16x6_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _vp, _sz, _vp]),
    "slk_act_backward_f32": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "slk_add_inplace_f32": (_i, [_vp, _vp, _sz, _vp]),
    "slk_train_im2col_cin1_f32": (_i, [_vp, _l, _l, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_train_im2col_f32": (_i, [_vp, _l, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "slk_train_col2im_f32": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _l, _vp]),
    "slk_adamski_update_f32": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _f, _f, _vp]),
    "slk_sgd_update_f32": (_i, [_vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _vp]),
}

_lib = None


class SloikaAmdError(RuntimeError):
    pass


def lib():
    """Load the HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SloikaAmdError(
                "sloika_amd: HIP extension %s is missing. Build it with `python -m sloika_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # torch first: it ships its own HIP runtime, and if this library's dependency on libamdhip64 is resolved before
        # torch has loaded its copy, torch.cuda.is_available() turns False for the rest of the process (seen on the GPU
        # box when a script touched the library before importing torch)
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.slk_abi_version() != 1:
            raise SloikaAmdError("sloika_amd: ABI version mismatch")
        _lib = handle
    return _lib


def error_string(code):
    return lib().slk_error_string(code).decode()


def check(rc, what=""):
    if rc == SLK_OK:
        return
    msg = "%s: %s (code %d)" % (what or "sloika_amd", error_string(rc), rc)
    if rc == SLK_ERR_INVALID_ARG:
        raise ValueError(msg)            # the reference asserts on these (layers.py:152-155, decode.py:50-52)
    raise SloikaAmdError(msg)


def require_gpu():
    """Raise unless a HIP device is usable through both the library and torch."""
    import torch
    if lib().slk_device_count() < 1 or not torch.cuda.is_available():
        raise SloikaAmdError("sloika_amd: no AMD GPU visible; the basecalling path only runs on the HIP kernels "
                             "(no CPU fallback)")
    return True
