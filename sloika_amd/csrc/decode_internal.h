// decode_internal.h -- pieces of decode.hip that other translation units of the library launch (not part of the C ABI).
#pragma once
#include "common.h"

// decode.py:84-91 on the packed 16-bit traceback of the nbase-4 forward kernels (viterbi_forward4_kernel's format):
// paths left aligned and -1 padded in path_out[B][T], lengths in len_out[B]; lens = per-chunk step counts or NULL.
__attribute__((visibility("hidden"))) int slk_backtrace_packed4(const uint8_t *tb, const int32_t *best, int T, int B, int nkmer,
                                                                int32_t *path_out, int32_t *len_out, const int *lens,
                                                                hipStream_t s);

// ... on the one-byte-per-four-states traceback of the fused kernel (softmax_viterbi.hip; decode.hip: viterbi_backtrace_kernel FMT 2)
__attribute__((visibility("hidden"))) int slk_backtrace_packed8(const uint8_t *tb, const int32_t *best, int T, int B, int nkmer,
                                                                int32_t *path_out, int32_t *len_out, const int *lens,
                                                                hipStream_t s);
