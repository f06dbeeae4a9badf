// f16split.h -- the pieces shared by the kernels that compute float32 products as three fp16 MFMA terms
// (gru_fused16.hip, gru_bar16.hip): v = hi + lo with both halves fp16, x.w ~ hi.lo + lo.hi + hi.hi in float32
// accumulators, operands scaled row-wise by powers of two so that any finite float32 value is in range.
#pragma once
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void keep(half8 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keepf(float &v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ float sel4(const f32x4 &a, int q)
{
    const float lo = (q & 1) ? a[1] : a[0];
    const float hi = (q & 1) ? a[3] : a[2];
    return (q & 2) ? hi : lo;
}

// two float32 -> one dword of fp16 "hi" parts and one of fp16 "lo" parts (v = hi + lo): the hi pair is one packed conversion, each
// lo = v - float(hi) one v_fma_mix_f32 that reads its half of the hi pair as fp16 ((-1) * hi + v: exact, the difference of a float32
// and its own 11-bit rounding is representable), the lo pair another packed conversion -- four instructions where the plain C form
// compiles to eight.  (It is asm for a second reason: from C hipcc folded the multiply or add that PRODUCED a value into the
// conversion of the value it subtracts -- v_fma_mixlo_f16, ONE rounding -- while the stored hi part was the conversion of the rounded
// float32; on an exact fp16 tie, 1 value in 8192, hi + lo was then off by an fp16 ulp.  Found by tools/experiments/f16_error_probe4.py.)
__device__ __forceinline__ void split2(float a, float b, unsigned &hi, unsigned &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %1, %2, %3"
        : "=&v"(hi), "=&v"(lo), "+v"(a), "+v"(b));
}
// The same into registers the caller keeps for it from step to step (`hi` and `lo` go in as well as out, so they never share a
// register with anything else): behind an asm statement hipcc pads for every MFMA that wrote the registers of the statement's
// outputs in the last few instructions -- it counts the statement as ONE wait state -- and fresh outputs tend to land in the
// registers of accumulators that have just died.
__device__ __forceinline__ void split2_kept(float a, float b, unsigned &hi, unsigned &lo)
{
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_cvt_pk_f16_f32 %1, %2, %3"
        : "+v"(hi), "+v"(lo), "+v"(a), "+v"(b));
}

// Power-of-two scale that brings a row whose largest magnitude is `amax` into [1, 2): scale = 2^(127 - e), inv = 2^(e - 127),
// e = biased exponent of amax kept inside [27, 227] so that both are normal numbers.  Multiplying by either is exact.
__device__ __forceinline__ float pow2_scale(float amax, float &inv)
{
    const int e = min(max((int)((__float_as_uint(amax) >> 23) & 0xff), 27), 227);
    inv = __uint_as_float((unsigned)e << 23);
    return __uint_as_float((unsigned)(254 - e) << 23);
}
// maximum over the four k groups of an MFMA operand row (lanes m, m+16, m+32, m+48).  gfx950's row and half swaps hand every
// lane its partner's value in one VALU instruction each (v_permlane16_swap: rows 1, 3 of the first operand <-> rows 0, 2 of
// the second; v_permlane32_swap: upper half <-> lower half); __shfl_xor would be two LDS round trips.
__device__ __forceinline__ float kgroup_max(float v)
{
    const unsigned u = __float_as_uint(v);
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned m = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// acc += W . h as a 3-term split; small terms first so that they are not absorbed by the large one
template <int ABL = 0>
__device__ __forceinline__ f32x4 mfma3(const half8 &w_hi, const half8 &w_lo, const half8 &h_hi, const half8 &h_lo, f32x4 acc)
{
    if constexpr (ABL & 1) {
        half8 a = w_hi, b = h_hi;
        asm volatile("" : "+v"(a), "+v"(b), "+v"(acc));
        return acc;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi, h_lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo, h_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi, h_hi, acc, 0, 0, 0);
    return acc;
}

