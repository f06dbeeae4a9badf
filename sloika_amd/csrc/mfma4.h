// mfma4.h -- helpers shared by the recurrent kernels built on v_mfma_f32_4x4x1_16b_f32 (see recurrent.hip).
#pragma once
#include <utility>

#include "common.h"

template <int CB, int AB>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, CB, AB, 0);
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

// Four accumulators in rotation: a 4x4x1 MFMA occupies the pipe for 8 cycles but its result is only readable as
// the next SrcC a few wait states later; with 4 independent chains no s_nop padding is needed.
template <int CB, int G, int... Is>
__device__ __forceinline__ void mfma_chain(const float *hp, const float *w, f32x4 (&acc)[4],
                                           std::integer_sequence<int, Is...>)
{
    ((acc[Is & 3] = mfma4<CB, Is % G>(hp[Is / G], w[Is], acc[Is & 3])), ...);
}
// two-accumulator form for register-starved instantiations (more waves per SIMD hide the dependent-issue gaps)
template <int CB, int G, int... Is>
__device__ __forceinline__ void mfma_chain2(const float *hp, const float *w, f32x4 (&acc)[2],
                                            std::integer_sequence<int, Is...>)
{
    ((acc[Is & 1] = mfma4<CB, Is % G>(hp[Is / G], w[Is], acc[Is & 1])), ...);
}

// Same, for the MFMAs OFF .. OFF+len(Is)-1 of a chain (lets a chain be issued in two parts).
template <int CB, int G, int OFF, int... Is>
__device__ __forceinline__ void mfma_chain_range(const float *hp, const float *w, f32x4 (&acc)[4],
                                                 std::integer_sequence<int, Is...>)
{
    ((acc[(OFF + Is) & 3] = mfma4<CB, (OFF + Is) % G>(hp[(OFF + Is) / G], w[OFF + Is], acc[(OFF + Is) & 3])), ...);
}

// Workgroup barrier for the LDS state exchange that does NOT drain global memory traffic: __syncthreads() would
// emit s_waitcnt vmcnt(0) and stall every step on the h_out stores and on the vI prefetch issued for the next step.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// K-slice sums without LDS round trips: lane l (+)= lane l^32 / l^16 with the gfx950 row-swap instructions,
// lane l (+)= lane l^8 with a DPP rotate inside the 16-lane row.  Every lane ends up with the full sum.
__device__ __forceinline__ float xor32_sum(float v)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_sum(float v)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor8_sum(float v)
{
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}

template <int S>
__device__ __forceinline__ f32x4 sum_slices(f32x4 v)
{
    if constexpr (S >= 8) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = xor8_sum(v[i]);
    }
    if constexpr (S >= 4) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = xor16_sum(v[i]);
    }
    if constexpr (S >= 2) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = xor32_sum(v[i]);
    }
    return v;
}

template <int ACT>
__device__ __forceinline__ float act_sel(int act, float x)
{
    if constexpr (ACT >= 0) return slk_act_t<ACT>(x);
    else return slk_act(act, x);
}

