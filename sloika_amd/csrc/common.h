// common.h -- shared device helpers for the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sloika_amd.h"

#define SLK_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int slk_launch_status()
{
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SLK_OK : SLK_ERR_LAUNCH;
}

// A value derived from per-device state (function attributes set with hipFuncSetAttribute, the CU count ...), evaluated once per
// HIP device of the calling thread instead of once per process: a process may drive several GPUs.
#define SLK_MAX_DEVICES 64
#define SLK_PER_DEVICE(TYPE, EXPR)                                                                 \
    ([&]() -> TYPE {                                                                               \
        static TYPE v_[SLK_MAX_DEVICES];                                                           \
        static unsigned char ok_[SLK_MAX_DEVICES];                                                 \
        int d_ = 0;                                                                                \
        if (hipGetDevice(&d_) != hipSuccess || d_ < 0 || d_ >= SLK_MAX_DEVICES) return (EXPR);     \
        if (!ok_[d_]) {                                                                            \
            v_[d_] = (EXPR);                                                                       \
            ok_[d_] = 1;                                                                           \
        }                                                                                          \
        return v_[d_];                                                                             \
    }())

static inline hipStream_t slk_stream(slk_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---- activations: sloika/activation.py:8-115 ------------------------------------------------------------
__device__ __forceinline__ float slk_clip(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence: these sit on the recurrent critical path
__device__ __forceinline__ float slk_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float slk_sigmoid(float x) { return slk_rcp(1.0f + __expf(-x)); }

// tanh through one exp: tanh(x) = 1 - 2/(exp(2x)+1); abs error < 3e-7 over the whole range.
// Five instructions (mul, exp, add, rcp, fma).  The values are those of  1 - 2 * rcp(__expf(2x) + 1)  bit for bit: doubling is
// exact, so x * (2 log2 e) rounds like (2x) * log2 e and fma(-2, r, 1) like 1 - 2r.
__device__ __forceinline__ float slk_tanh(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.885390043258667f);       // 2 * 0x3fb8aa3b
    return fmaf(-2.0f, slk_rcp(e + 1.0f), 1.0f);
}

// elu's negative branch (activation.py:52-57 uses T.expm1): exp(x) - 1 through v_exp_f32 has an ABSOLUTE error of one ulp of
// 1.0 (6e-8), i.e. no relative accuracy left for tiny |x| -- and training multiplies by y + 1 and feeds small gradients through
// it.  Above -2^-6 the series x + x^2 (1/2 + x/6) is exact to float32 (the next term is < 2e-7 of x) and below it the
// hardware form is within 4e-6 relative; expm1f itself costs ~25 instructions per value.
// Both forms are evaluated and one selected (the same values as branching between them: v_exp_f32 of x log2 e is what __expf
// compiles to): with branches a wave pays three exec-mask updates on the scalar unit per value, ~8 cycles each.
__device__ __forceinline__ float slk_elu(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 1.4426950216293335f) - 1.0f;       // 0x3fb8aa3b
    const float ser = fmaf(x * x, fmaf(x, 0.16666667f, 0.5f), x);
    const float neg = x > -0.015625f ? ser : e;
    return x > 0.0f ? x : neg;
}

template <int ACT>
__device__ __forceinline__ float slk_act_t(float x)
{
    if constexpr (ACT == SLK_ACT_LINEAR) return x;
    else if constexpr (ACT == SLK_ACT_TANH) return slk_tanh(x);
    else if constexpr (ACT == SLK_ACT_SIGMOID) return slk_sigmoid(x);
    else if constexpr (ACT == SLK_ACT_ELU) return slk_elu(x);
    else if constexpr (ACT == SLK_ACT_RELU) return fmaxf(x, 0.0f);
    else return x;
}

__device__ __forceinline__ float slk_act(int act, float x)
{
    switch (act) {
    case SLK_ACT_LINEAR: return x;
    case SLK_ACT_TANH: return slk_tanh(x);
    case SLK_ACT_SIGMOID: return slk_sigmoid(x);
    case SLK_ACT_ELU: return slk_elu(x);
    case SLK_ACT_RELU: return fmaxf(x, 0.0f);
    case SLK_ACT_RELU_SMOOTH: {
        float y = slk_clip(x, 0.0f, 1.0f);
        return y * y - 2.0f * y + x + fabsf(x);
    }
    case SLK_ACT_SOFTPLUS: return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
    case SLK_ACT_EXP: return expf(x);
    case SLK_ACT_ERF: return erff(x);
    case SLK_ACT_L1ML2: return x / sqrtf(1.0f + 0.5f * x * x);
    case SLK_ACT_FAIR: return x / (1.0f + fabsf(x) / 1.3998f);
    case SLK_ACT_RETU: return slk_tanh(fmaxf(x, 0.0f));
    case SLK_ACT_TANH_PM: return slk_clip(x, -1.0f, 1.0f);
    case SLK_ACT_SIGMOID_PM: return slk_clip(0.5f + 0.25f * x, 0.0f, 1.0f);
    case SLK_ACT_BOUNDED_LINEAR: return slk_clip(x, -1.0f, 1.0f);
    case SLK_ACT_SIN: return sinf(x);
    case SLK_ACT_CAUCHY: { float u = x / 2.3849f; return x / (1.0f + u * u); }
    case SLK_ACT_GEMAN_MCCLURE: { float u = 1.0f + x * x; return x / (u * u); }
    case SLK_ACT_WELSH: { float u = x / 2.9846f; return x * expf(-(u * u)); }
    default: return x;
    }
}

static inline bool slk_act_valid(int act) { return act >= 0 && act < SLK_ACT_COUNT; }
