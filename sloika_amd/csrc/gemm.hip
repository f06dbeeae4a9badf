// gemm.hip -- time-parallel contractions of the network on gfx950 fp32 MFMA:
//   y = act(x . W^T + b)     FeedForward.run sloika/layers.py:157-158, the input projections of Gru.step /
//                            Lstm.step (layers.py:1011, :678) hoisted over all T*B rows, Softmax.run's tensordot
//                            (layers.py:310)
//   row softmax              layers.py:311-314
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain), which is what the 1e-4 layer parity needs;
// there is no TF32-like path on gfx950.  Tile: 128 rows x (32*NT) columns per 256-thread workgroup, each of the
// 4 waves owns 32 rows x 32*NT columns (NT accumulators of 16 VGPRs), K staged through LDS in slabs of 32.
// LDS rows are padded to 36 floats so that the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots.
// The contraction index is permuted (lane half h takes k = 16h + s) so each lane reads 4 consecutive k per
// ds_read_b128; A and B use the same permutation, so the sum is unchanged.
#include "common.h"

#define GEMM_BM 128
#define GEMM_BK 32
#define GEMM_LDS_LD 36

// 4 consecutive k of one row, zero beyond K or when !ok.  `row` must point at a readable row.
template <bool ALIGNED>
__device__ __forceinline__ float4 load_row4(const float *row, int gk, int K, bool ok)
{
    float4 v;
    if (ALIGNED) {                       // K % 4 == 0: a float4 is entirely inside or entirely outside
        bool in = ok && gk < K;
        v = *reinterpret_cast<const float4 *>(row + (gk < K ? gk : 0));
        if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bool in = ok && gk + j < K;
            float t = row[gk + j < K ? gk + j : 0];
            e[j] = in ? t : 0.0f;
        }
        v = make_float4(e[0], e[1], e[2], e[3]);
    }
    return v;
}

// out-of-line so that the rarely used activations are not inlined 48 times into every epilogue
__device__ __noinline__ float gemm_act_generic(int act, float x) { return slk_act(act, x); }

// ACT: compile-time activation id for the common cases, -1 = runtime `act` through gemm_act_generic
template <int NT, bool ALIGNED, int ACT>
__global__ void __launch_bounds__(256) gemm_bias_act_kernel(const float *__restrict__ x, long ldx,
                                                            const float *__restrict__ W,
                                                            const float *__restrict__ bias, float *__restrict__ y,
                                                            long ldy, long M, int K, int N, int act, int ntile_n)
{
    constexpr int BN = 32 * NT;
    __shared__ __attribute__((aligned(16))) float xs[GEMM_BM * GEMM_LDS_LD];
    __shared__ __attribute__((aligned(16))) float ws[BN * GEMM_LDS_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long bid = blockIdx.x;
    const long m0 = (bid / ntile_n) * GEMM_BM;
    const int n0 = (int)(bid % ntile_n) * BN;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;

    for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
        // ---- stage x[128][32] and W[BN][32] slabs (zero filled outside M/N/K) ----
        // All global loads are issued back to back from clamped (always valid) addresses and masked afterwards:
        // no branches, so the compiler keeps every load in flight instead of waiting on each one in turn.
        constexpr int XL = (GEMM_BM * GEMM_BK / 4) / 256;               // float4 loads per thread for x
        constexpr int WL = (BN * GEMM_BK / 4 + 255) / 256;              // ... for W
        float4 xv[XL], wv[WL];
#pragma unroll
        for (int i = 0; i < XL; i++) {
            int idx = tid + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
            long gr = m0 + row;
            int gk = k0 + c4;
            xv[i] = load_row4<ALIGNED>(x + (gr < M ? gr : M - 1) * ldx, gk, K, gr < M);
        }
#pragma unroll
        for (int i = 0; i < WL; i++) {
            int idx = tid + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
            int gn = n0 + row, gk = k0 + c4;
            bool ok = idx < BN * GEMM_BK / 4 && gn < N;
            wv[i] = load_row4<ALIGNED>(W + (size_t)(gn < N ? gn : N - 1) * K, gk, K, ok);
        }
#pragma unroll
        for (int i = 0; i < XL; i++) {
            int idx = tid + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<float4 *>(&xs[row * GEMM_LDS_LD + c4]) = xv[i];
        }
#pragma unroll
        for (int i = 0; i < WL; i++) {
            int idx = tid + 256 * i, row = idx >> 3, c4 = (idx & 7) * 4;
            if (idx < BN * GEMM_BK / 4) *reinterpret_cast<float4 *>(&ws[row * GEMM_LDS_LD + c4]) = wv[i];
        }
        __syncthreads();
        // ---- 16 k-steps of 32x32x2 per accumulator ----
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 a4 = *reinterpret_cast<const float4 *>(&xs[(32 * wave + r) * GEMM_LDS_LD + h * 16 + 4 * q]);
            float4 b4[NT];
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
                b4[nt] = *reinterpret_cast<const float4 *>(&ws[(32 * nt + r) * GEMM_LDS_LD + h * 16 + 4 * q]);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4[nt].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4[nt].y, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4[nt].z, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4[nt].w, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*h][col = lane&31] ----
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        int col = n0 + 32 * nt + r;
        if (col >= N) continue;
        float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            long row = m0 + 32 * wave + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (row < M) {
                float v = acc[nt][reg] + bv;
                if constexpr (ACT >= 0) v = slk_act_t<ACT>(v);
                else v = gemm_act_generic(act, v);
                y[row * ldy + col] = v;
            }
        }
    }
}

template <int NT>
static int launch_gemm(const float *x, long ldx, const float *W, const float *bias, float *y, long ldy, long M, int K,
                       int N, int act, hipStream_t s)
{
    int ntile_n = (N + 32 * NT - 1) / (32 * NT);
    long ntile_m = (M + GEMM_BM - 1) / GEMM_BM;
    long blocks = ntile_m * ntile_n;
    if (blocks > 0x7fffffffL) return SLK_ERR_UNSUPPORTED;
    bool aligned = (ldx % 4 == 0) && (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
#define GEMM_LAUNCH(AL, AC)                                                                                   \
    hipLaunchKernelGGL((gemm_bias_act_kernel<NT, AL, AC>), dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, W, bias, y, \
                       ldy, M, K, N, act, ntile_n)
    if (aligned) {
        if (act == SLK_ACT_LINEAR) GEMM_LAUNCH(true, SLK_ACT_LINEAR);
        else if (act == SLK_ACT_TANH) GEMM_LAUNCH(true, SLK_ACT_TANH);
        else GEMM_LAUNCH(true, -1);
    } else {
        if (act == SLK_ACT_LINEAR) GEMM_LAUNCH(false, SLK_ACT_LINEAR);
        else if (act == SLK_ACT_TANH) GEMM_LAUNCH(false, SLK_ACT_TANH);
        else GEMM_LAUNCH(false, -1);
    }
#undef GEMM_LAUNCH
    return slk_launch_status();
}

extern "C" int slk_gemm_bias_act_f32(const float *x, long ldx, const float *W, const float *bias, float *y, long ldy,
                                     long M, int K, int N, int act, slk_stream_t stream)
{
    if (!x || !W || !y || M < 0 || K < 1 || N < 1 || ldx < K || ldy < N || !slk_act_valid(act)) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    // pick the column-tile count that wastes the fewest MFMA columns (ties -> wider tile)
    int best = 3, best_cost = 1 << 30;
    for (int nt = 3; nt >= 1; nt--) {
        int cost = ((N + 32 * nt - 1) / (32 * nt)) * nt;
        if (cost < best_cost) { best_cost = cost; best = nt; }
    }
    hipStream_t s = slk_stream(stream);
    switch (best) {
    case 1: return launch_gemm<1>(x, ldx, W, bias, y, ldy, M, K, N, act, s);
    case 2: return launch_gemm<2>(x, ldx, W, bias, y, ldy, M, K, N, act, s);
    default: return launch_gemm<3>(x, ldx, W, bias, y, ldy, M, K, N, act, s);
    }
}

// ------------------------------------------------------------------------------------------------------
// Row softmax, one wave per row, the row held in registers (N <= 64*MAXE) or re-read (generic).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One code path for the row statistics so that the posterior written by softmax_rows and the one the decoder rebuilds
// from (logit, max, 1/sum) are bit-identical:  p = expf(l - m) * (1 / s)   with s summed in lane-strided order.
template <int MAXE, bool STATS_ONLY>
__global__ void __launch_bounds__(256) softmax_rows_kernel(float *__restrict__ y, long M, int N,
                                                           float2 *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    for (long row = wave0; row < M; row += nwave) {
        float *p = y + row * N;
        float v[MAXE];
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            int j = lane + 64 * e;
            v[e] = j < N ? p[j] : -INFINITY;
            m = fmaxf(m, v[e]);
        }
        m = wave_max(m);
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < MAXE; e++) {
            v[e] = __expf(v[e] - m);
            s += v[e];
        }
        s = wave_sum(s);
        const float r = 1.0f / s;
        if (STATS_ONLY) {
            if (lane == 0) stats[row] = make_float2(m, r);
        } else {
#pragma unroll
            for (int e = 0; e < MAXE; e++) {
                int j = lane + 64 * e;
                if (j < N) p[j] = v[e] * r;
            }
        }
    }
}

template <bool STATS_ONLY>
__global__ void __launch_bounds__(256) softmax_rows_generic_kernel(float *__restrict__ y, long M, int N,
                                                                   float2 *__restrict__ stats)
{
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    for (long row = wave0; row < M; row += nwave) {
        float *p = y + row * N;
        float m = -INFINITY;
        for (int j = lane; j < N; j += 64) m = fmaxf(m, p[j]);
        m = wave_max(m);
        float s = 0.0f;
        for (int j = lane; j < N; j += 64) s += __expf(p[j] - m);
        s = wave_sum(s);
        const float r = 1.0f / s;
        if (STATS_ONLY) {
            if (lane == 0) stats[row] = make_float2(m, r);
        } else {
            for (int j = lane; j < N; j += 64) p[j] = __expf(p[j] - m) * r;
        }
    }
}

template <bool STATS_ONLY>
static int launch_softmax(float *y, long M, int N, float2 *stats, hipStream_t s)
{
    long blocks = (M + 3) / 4;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (N <= 64 * 4)
        hipLaunchKernelGGL((softmax_rows_kernel<4, STATS_ONLY>), dim3((unsigned)blocks), dim3(256), 0, s, y, M, N, stats);
    else if (N <= 64 * 17)
        hipLaunchKernelGGL((softmax_rows_kernel<17, STATS_ONLY>), dim3((unsigned)blocks), dim3(256), 0, s, y, M, N, stats);
    else
        hipLaunchKernelGGL((softmax_rows_generic_kernel<STATS_ONLY>), dim3((unsigned)blocks), dim3(256), 0, s, y, M, N, stats);
    return slk_launch_status();
}

extern "C" int slk_softmax_rows_f32(float *y, long M, int N, slk_stream_t stream)
{
    if (!y || M < 0 || N < 1) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    return launch_softmax<false>(y, M, N, nullptr, slk_stream(stream));
}

extern "C" int slk_softmax_rowstats_f32(const float *logits, long M, int N, float *stats, slk_stream_t stream)
{
    if (!logits || !stats || M < 0 || N < 1) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    return launch_softmax<true>(const_cast<float *>(logits), M, N, reinterpret_cast<float2 *>(stats), slk_stream(stream));
}

extern "C" int slk_linear_softmax_f32(const float *x, long ldx, const float *W, const float *bias, float *y, long M,
                                      int K, int N, slk_stream_t stream)
{
    int rc = slk_gemm_bias_act_f32(x, ldx, W, bias, y, N, M, K, N, SLK_ACT_LINEAR, stream);
    if (rc != SLK_OK) return rc;
    return slk_softmax_rows_f32(y, M, N, stream);
}
