// gemm_rows.hip -- "x-stationary" projection for wide outputs (the Softmax layer's 96 -> 1025 tensordot,
// sloika/layers.py:310-313) on gfx950 fp32 MFMA, with the softmax row statistics fused in.
//
// A 512-thread workgroup owns 128 rows for ALL N columns:
//   * each wave keeps its 32 rows of x as MFMA A-fragments in registers for the whole kernel (K/2 VGPRs), so x is
//     read from HBM exactly once (the tiled kernel in gemm.hip re-stages it once per column tile: 11x for N=1025);
//   * the weight matrix streams through LDS in 64-column tiles, double buffered (global -> registers -> LDS), one
//     s_barrier per tile; waves are arranged 4 (rows) x 2 (columns): one 32x32 accumulator each;
//   * logits are written with a row stride `ldy` (>= N); a stride that is a multiple of 32 floats makes every store
//     a full 128-byte line;
//   * optional per-row statistics (max, 1/sum exp(l - max)) are accumulated online per lane while the tiles go by and
//     reduced once at the end, so neither a softmax pass nor a statistics pass over the 3.4 GB logits is needed.
#include "common.h"

#define GR_BM 128
#define GR_BN 64
#define GR_LD 132   /* LDS row stride of a weight tile in floats: K <= 128 plus padding, 16-byte aligned, odd*4 */

template <int AREG, bool STATS>
__global__ void __launch_bounds__(512, 2) gemm_rows_kernel(const float *__restrict__ x, long ldx,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           float *__restrict__ y, long ldy, long M, int K, int N,
                                                           float2 *__restrict__ stats)
{
    // K <= 2*AREG.  Contraction index permuted: lane half h owns k in [h*AREG, (h+1)*AREG) (zero padded past K).
    __shared__ __attribute__((aligned(16))) float ws[2][GR_BN * GR_LD];
    __shared__ float2 red[2][GR_BM];                       // per column-half partial (m, s) of each row
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const long m0 = (long)blockIdx.x * GR_BM;
    const int ntiles = (N + GR_BN - 1) / GR_BN;

    // ---- A fragments: x[m0 + 32*wm + r][h*AREG + s], s = 0..AREG-1 ----
    float a[AREG];
    {
        long row = m0 + 32 * wm + r;
        const float *xr = x + (row < M ? row : M - 1) * ldx;
#pragma unroll
        for (int s = 0; s < AREG; s++) {
            int k = h * AREG + s;
            float v = xr[k < K ? k : 0];
            a[s] = (row < M && k < K) ? v : 0.0f;
        }
    }
    // ---- W tile staging: 64 rows x (2*AREG) floats, as float4 pieces; thread -> (row, piece) ----
    constexpr int KP = 2 * AREG;                           // padded K
    constexpr int PIECES = GR_BN * KP / 4;                 // float4 per tile
    constexpr int NLD = (PIECES + 511) / 512;
    const bool w_vec = (K % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    float4 wreg[NLD];
    int st_c[NLD], st_k4[NLD];                             // this thread's (tile row, first k) per staged float4
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int idx = tid + 512 * j;
        st_c[j] = idx / (KP / 4);
        st_k4[j] = (idx % (KP / 4)) * 4;
    }
    auto load_tile = [&](int nt) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int idx = tid + 512 * j;
            const int c = st_c[j], k4 = st_k4[j];
            const int gn = nt * GR_BN + c;
            const bool okc = idx < PIECES && gn < N;
            const float *p = W + (size_t)(gn < N ? gn : N - 1) * K;
            float4 v;
            if (w_vec) {
                v = *reinterpret_cast<const float4 *>(p + (k4 < K ? k4 : 0));
                if (!(okc && k4 < K)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float t = p[k4 + q < K ? k4 + q : 0];
                    e[q] = (okc && k4 + q < K) ? t : 0.0f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            wreg[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int idx = tid + 512 * j;
            if (idx < PIECES) *reinterpret_cast<float4 *>(&ws[buf][st_c[j] * GR_LD + st_k4[j]]) = wreg[j];
        }
    };

    // store addressing: wave-uniform 64-bit base (row block, tile column) + per-lane 32-bit offsets fixed for the kernel
    const int wm_u = __builtin_amdgcn_readfirstlane(wm), wn_u = __builtin_amdgcn_readfirstlane(wn);
    unsigned voff[16];
#pragma unroll
    for (int reg = 0; reg < 16; reg++) voff[reg] = (unsigned)(((reg & 3) + 8 * (reg >> 2) + 4 * h) * ldy + r);
    float *const ywave = y + (m0 + 32 * wm_u) * ldy + 32 * wn_u;
    const long rows_left = M - (m0 + 32 * wm_u);           // rows of this wave's block that exist

    float rm[16], rs[16];                                  // online softmax state of this lane's columns, per row reg
#pragma unroll
    for (int i = 0; i < 16; i++) { rm[i] = -INFINITY; rs[i] = 0.0f; }

    load_tile(0);
    store_tile(0);
    if (ntiles > 1) load_tile(1);
    __syncthreads();

    for (int nt = 0; nt < ntiles; nt++) {
        const float *wt = &ws[nt & 1][(32 * wn + r) * GR_LD + h * AREG];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
        for (int q = 0; q < AREG / 4; q++) {
            float4 b4 = *reinterpret_cast<const float4 *>(wt + 4 * q);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 0], b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], b4.w, acc, 0, 0, 0);
        }
        // the other buffer is free (everyone finished tile nt-1 before the barrier that ended it): refill it
        if (nt + 1 < ntiles) store_tile((nt + 1) & 1);
        if (nt + 2 < ntiles) load_tile(nt + 2);
        // ---- epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*h][col = r] ----
        const int col = nt * GR_BN + 32 * wn_u + r;
        const bool colok = col < N;
        const float bv = (bias && colok) ? bias[col] : 0.0f;
        float *const ytile = ywave + nt * GR_BN;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int rloc = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float v = acc[reg] + bv;
            if (colok && rloc < rows_left) ytile[voff[reg]] = v;
            if (STATS && colok) {
                // online softmax with ONE exponential per element: the larger of (running max, v) keeps weight 1
                const float e = __expf(-fabsf(v - rm[reg]));
                rs[reg] = (v <= rm[reg]) ? rs[reg] + e : rs[reg] * e + 1.0f;
                rm[reg] = fmaxf(rm[reg], v);
            }
        }
        __syncthreads();
    }
    if (STATS) {
        // combine the 32 lanes that share a row (same h), then the two column halves (wn) through LDS
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            float m = rm[reg], s = rs[reg];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const float om = __shfl_xor(m, o), os = __shfl_xor(s, o);
                const float mn = fmaxf(m, om);
                // exp(-inf - -inf) would be NaN: a side that has seen no column contributes nothing
                const float sa = (m == -INFINITY) ? 0.0f : s * __expf(m - mn);
                const float sb = (om == -INFINITY) ? 0.0f : os * __expf(om - mn);
                s = sa + sb;
                m = mn;
            }
            if (r == 0) red[wn][32 * wm + (reg & 3) + 8 * (reg >> 2) + 4 * h] = make_float2(m, s);
        }
        __syncthreads();
        if (tid < GR_BM && m0 + tid < M) {
            const float2 p0 = red[0][tid], p1 = red[1][tid];
            const float mn = fmaxf(p0.x, p1.x);
            const float s = ((p0.x == -INFINITY) ? 0.0f : p0.y * __expf(p0.x - mn)) +
                            ((p1.x == -INFINITY) ? 0.0f : p1.y * __expf(p1.x - mn));
            stats[m0 + tid] = make_float2(mn, 1.0f / s);
        }
    }
}

template <bool STATS>
static int launch_rows(const float *x, long ldx, const float *W, const float *bias, float *y, long ldy, long M, int K,
                       int N, float2 *stats, hipStream_t s)
{
    long blocks = (M + GR_BM - 1) / GR_BM;
    if (blocks > 0x7fffffffL) return SLK_ERR_UNSUPPORTED;
    dim3 grid((unsigned)blocks), block(512);
    if (K <= 64) hipLaunchKernelGGL((gemm_rows_kernel<32, STATS>), grid, block, 0, s, x, ldx, W, bias, y, ldy, M, K, N, stats);
    else if (K <= 96) hipLaunchKernelGGL((gemm_rows_kernel<48, STATS>), grid, block, 0, s, x, ldx, W, bias, y, ldy, M, K, N, stats);
    else if (K <= 128) hipLaunchKernelGGL((gemm_rows_kernel<64, STATS>), grid, block, 0, s, x, ldx, W, bias, y, ldy, M, K, N, stats);
    else return SLK_ERR_UNSUPPORTED;
    return slk_launch_status();
}

// logits = x.W^T + b with optional softmax row statistics.  Returns SLK_ERR_UNSUPPORTED for K > 128 (callers fall
// back to slk_gemm_bias_act_f32 + slk_softmax_rowstats_f32).
extern "C" int slk_linear_rowstats_f32(const float *x, long ldx, const float *W, const float *bias, float *y, long ldy,
                                       long M, int K, int N, float *stats, slk_stream_t stream)
{
    if (!x || !W || !y || M < 0 || K < 1 || N < 1 || ldx < K || ldy < N) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    if (stats) return launch_rows<true>(x, ldx, W, bias, y, ldy, M, K, N, reinterpret_cast<float2 *>(stats), slk_stream(stream));
    return launch_rows<false>(x, ldx, W, bias, y, ldy, M, K, N, nullptr, slk_stream(stream));
}

// posterior from logits + statistics, in place allowed: p = exp(l - max) * inv_sum  (layers.py:311-314)
__global__ void normalise_rows_kernel(const float *__restrict__ logits, long ld_in, const float2 *__restrict__ stats,
                                      float *__restrict__ post, long ld_out, long M, int N)
{
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const long row = (long)(i / N);
        const int c = (int)(i - (size_t)row * N);
        const float2 st = stats[row];
        post[row * ld_out + c] = __expf(logits[row * ld_in + c] - st.x) * st.y;
    }
}

extern "C" int slk_softmax_from_stats_f32(const float *logits, long ld_in, const float *stats, float *post, long ld_out,
                                          long M, int N, slk_stream_t stream)
{
    if (!logits || !stats || !post || M < 0 || N < 1 || ld_in < N || ld_out < N) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    size_t total = (size_t)M * N, blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(normalise_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), logits, ld_in,
                       reinterpret_cast<const float2 *>(stats), post, ld_out, M, N);
    return slk_launch_status();
}
