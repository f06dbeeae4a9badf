// gemm_rows_f16x3.hip -- the wide Softmax projection (sloika/layers.py:310-313) on the FP16 matrix pipe with
// float32-grade accuracy: every float32 operand is split into two halves  v = hi + lo  (hi = fp16(v),
// lo = fp16(v - hi), 22 significand bits together) and the product is evaluated as
//        x.w  ~=  x_hi.w_hi + x_hi.w_lo + x_lo.w_hi          (the lo.lo term is < 2^-22 relative and dropped)
// with float32 accumulation inside v_mfma_f32_32x32x16_f16.  Three fp16 MFMAs do the work of sixteen fp32 ones
// (the fp32-input MFMA runs at 1/16 of the fp16 rate on gfx950 and there is no TF32 path), i.e. ~5x the fp32 MFMA
// throughput at an error of a few float32 ulps -- far inside the 1e-4 layer tolerance (tests/test_gpu_gemm.py).
//
// Structure = gemm_rows.hip (x-stationary, 128 rows x all N columns per 512-thread workgroup, weight tiles double
// buffered in LDS, row statistics accumulated online); the weights arrive pre-split (slk_split_f16x2_f32).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define GH_BM 128
#define GH_BN 64

__global__ void split_f16x2_kernel(const float *__restrict__ w, int rows, int K, int KP, _Float16 *__restrict__ hi,
                                   _Float16 *__restrict__ lo)
{
    const size_t total = (size_t)rows * KP;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % KP);
        const size_t r = i / KP;
        const float v = k < K ? w[r * K + k] : 0.0f;
        const _Float16 h = (_Float16)v;
        hi[i] = h;
        lo[i] = (_Float16)(v - (float)h);
    }
}

// Split a float32 matrix [rows][K] into fp16 hi/lo parts [rows][KP], KP = K rounded up to a multiple of 16 (zero padded).
extern "C" int slk_split_f16x2_f32(const float *w, int rows, int K, void *hi, void *lo, slk_stream_t stream)
{
    if (!w || !hi || !lo || rows < 1 || K < 1) return SLK_ERR_INVALID_ARG;
    const int KP = (K + 15) / 16 * 16;
    size_t total = (size_t)rows * KP, blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), w, rows, K, KP,
                       static_cast<_Float16 *>(hi), static_cast<_Float16 *>(lo));
    return slk_launch_status();
}

template <int KS, bool STATS>
__global__ void __launch_bounds__(512, 2) gemm_rows_f16x3_kernel(const float *__restrict__ x, long ldx,
                                                                 const _Float16 *__restrict__ Whi,
                                                                 const _Float16 *__restrict__ Wlo,
                                                                 const float *__restrict__ bias, float *__restrict__ y,
                                                                 long ldy, long M, int K, int N,
                                                                 float2 *__restrict__ stats)
{
    constexpr int KP = 16 * KS;                    // padded K (halves per weight row in Whi/Wlo)
    constexpr int LD = KP + 8;                     // LDS row stride in halves: (KP+8)*2 B = odd multiple of 16 B
    __shared__ __attribute__((aligned(16))) _Float16 wsh[2][GH_BN * LD];
    __shared__ __attribute__((aligned(16))) _Float16 wsl[2][GH_BN * LD];
    __shared__ float2 red[2][GH_BM];
    // per-wave 32x32 staging tile for the epilogue: the accumulator (lane = column, register = row) is turned into
    // 16-byte row segments so that the logits leave as dwordx4 stores (4 per wave and tile instead of 16 dword
    // stores -- the dword form is store-issue bound at ~2 TB/s)
    constexpr int EPLD = 36;
    __shared__ __attribute__((aligned(16))) float eps[8][32 * EPLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    const long m0 = (long)blockIdx.x * GH_BM;
    const int ntiles = (N + GH_BN - 1) / GH_BN;

    // ---- A fragments: x[row][16s + 8h + j], split into hi/lo halves ----
    half8 ahi[KS], alo[KS];
    {
        const long row = m0 + 32 * wm + r;
        const float *xr = x + (row < M ? row : M - 1) * ldx;
#pragma unroll
        for (int s = 0; s < KS; s++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 16 * s + 8 * h + j;
                float v = xr[k < K ? k : 0];
                v = (row < M && k < K) ? v : 0.0f;
                const _Float16 hv = (_Float16)v;
                ahi[s][j] = hv;
                alo[s][j] = (_Float16)(v - (float)hv);
            }
        }
    }
    // ---- weight tile staging: 64 rows x KP halves for hi and lo, as 16-byte pieces ----
    constexpr int PPR = KP / 8;                                // pieces per row
    constexpr int PIECES = GH_BN * PPR;                        // per array
    constexpr int NLD = (2 * PIECES + 511) / 512;
    uint4 wreg[NLD];
    int st_off[NLD], st_src[NLD];
    bool st_lo[NLD], st_ok[NLD];
#pragma unroll
    for (int j = 0; j < NLD; j++) {
        const int idx = tid + 512 * j;
        st_ok[j] = idx < 2 * PIECES;
        st_lo[j] = idx >= PIECES;
        const int p = st_lo[j] ? idx - PIECES : idx;
        const int c = p / PPR, k8 = (p % PPR) * 8;
        st_off[j] = c * LD + k8;
        st_src[j] = c * KP + k8;
    }
    auto load_tile = [&](int nt) {
#pragma unroll
        for (int j = 0; j < NLD; j++) {
            const int c = st_off[j] / LD;
            const int gn = nt * GH_BN + c;
            const _Float16 *base = st_lo[j] ? Wlo : Whi;
            const size_t off = (size_t)(gn < N ? nt * GH_BN : 0) * KP + (gn < N ? st_src[j] : 0);
            uint4 v = *reinterpret_cast<const uint4 *>(base + (st_ok[j] ? off : 0));
            if (!(st_ok[j] && gn < N)) v = make_uint4(0u, 0u, 0u, 0u);
            wreg[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; j++)
            if (st_ok[j]) *reinterpret_cast<uint4 *>(&(st_lo[j] ? wsl : wsh)[buf][st_off[j]]) = wreg[j];
    };

    float *const ywave = y + (m0 + 32 * wm) * ldy + 32 * wn;
    const long rows_left = M - (m0 + 32 * wm);
    float *const ep = eps[wave];
    // read-back role of this lane: 4 passes, pass q covers rows 8q..8q+7; lane -> (row 8q + lane/8, columns 4*(lane%8)..+3)
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    float rm[16], rs[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { rm[i] = -INFINITY; rs[i] = 0.0f; }

    load_tile(0);
    store_tile(0);
    if (ntiles > 1) load_tile(1);
    __syncthreads();

    for (int nt = 0; nt < ntiles; nt++) {
        const _Float16 *th = &wsh[nt & 1][(32 * wn + r) * LD + 8 * h];
        const _Float16 *tl = &wsl[nt & 1][(32 * wn + r) * LD + 8 * h];
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const half8 bh = *reinterpret_cast<const half8 *>(th + 16 * s);
            const half8 bl = *reinterpret_cast<const half8 *>(tl + 16 * s);
            // small terms first so that they are not absorbed by the large one
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[s], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[s], bh, acc, 0, 0, 0);
        }
        if (nt + 1 < ntiles) store_tile((nt + 1) & 1);
        if (nt + 2 < ntiles) load_tile(nt + 2);
        // ---- epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*h][col = r] ----
        const int col = nt * GH_BN + 32 * wn + r;
        const bool colok = col < N;
        const float bv = (bias && colok) ? bias[col] : 0.0f;
        float *const ytile = ywave + nt * GH_BN;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int rloc = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const float v = acc[reg] + bv;
            ep[rloc * EPLD + r] = v;
            if (STATS && colok) {
                const float e = __expf(-fabsf(v - rm[reg]));
                rs[reg] = (v <= rm[reg]) ? rs[reg] + e : rs[reg] * e + 1.0f;
                rm[reg] = fmaxf(rm[reg], v);
            }
        }
        // same wave wrote and reads: only the LDS counter has to drain (no barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int cbase = nt * GH_BN + 32 * wn + ec;               // first of this lane's 4 columns
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int rloc = 8 * q + er;
            const float4 v4 = *reinterpret_cast<const float4 *>(&ep[rloc * EPLD + ec]);
            if (rloc < rows_left) {
                float *dst = ytile + (size_t)rloc * ldy + ec;
                if (vec_ok && cbase + 3 < N) *reinterpret_cast<float4 *>(dst) = v4;
                else {
                    if (cbase + 0 < N) dst[0] = v4.x;
                    if (cbase + 1 < N) dst[1] = v4.y;
                    if (cbase + 2 < N) dst[2] = v4.z;
                    if (cbase + 3 < N) dst[3] = v4.w;
                }
            }
        }
        __syncthreads();
    }
    if (STATS) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            float m = rm[reg], s = rs[reg];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const float om = __shfl_xor(m, o), os = __shfl_xor(s, o);
                const float mn = fmaxf(m, om);
                const float sa = (m == -INFINITY) ? 0.0f : s * __expf(m - mn);
                const float sb = (om == -INFINITY) ? 0.0f : os * __expf(om - mn);
                s = sa + sb;
                m = mn;
            }
            if (r == 0) red[wn][32 * wm + (reg & 3) + 8 * (reg >> 2) + 4 * h] = make_float2(m, s);
        }
        __syncthreads();
        if (tid < GH_BM && m0 + tid < M) {
            const float2 p0 = red[0][tid], p1 = red[1][tid];
            const float mn = fmaxf(p0.x, p1.x);
            const float s = ((p0.x == -INFINITY) ? 0.0f : p0.y * __expf(p0.x - mn)) +
                            ((p1.x == -INFINITY) ? 0.0f : p1.y * __expf(p1.x - mn));
            stats[m0 + tid] = make_float2(mn, 1.0f / s);
        }
    }
}

template <int KS>
static int launch_f16x3(const float *x, long ldx, const _Float16 *hi, const _Float16 *lo, const float *bias, float *y,
                        long ldy, long M, int K, int N, float2 *stats, hipStream_t s)
{
    dim3 grid((unsigned)((M + GH_BM - 1) / GH_BM)), block(512);
    if (stats) hipLaunchKernelGGL((gemm_rows_f16x3_kernel<KS, true>), grid, block, 0, s, x, ldx, hi, lo, bias, y, ldy, M, K, N, stats);
    else hipLaunchKernelGGL((gemm_rows_f16x3_kernel<KS, false>), grid, block, 0, s, x, ldx, hi, lo, bias, y, ldy, M, K, N, stats);
    return slk_launch_status();
}

// logits = x.W^T + b from pre-split weights (slk_split_f16x2_f32), optional softmax row statistics.  K <= 128.
extern "C" int slk_linear_rowstats_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *bias,
                                         float *y, long ldy, long M, int K, int N, float *stats, slk_stream_t stream)
{
    if (!x || !W_hi || !W_lo || !y || M < 0 || K < 1 || N < 1 || ldx < K || ldy < N) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    if ((M + GH_BM - 1) / GH_BM > 0x7fffffffL) return SLK_ERR_UNSUPPORTED;
    const _Float16 *hi = static_cast<const _Float16 *>(W_hi), *lo = static_cast<const _Float16 *>(W_lo);
    float2 *st = reinterpret_cast<float2 *>(stats);
    hipStream_t s = slk_stream(stream);
    switch ((K + 15) / 16) {
    case 1: return launch_f16x3<1>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 2: return launch_f16x3<2>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 3: return launch_f16x3<3>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 4: return launch_f16x3<4>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 5: return launch_f16x3<5>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 6: return launch_f16x3<6>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 7: return launch_f16x3<7>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    case 8: return launch_f16x3<8>(x, ldx, hi, lo, bias, y, ldy, M, K, N, st, s);
    default: return SLK_ERR_UNSUPPORTED;
    }
}
