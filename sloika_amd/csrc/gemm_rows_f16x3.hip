// gemm_rows_f16x3.hip -- the wide Softmax projection (sloika/layers.py:310-313) on the FP16 matrix pipe with
// float32-grade accuracy: every float32 operand is split into two halves  v = hi + lo  (hi = fp16(v),
// lo = fp16(v - hi), 22 significand bits together) and the product is evaluated as
//        x.w  ~=  x_hi.w_hi + x_hi.w_lo + x_lo.w_hi          (the lo.lo term is < 2^-22 relative and dropped)
// with float32 accumulation inside v_mfma_f32_32x32x16_f16.  Three fp16 MFMAs do the work of sixteen fp32 ones
// (the fp32-input MFMA runs at 1/16 of the fp16 rate on gfx950 and there is no TF32 path), i.e. ~5x the fp32 MFMA
// throughput at an error of a few float32 ulps -- far inside the 1e-4 layer tolerance (tests/test_gpu_gemm.py).
//
// x-stationary like gemm_rows.hip (128 rows (GH_NWM = 4 groups of 32) x all N columns per workgroup, row statistics accumulated online); the weights
// arrive pre-split (slk_split_f16x2_f32).
//
// Range: fp16 overflows at 65504 and loses its lo half below 6e-5, so every row of x (in the kernel) and every row of W (in
// slk_split_f16x2_f32) is first scaled by a power of two that brings its largest magnitude into [1, 2); the scaling is
// exact and is undone on the float32 accumulators (one multiply by the product row's inverse scale, one fused
// multiply-add with the weight row's inverse scale and the bias).  Any finite float32 operand is therefore handled with
// 22 significand bits relative to its row maximum -- float32-grade.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Row groups of 32 per workgroup.  Every workgroup streams the whole of W through LDS once, so W traffic per row falls with
// the workgroup height.  Back-to-back launches in one process (tools/build_gemm_variants.sh): 2 groups 1.35 ms, 4 groups 1.03,
// 5 groups 0.97, 6 groups 0.99 (13 waves need <= 128 registers: spills) -- but inside the pipeline, between the last GRU and
// the decoder, 4 and 5 groups both take 0.91-0.92 ms (alternating runs on one device), so the default stays at 4, which keeps
// the patch stores for K up to 128 and fits 168 registers without scratch.
#ifndef GH_NWM
#define GH_NWM 4
#endif
#define GH_BM (32 * GH_NWM)
#define GH_THREADS (64 * (2 * GH_NWM + 1))
#ifndef GH_RINGMAX
#define GH_RINGMAX 3
#endif
#ifndef GH_BIASMAX
#define GH_BIASMAX 2048
#endif
#define GH_BN 64

// power-of-two scale that brings a row whose largest magnitude is amax into [1, 2) (exponent kept inside [27, 227])
__device__ __forceinline__ float gh_pow2_scale(float amax, float &inv)
{
    const int e = min(max((int)((__float_as_uint(amax) >> 23) & 0xff), 27), 227);
    inv = __uint_as_float((unsigned)e << 23);
    return __uint_as_float((unsigned)(254 - e) << 23);
}

// one wave per weight row: row maximum, power-of-two scale, hi/lo halves of the scaled row
__global__ void split_f16x2_kernel(const float *__restrict__ w, int rows, int K, int KP, _Float16 *__restrict__ hi,
                                   _Float16 *__restrict__ lo, float *__restrict__ inv_scale)
{
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < rows; r += gridDim.x * (blockDim.x >> 6)) {
        float m = 0.0f;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(w[(size_t)r * K + k]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float inv;
        const float sc = gh_pow2_scale(m, inv);
        if (lane == 0) inv_scale[r] = inv;
        for (int k = lane; k < KP; k += 64) {
            const float v = k < K ? w[(size_t)r * K + k] * sc : 0.0f;
            const _Float16 h = (_Float16)v;
            hi[(size_t)r * KP + k] = h;
            lo[(size_t)r * KP + k] = (_Float16)(v - (float)h);
        }
    }
}

// Split a float32 matrix [rows][K] into fp16 hi/lo parts [rows][KP], KP = K rounded up to a multiple of 16 (zero padded),
// each row scaled by a power of two to a maximum in [1, 2); inv_scale[rows] receives the inverse scales.
extern "C" int slk_split_f16x2_f32(const float *w, int rows, int K, void *hi, void *lo, float *inv_scale, slk_stream_t stream)
{
    if (!w || !hi || !lo || !inv_scale || rows < 1 || K < 1) return SLK_ERR_INVALID_ARG;
    const int KP = (K + 15) / 16 * 16;
    int blocks = (rows + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), w, rows, K, KP,
                       static_cast<_Float16 *>(hi), static_cast<_Float16 *>(lo), inv_scale);
    return slk_launch_status();
}

// Kernel structure
//   * GH_BM = 128 rows x all N columns per workgroup; the MFMA computes the TRANSPOSED 64-column tile D[W column][x row]
//     (A = weights from LDS, B = x held in registers for the whole row block): a lane then owns 16 columns of ONE output
//     row, four of them consecutive per accumulator quad, so the logits leave as 16-byte stores straight from the
//     accumulators and the online softmax statistics are two scalars per lane.
//   * waves 0-7 compute; wave 8 does nothing but stream the weight tiles into a 3-slot LDS ring with LDS-DMA
//     (global_load_lds_dwordx4), two tiles ahead.  The split exists because of vmcnt: it counts loads AND stores and
//     they complete out of order with respect to each other, so a wave that has logit stores in flight can only wait
//     for a weight load with vmcnt(0) -- i.e. for a full store round trip (microseconds) on every tile.  The compute
//     waves issue no loads in the loop, hence never wait on memory; the loader wave has no stores.
//   * one s_barrier per tile (LDS counter only: __syncthreads() would drain vmcnt as well).
// ACT: activation applied to the stored values (SLK_ACT_LINEAR for logits; FeedForward layers use tanh etc.); the row
// statistics (STATS) are those of the pre-activation values and only make sense with SLK_ACT_LINEAR.
// TRSTORE: the full, aligned tiles leave through a per-wave LDS patch so that one store instruction writes 8 rows x 128
// contiguous bytes (whole cache lines) instead of 32 rows x 32 bytes: straight from the accumulators a lane owns 16 bytes of
// ONE row, and the store path works through 32 different lines per instruction (measured: 0.92 ms, 0.60 without stores).
#ifndef GH_TRSTORE
#define GH_TRSTORE true
#endif
// XM (the softmax layer of the training step, train_network.py:128-136, as two passes over the SAME products instead of a logits
// tensor that is written, read, overwritten with its gradient and read again):
//   1  statistics pass: no logits are stored.  Per row: maximum, 1 / sum of exp(l - maximum), the first column that attains the
//      maximum (T.argmax's rule) and the logit of the row's label, and from them the row's loss term, its accuracy term and
//      the coefficient of its gradient -- the arithmetic of softmax_xent_grad_kernel (train.hip), which this pair replaces --
//      leaving xrow[row] = {maximum, 1 / sum, coefficient, label}
//   2  gradient pass: the element stored is coefficient * (p - [column == label]), p = exp(l - maximum) / sum, with l the
//      same product bit for bit; columns N .. ldy - 1 are written as zeros (the gradient is contracted with a padded row length)
struct XentArgs {
    float4 *xrow;                  // [M] pass 1 writes, pass 2 reads
    const int32_t *labels;         // [M] = [T][B]
    const float *weights;          // [M]
    float *loss_rows, *correct_rows;
    int T, B, drop;
    float min_prob;
};
template <int KS, bool STATS, int ACT, bool TRSTORE = GH_TRSTORE, int XM = 0>
__global__ void __launch_bounds__(GH_THREADS) gemm_rows_f16x3_kernel(const float *__restrict__ x, long ldx,
                                                              const _Float16 *__restrict__ Whi,
                                                              const _Float16 *__restrict__ Wlo,
                                                              const float *__restrict__ winv,
                                                              const float *__restrict__ bias, float *__restrict__ y,
                                                              long ldy, long M, int K, int N,
                                                              float2 *__restrict__ stats, XentArgs xa)
{
    static_assert(XM == 0 || (ACT == SLK_ACT_LINEAR && (XM == 1) == STATS), "cross-entropy passes: linear logits, statistics in pass 1");
    constexpr int KP = 16 * KS;                    // padded K (halves per weight row in Whi/Wlo)
    constexpr int LD = KP + 8;                     // LDS row stride in halves: (KP+8)*2 B = odd multiple of 16 B
    constexpr int RING = (KS <= 9 && GH_RINGMAX >= 3) ? 3 : 2;          // weight-tile slots (a tile is 64 x (KP+8) halves, twice): LDS budget
    constexpr int PPR = KP / 8 + 1;                // 16-byte pieces per LDS row (the last one is padding)
    constexpr int BIAS_MAX = GH_BIASMAX + GH_BN;
    __shared__ __attribute__((aligned(16))) _Float16 wsh[RING][GH_BN * LD];
    __shared__ __attribute__((aligned(16))) _Float16 wsl[RING][GH_BN * LD];
    __shared__ float2 red[2][GH_BM];
    __shared__ int redarg[XM == 1 ? 2 : 1][XM == 1 ? GH_BM : 1];
    __shared__ float redlab[XM == 1 ? GH_BM : 1];
    __shared__ __attribute__((aligned(16))) float bias_lds[BIAS_MAX];   // zero padded to whole tiles
    __shared__ __attribute__((aligned(16))) float winv_lds[BIAS_MAX];   // inverse scales of the weight rows (= columns here)
    constexpr int TP = 36;                          // floats per row of a wave's 32 x 32 store patch (144 B: no bank clash)
    constexpr bool TR = TRSTORE && XM != 1 && KS <= (GH_NWM > 4 ? 6 : 8);         // (K > 128: the weight ring leaves no room for the patches)
    __shared__ __attribute__((aligned(16))) float patch[TR ? 2 * GH_NWM * 32 * TP : 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long m0 = (long)blockIdx.x * GH_BM;
    const int ntiles = (N + GH_BN - 1) / GH_BN;
    auto tile_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    for (int i = tid; i < ntiles * GH_BN; i += GH_THREADS) {
        bias_lds[i] = (bias && i < N) ? bias[i] : 0.0f;
        winv_lds[i] = i < N ? winv[i] : 0.0f;
    }

    if (wave == 2 * GH_NWM) {
        // =============================== loader wave ===============================
        // one tile = 64 rows x PPR pieces per array = PPR chunks of 64 pieces (1 KiB) each, contiguous in LDS;
        // columns beyond N re-read row N-1 (their products are never stored nor counted)
        auto dma_tile = [&](int nt) {
            const int slot = nt % RING;
#pragma unroll
            for (int arr = 0; arr < 2; arr++) {
                const _Float16 *W = arr ? Wlo : Whi;
                _Float16 *dst = arr ? wsl[slot] : wsh[slot];
#pragma unroll
                for (int ch = 0; ch < PPR; ch++) {
                    const int P = 64 * ch + lane, c = P / PPR, pc = P % PPR;
                    const int gn = min(nt * GH_BN + c, N - 1);
                    const _Float16 *src = W + (size_t)gn * KP + 8 * (pc < PPR - 1 ? pc : 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)(dst + 512 * ch), 16, 0, 0);
                }
            }
        };
        dma_tile(0);
        if (RING > 2 && ntiles > 1) dma_tile(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tile_barrier();                                            // (P) the first RING-1 tiles are in LDS, bias staged
        for (int nt = 0; nt < ntiles; nt++) {
            // slot (nt+RING-1) % RING held tile nt-1, which every compute wave finished before the previous barrier
            if (nt + RING - 1 < ntiles) dma_tile(nt + RING - 1);
            // tile nt+1 must be complete before the compute waves pass this barrier; with three slots the requests
            // just issued (the PPR*2 youngest; loads complete in order) may stay in flight
            if (RING > 2 && nt + 2 < ntiles && 2 * PPR <= 63) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPR <= 63 ? 2 * PPR : 0) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tile_barrier();
        }
        if (STATS) tile_barrier();                                 // the statistics reduction's barrier
        return;
    }

    // =============================== compute waves ===============================
    const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    // ---- x fragments: x[row][16s + 8h + j], split into hi/lo halves ----
    half8 ahi[KS], alo[KS];
    float xinv;                                    // inverse of this lane's row scale
    const long row = m0 + 32 * wm + r;
    const bool rowok = row < M;
    {
        const float *xr = x + (rowok ? row : M - 1) * ldx;
        // 16-byte loads, all in flight together (element-wise conditional loads serialise: one memory round trip each)
        const bool fast = (K % 8 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        float xv[KS][8];
        if (fast) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int k0 = 16 * s + 8 * h;
                const float *src = xr + (k0 < K ? k0 : 0);
                const float4 u0 = *reinterpret_cast<const float4 *>(src), u1 = *reinterpret_cast<const float4 *>(src + 4);
                const bool ok = rowok && (k0 < K);
                xv[s][0] = ok ? u0.x : 0.0f; xv[s][1] = ok ? u0.y : 0.0f; xv[s][2] = ok ? u0.z : 0.0f; xv[s][3] = ok ? u0.w : 0.0f;
                xv[s][4] = ok ? u1.x : 0.0f; xv[s][5] = ok ? u1.y : 0.0f; xv[s][6] = ok ? u1.z : 0.0f; xv[s][7] = ok ? u1.w : 0.0f;
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; s++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 16 * s + 8 * h + j;
                    const float v = xr[k < K ? k : 0];
                    xv[s][j] = (rowok && k < K) ? v : 0.0f;
                }
            }
        }
        // row scale: the row's K is split between lanes l and l ^ 32
        float amax = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; s++)
#pragma unroll
            for (int j = 0; j < 8; j++) amax = fmaxf(amax, fabsf(xv[s][j]));
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        const float xs = gh_pow2_scale(amax, xinv);
#pragma unroll
        for (int s = 0; s < KS; s++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float v = xv[s][j] * xs;
                const _Float16 hv = (_Float16)v;
                ahi[s][j] = hv;
                alo[s][j] = (_Float16)(v - (float)hv);
            }
        }
    }
    // this lane's output row and the 16 columns it holds per 64-column tile: 32*wn + 8*q + 4*h + (0..3), q = 0..3
    float *const yrow = y + (rowok ? row : 0) * ldy + 32 * wn + 4 * h;
    const bool vec_ok = (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    float rmax = -INFINITY, rsum = 0.0f;           // online softmax statistics of this lane's share of the row
    int rarg = 0x7fffffff;                         // XM 1: first column of this lane's share that attains rmax
    float xmx = 0.0f, xinvs = 0.0f, xcoef = 0.0f;  // XM 2: the row's maximum, 1 / sum, coefficient
    int xlabel = -1;
    float llab = 0.0f;                             // XM 1: the label's logit, in the one lane of the row's four that meets it
    bool lfound = false;
    if constexpr (XM == 1) xlabel = xa.labels[rowok ? row : M - 1];
    if constexpr (XM == 2) {
        const float4 rc = xa.xrow[rowok ? row : M - 1];
        xmx = rc.x; xinvs = rc.y; xcoef = rc.z; xlabel = __float_as_int(rc.w);
    }
    auto note_label = [&](const f32x16 &acc, int cbase) {
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const bool hit = cbase + 8 * (reg >> 2) + (reg & 3) == xlabel;
            llab = hit ? acc[reg] : llab;
            lfound |= hit;
        }
    };
    // XM 2: logit -> gradient element
    auto xgrad = [&](float l, int col) { return xcoef * (__expf(l - xmx) * xinvs - (col == xlabel ? 1.0f : 0.0f)); };

    tile_barrier();                                                // (P)
    // Software pipeline: the epilogue of tile nt-1 (stores, statistics) is in the same straight-line block as the MFMA
    // chain of tile nt, so the scheduler can slot its stores and VALU work between the dependent MFMAs, and the logit
    // stores of a workgroup spread over the whole tile time instead of arriving as a burst after every barrier.
    auto mma = [&](int nt) {
        const _Float16 *th = &wsh[nt % RING][(32 * wn + r) * LD + 8 * h];
        const _Float16 *tl = &wsl[nt % RING][(32 * wn + r) * LD + 8 * h];
        const int cbase = nt * GH_BN + 32 * wn + 4 * h;            // column of acc[0]; acc[4q+i] is column cbase + 8q + i
        f32x16 acc;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) acc[reg] = 0.0f;
        (void)cbase;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const half8 bh = *reinterpret_cast<const half8 *>(th + 16 * s);
            const half8 bl = *reinterpret_cast<const half8 *>(tl + 16 * s);
            // D[m = W column][n = x row]; small terms first so that they are not absorbed by the large one
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, alo[s], acc, 0, 0, 0);
            // The first MFMA of a tile has C = 0 and a freshly defined destination: where one of its operands is not used again
            // (alo[0] in the last tile of a workgroup) hipcc (ROCm 7.2) lets the destination overlap that operand's registers,
            // which gfx950 does not execute safely for this 16-pass instruction (csrc/softmax_viterbi.hip has the story).  An
            // empty asm that takes the result and the operands keeps them apart.
            if (s == 0) asm volatile("" : "+v"(acc) : "v"(alo[0]), "v"(bh));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ahi[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ahi[s], acc, 0, 0, 0);
        }
        return acc;
    };
    // scaled accumulators -> logits: (acc * row inverse scale) * column inverse scale + bias
    auto finish = [&](int nt, const f32x16 &raw) {
        const int cbase = nt * GH_BN + 32 * wn + 4 * h;
        f32x16 v;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const float4 b4 = *reinterpret_cast<const float4 *>(&bias_lds[cbase + 8 * q]);
            const float4 w4 = *reinterpret_cast<const float4 *>(&winv_lds[cbase + 8 * q]);
            v[4 * q] = fmaf(raw[4 * q] * xinv, w4.x, b4.x);
            v[4 * q + 1] = fmaf(raw[4 * q + 1] * xinv, w4.y, b4.y);
            v[4 * q + 2] = fmaf(raw[4 * q + 2] * xinv, w4.z, b4.z);
            v[4 * q + 3] = fmaf(raw[4 * q + 3] * xinv, w4.w, b4.w);
        }
        return v;
    };
    // acc[reg] = logit(row, column cbase + 8*(reg>>2) + (reg&3))
    auto epilogue = [&](int nt, const f32x16 &raw) {
        const f32x16 acc = finish(nt, raw);
        const int cbase = nt * GH_BN + 32 * wn + 4 * h;
        const bool tile_full = (nt + 1) * GH_BN <= N;              // workgroup-uniform: every column of the tile exists
        const bool full = tile_full || cbase + 27 < N;             // all 16 columns of this lane exist
        f32x16 o = acc;
        if constexpr (ACT != SLK_ACT_LINEAR) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) o[reg] = slk_act_t<ACT>(acc[reg]);
        }
        if constexpr (XM == 2) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) o[reg] = xgrad(acc[reg], cbase + 8 * (reg >> 2) + (reg & 3));
        }
        if constexpr (XM == 1) {
            // nothing is stored
        } else if (tile_full && vec_ok) {
            if (rowok) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    *reinterpret_cast<float4 *>(yrow + nt * GH_BN + 8 * q) =
                        make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            }
        } else if (rowok && vec_ok && (N & 3) == 0 && XM != 2) {
            // the partial last tile of a row length that is a multiple of four (a Gru projection: 3 n columns): a lane's four
            // consecutive columns exist together or not at all -- one 16-byte store instead of four guarded dword stores
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (cbase + 8 * q < N)
                    *reinterpret_cast<float4 *>(yrow + nt * GH_BN + 8 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        } else if (rowok) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float *dst = yrow + nt * GH_BN + 8 * q;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int col = cbase + 8 * q + i;
                    if (col < N) dst[i] = o[4 * q + i];
                    else if (XM == 2 && col < ldy) dst[i] = 0.0f;
                }
            }
        }
        if (STATS) {
            // tile maximum first, then one exp per element against the new running maximum
            float tm = -INFINITY;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const bool ok = full || (cbase + 8 * (reg >> 2) + (reg & 3) < N);
                tm = fmaxf(tm, ok ? acc[reg] : -INFINITY);
            }
            const float mn = fmaxf(rmax, tm);
            if constexpr (XM == 1) {
                // first column of the tile that attains its maximum (columns rise with reg; scanned downwards, the smallest wins);
                // a tile only takes over from the earlier ones -- whose columns are all smaller -- when it is strictly larger
                int ti = 0x7fffffff;
#pragma unroll
                for (int reg = 15; reg >= 0; reg--) {
                    const int col = cbase + 8 * (reg >> 2) + (reg & 3);
                    const bool ok = full || col < N;
                    ti = (ok && acc[reg] == tm) ? col : ti;
                }
                rarg = tm > rmax ? ti : rarg;
                note_label(acc, cbase);
            }
            if (mn > -INFINITY) {
                float sum = (rmax == -INFINITY) ? 0.0f : rsum * __expf(rmax - mn);
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const bool ok = full || (cbase + 8 * (reg >> 2) + (reg & 3) < N);
                    sum += ok ? __expf(acc[reg] - mn) : 0.0f;
                }
                rsum = sum;
                rmax = mn;
            }
        }
    };
    // Straight-line epilogue for the common case (whole workgroup inside M, aligned rows, every column of the tile
    // exists): no branch, so that it shares a scheduling region with the next tile's MFMA chain.
    auto epilogue_fast = [&](int nt, const f32x16 &raw) {
        const f32x16 acc = finish(nt, raw);
        f32x16 o = acc;
        if constexpr (ACT != SLK_ACT_LINEAR) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) o[reg] = slk_act_t<ACT>(acc[reg]);
        }
        if constexpr (XM == 2) {
            const int cbase = nt * GH_BN + 32 * wn + 4 * h;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) o[reg] = xgrad(acc[reg], cbase + 8 * (reg >> 2) + (reg & 3));
        }
        if constexpr (XM == 1) {
            // nothing is stored
        } else if constexpr (TR) {
            // my 16 values -> patch[row r][column 8q + 4h + i]; then lane l takes 16 bytes of row 8i + (l >> 3), so that
            // eight lanes cover the 128 bytes of a row and one instruction writes eight whole lines
            float *pw = patch + wave * (32 * TP);
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4 *>(&pw[r * TP + 8 * q + 4 * h]) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
            float *const ybase = y + (m0 + 32 * wm) * ldy + nt * GH_BN + 32 * wn;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int rr = 8 * i + (lane >> 3), cc = 4 * (lane & 7);
                const float4 v = *reinterpret_cast<const float4 *>(&pw[rr * TP + cc]);
                *reinterpret_cast<float4 *>(ybase + (long)rr * ldy + cc) = v;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<float4 *>(yrow + nt * GH_BN + 8 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
        if (STATS) {
            float tm = acc[0];
#pragma unroll
            for (int reg = 1; reg < 16; reg++) tm = fmaxf(tm, acc[reg]);
            if constexpr (XM == 1) {
                const int cbase = nt * GH_BN + 32 * wn + 4 * h;
                int ti = 0x7fffffff;
#pragma unroll
                for (int reg = 15; reg >= 0; reg--) ti = acc[reg] == tm ? cbase + 8 * (reg >> 2) + (reg & 3) : ti;
                rarg = tm > rmax ? ti : rarg;
                note_label(acc, cbase);
            }
            const float mn = fmaxf(rmax, tm);
            const float ms = fmaxf(mn, -3.0e38f);                  // all -inf so far: exp(-inf - ms) = 0, no NaN
            float sum = rsum * __expf(fmaxf(rmax, -3.0e38f) - ms);  // rsum = 0 while rmax = -inf
#pragma unroll
            for (int reg = 0; reg < 16; reg++) sum += __expf(acc[reg] - ms);
            rsum = sum;
            rmax = mn;
        }
    };
    // ask the scheduler to slot the epilogue between the dependent MFMAs: per MFMA a few VALU ops and one transcendental
    auto interleave_hint = [] {
#pragma unroll
        for (int i = 0; i < 3 * KS; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, (STATS || XM == 2) ? 5 : 1, 0);   // VALU
            if (STATS || XM == 2) __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);    // one transcendental
        }
    };
    // two accumulator sets swap roles statically (loop unrolled by two): no register copies between tiles
    const bool wg_fast = vec_ok && (m0 + GH_BM <= M);
    const int nfull = N / GH_BN;                                   // tiles whose 64 columns all exist
    f32x16 accA = mma(0), accB = accA;
    tile_barrier();
    int nt = 1;
    if (wg_fast) {
        // epilogues of full tiles 0 .. nfast-1 next to the MFMAs of tiles 1 .. nfast
        const int nfast = nfull < ntiles - 1 ? nfull : ntiles - 1;
#ifndef GH_PHASE
#define GH_PHASE 1               /* 0: every wave in the same order (round 1); measured 1.030 -> 1.005 ms in one process */
#endif
        // Waves w and w+4 share a SIMD and the tile barrier keeps them in step: with the same instruction order both want the
        // matrix pipe at the same time and the vector unit at the same time.  So the upper four run a tile period the other way
        // round -- the epilogue of the previous tile FIRST, then this tile's MFMAs -- and each half's MFMAs run under the other
        // half's epilogue.
        if (GH_PHASE == 0 || wave < GH_NWM) {
            for (; nt + 1 <= nfast; nt += 2) {
                accB = mma(nt);
                if (GH_PHASE == 2) __builtin_amdgcn_sched_barrier(0);
                epilogue_fast(nt - 1, accA);
                if (GH_PHASE != 2) interleave_hint();
                tile_barrier();
                accA = mma(nt + 1);
                if (GH_PHASE == 2) __builtin_amdgcn_sched_barrier(0);
                epilogue_fast(nt, accB);
                if (GH_PHASE != 2) interleave_hint();
                tile_barrier();
            }
        } else {
            for (; nt + 1 <= nfast; nt += 2) {
                epilogue_fast(nt - 1, accA);
                __builtin_amdgcn_sched_barrier(0);
                accB = mma(nt);
                tile_barrier();
                epilogue_fast(nt, accB);
                __builtin_amdgcn_sched_barrier(0);
                accA = mma(nt + 1);
                tile_barrier();
            }
        }
    }
    // everything else (ragged workgroups, the partial last tile, an odd tile left over): the guarded epilogue
    for (; nt < ntiles; nt += 2) {
        accB = mma(nt);
        epilogue(nt - 1, accA);
        tile_barrier();
        if (nt + 1 < ntiles) {
            accA = mma(nt + 1);
            epilogue(nt, accB);
            tile_barrier();
        } else {
            accA = accB;                                           // the last tile's results, for the epilogue below
        }
    }
    epilogue(ntiles - 1, accA);
    if (STATS) {
        // combine the two column halves (lanes l and l^32 hold the same row), then the two column waves
        {
            const float om = __shfl_xor(rmax, 32), os = __shfl_xor(rsum, 32);
            const int oa = __shfl_xor(rarg, 32);
            const float mn = fmaxf(rmax, om);
            const float sa = (rmax == -INFINITY) ? 0.0f : rsum * __expf(rmax - mn);
            const float sb = (om == -INFINITY) ? 0.0f : os * __expf(om - mn);
            rarg = om > rmax ? oa : (om == rmax ? min(rarg, oa) : rarg);
            rsum = sa + sb;
            rmax = mn;
        }
        if (h == 0) {
            red[wn][32 * wm + r] = make_float2(rmax, rsum);
            if constexpr (XM == 1) redarg[wn][32 * wm + r] = rarg;
        }
        if constexpr (XM == 1) {
            if (lfound) redlab[32 * wm + r] = llab;
        }
        tile_barrier();
        if (tid < GH_BM && m0 + tid < M) {
            const float2 p0 = red[0][tid], p1 = red[1][tid];
            const float mn = fmaxf(p0.x, p1.x);
            const float s = ((p0.x == -INFINITY) ? 0.0f : p0.y * __expf(p0.x - mn)) +
                            ((p1.x == -INFINITY) ? 0.0f : p1.y * __expf(p1.x - mn));
            if constexpr (XM == 1) {
                const int a0 = redarg[0][tid], a1 = redarg[1][tid];
                const int arg = p1.x > p0.x ? a1 : (p1.x == p0.x ? min(a0, a1) : a0);
                // the row's terms of train_network.py:128-136 (as softmax_xent_grad_kernel computes them)
                const long m = m0 + tid;
                const int t = (int)(m / xa.B), label = xa.labels[m];
                const bool counted = t >= xa.drop && t < xa.T - xa.drop;
                const float count = (float)(xa.T - 2 * xa.drop) * (float)xa.B, inv = 1.0f / s;
                const float p_lab = __expf(redlab[tid] - mn) * inv;
                const float post_lab = xa.min_prob + (1.0f - xa.min_prob) * p_lab;
                const float w = counted ? xa.weights[m] / count : 0.0f;
                const float coef = w * (1.0f - xa.min_prob) * p_lab / post_lab;
                xa.loss_rows[m] = counted ? -w * logf(post_lab) : 0.0f;
                xa.correct_rows[m] = (counted && arg == label) ? 1.0f / count : 0.0f;
                xa.xrow[m] = make_float4(mn, inv, coef, __int_as_float(label));
            } else {
                stats[m0 + tid] = make_float2(mn, 1.0f / s);
            }
        }
    }
}

template <int KS>
static int launch_f16x3(const float *x, long ldx, const _Float16 *hi, const _Float16 *lo, const float *winv, const float *bias, float *y,
                        long ldy, long M, int K, int N, float2 *stats, int act, hipStream_t s)
{
    dim3 grid((unsigned)((M + GH_BM - 1) / GH_BM)), block(GH_THREADS);
#define F16X3_LAUNCH(ST, A) \
    hipLaunchKernelGGL((gemm_rows_f16x3_kernel<KS, ST, A>), grid, block, 0, s, x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, stats, XentArgs{})
    if (stats) F16X3_LAUNCH(true, SLK_ACT_LINEAR);
    else if (act == SLK_ACT_LINEAR) F16X3_LAUNCH(false, SLK_ACT_LINEAR);
    else if (act == SLK_ACT_TANH) F16X3_LAUNCH(false, SLK_ACT_TANH);
    else if (act == SLK_ACT_SIGMOID) F16X3_LAUNCH(false, SLK_ACT_SIGMOID);
    else if (act == SLK_ACT_RELU) F16X3_LAUNCH(false, SLK_ACT_RELU);
    else if (act == SLK_ACT_ELU) F16X3_LAUNCH(false, SLK_ACT_ELU);
    else return SLK_ERR_UNSUPPORTED;
#undef F16X3_LAUNCH
    return slk_launch_status();
}

static int dispatch_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *winv, const float *bias, float *y, long ldy,
                          long M, int K, int N, float *stats, int act, slk_stream_t stream)
{
    if (!x || !W_hi || !W_lo || !winv || !y || M < 0 || K < 1 || N < 1 || ldx < K || ldy < N || !slk_act_valid(act))
        return SLK_ERR_INVALID_ARG;
    if (stats && act != SLK_ACT_LINEAR) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    if ((M + GH_BM - 1) / GH_BM > 0x7fffffffL || N > GH_BIASMAX) return SLK_ERR_UNSUPPORTED;   // bias vector is staged in LDS
    const _Float16 *hi = static_cast<const _Float16 *>(W_hi), *lo = static_cast<const _Float16 *>(W_lo);
    float2 *st = reinterpret_cast<float2 *>(stats);
    hipStream_t s = slk_stream(stream);
    switch ((K + 15) / 16) {
    case 1: return launch_f16x3<1>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 2: return launch_f16x3<2>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 3: return launch_f16x3<3>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 4: return launch_f16x3<4>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 5: return launch_f16x3<5>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 6: return launch_f16x3<6>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 7: return launch_f16x3<7>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 8: return launch_f16x3<8>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 9: return launch_f16x3<9>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 10: return launch_f16x3<10>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 11: return launch_f16x3<11>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    case 12: return launch_f16x3<12>(x, ldx, hi, lo, winv, bias, y, ldy, M, K, N, st, act, s);
    default: return SLK_ERR_UNSUPPORTED;
    }
}

// logits = x.W^T + b from pre-split weights (slk_split_f16x2_f32: halves + inverse row scales), optional softmax row
// statistics.  K <= 192, N <= 2048.
extern "C" int slk_linear_rowstats_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *W_inv_scale,
                                         const float *bias, float *y, long ldy, long M, int K, int N, float *stats,
                                         slk_stream_t stream)
{
    return dispatch_f16x3(x, ldx, W_hi, W_lo, W_inv_scale, bias, y, ldy, M, K, N, stats, SLK_ACT_LINEAR, stream);
}

// y = act(x.W^T + b) from pre-split weights: FeedForward.run (sloika/layers.py:157-158) on the fp16 pipe.
// act: linear, tanh, sigmoid, relu or elu (others: SLK_ERR_UNSUPPORTED -> use slk_gemm_bias_act_f32).
extern "C" int slk_gemm_bias_act_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *W_inv_scale,
                                       const float *bias, float *y, long ldy, long M, int K, int N, int act, slk_stream_t stream)
{
    return dispatch_f16x3(x, ldx, W_hi, W_lo, W_inv_scale, bias, y, ldy, M, K, N, nullptr, act, stream);
}

// The softmax layer of a training step without a logits tensor (train_network.py:128-136): pass 1 computes the products and keeps
// only the rows' statistics, loss and accuracy terms; pass 2 computes them again and stores d loss / d logits [M][ld] (columns
// N .. ld - 1 zero).  Against logits -> statistics -> gradient in place (slk_linear_rowstats_f16x3 + slk_softmax_xent_grad_f32) the
// logits are neither written nor read back: 4 M ld bytes of traffic instead of 12 M ld, for 6 M N K more fp16 products.
// xrow: scratch of 4 M floats (16-byte aligned).  The usual layer widths only (K in 49..128): SLK_ERR_UNSUPPORTED otherwise.
template <int KS>
static int launch_xent(const float *x, long ldx, const _Float16 *hi, const _Float16 *lo, const float *winv, const float *bias, float *grad,
                       long ld, long M, int K, int N, const XentArgs &xa, hipStream_t s)
{
    dim3 grid((unsigned)((M + GH_BM - 1) / GH_BM)), block(GH_THREADS);
    hipLaunchKernelGGL((gemm_rows_f16x3_kernel<KS, true, SLK_ACT_LINEAR, GH_TRSTORE, 1>), grid, block, 0, s, x, ldx, hi, lo, winv, bias, grad, ld,
                       M, K, N, (float2 *)nullptr, xa);
    hipLaunchKernelGGL((gemm_rows_f16x3_kernel<KS, false, SLK_ACT_LINEAR, GH_TRSTORE, 2>), grid, block, 0, s, x, ldx, hi, lo, winv, bias, grad, ld,
                       M, K, N, (float2 *)nullptr, xa);
    return slk_launch_status();
}

extern "C" int slk_linear_xent_grad_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *W_inv_scale,
                                          const float *bias, float *grad, long ld, int K, int N, const int32_t *labels,
                                          const float *weights, int T, int B, int drop, float min_prob, float *loss_rows,
                                          float *correct_rows, float *xrow, slk_stream_t stream)
{
    if (!x || !W_hi || !W_lo || !W_inv_scale || !grad || !labels || !weights || !loss_rows || !correct_rows || !xrow || T < 1 || B < 1 ||
        K < 1 || N < 1 || ldx < K || ld < N || drop < 0 || 2 * drop >= T || !(min_prob >= 0.0f && min_prob < 1.0f) ||
        (reinterpret_cast<uintptr_t>(xrow) & 15) != 0)
        return SLK_ERR_INVALID_ARG;
    const long M = (long)T * B;
    if ((M + GH_BM - 1) / GH_BM > 0x7fffffffL || N > GH_BIASMAX || ld > ((long)(N + GH_BN - 1) / GH_BN) * GH_BN) return SLK_ERR_UNSUPPORTED;
    const _Float16 *hi = static_cast<const _Float16 *>(W_hi), *lo = static_cast<const _Float16 *>(W_lo);
    XentArgs xa{reinterpret_cast<float4 *>(xrow), labels, weights, loss_rows, correct_rows, T, B, drop, min_prob};
    hipStream_t s = slk_stream(stream);
    switch ((K + 15) / 16) {
    case 4: return launch_xent<4>(x, ldx, hi, lo, W_inv_scale, bias, grad, ld, M, K, N, xa, s);
    case 6: return launch_xent<6>(x, ldx, hi, lo, W_inv_scale, bias, grad, ld, M, K, N, xa, s);
    case 7: return launch_xent<7>(x, ldx, hi, lo, W_inv_scale, bias, grad, ld, M, K, N, xa, s);
    case 8: return launch_xent<8>(x, ldx, hi, lo, W_inv_scale, bias, grad, ld, M, K, N, xa, s);
    default: return SLK_ERR_UNSUPPORTED;
    }
}
