// gemm_bf16x6.hip -- y[M][N] = act(x[M][K] . W[N][K]^T + b) on the bf16 matrix pipe with float32-grade products, for the shapes
// gemm_rows_f16x3.hip does not take (K > 192): the dL/dx products of the training step (train.py: Softmax dx K = the padded logits
// row, N = 96; Gru dx K = 3n; Lstm dx K = 4n), which ran on the fp32 matrix pipe (gemm.hip, 108 TFLOP/s = 0.69 of its peak).
//
// Arithmetic: train.hip's bf16 scheme -- every float32 operand is cut into three bf16 pieces (8 significand bits each, float32's
// exponent range: gradients need no scaling) and a product is six v_mfma_f32_32x32x16_bf16 terms in float32 accumulators; what is
// dropped is below 2^-24 of the product.  Six bf16 MFMAs do the work of sixteen fp32 ones.
//
// Plan: gemm.hip's -- 128 rows x (32 NT) columns per 256-thread workgroup, each wave 32 rows x all columns, K staged through LDS in
// slabs of 32, several workgroups per CU hiding each other's loads (a one-wave-per-SIMD kernel with register prefetch reached
// 1.35 ms on the 1040-wide product against 1.50 here before: it issued its loads, cuts and MFMAs one after the other).  x goes
// through LDS as float32 and is cut by the wave that multiplies it (16 values per lane and slab); W arrives pre-cut
// (slk_pack_bf16x3_f32: [piece][N][K rounded up to 32] bf16, zero padded) so that nobody cuts it four times.
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define GB_BM 128                                        // rows per workgroup and row tile of its waves (WR row tiles: WR times as many)
#ifndef GB_BK
#define GB_BK 32                                         // (64: 256-byte runs per row and half the barriers, two workgroups per CU -- measured below)
#endif
#ifndef GB_WR
#define GB_WR 2                                          // row tiles per wave of the wide variant
#endif
#define GB_XLD (GB_BK + 4)                               // floats per x row in LDS (gemm.hip)
#define GB_WLD (GB_BK + 8)                               // bf16 per W row in LDS: the 16-byte reads of 16 rows hit 64 banks

// v = p1 + p2 + p3, each piece the top 16 bits of what is left (train.hip: split_bf16x3)
__device__ __forceinline__ void gb_split(const float (&v)[8], bf16x8 &p1, bf16x8 &p2, bf16x8 &p3)
{
    union { bf16x8 v; unsigned u[4]; } o1, o2, o3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // a piece = the top 16 bits of what is left; v_perm_b32 packs the top halves of a pair of values in one instruction (no
        // masking, shifting and or-ing of the pieces themselves: 11 instructions per pair instead of 16)
        float r1[2], r2[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float x = v[2 * j + e];
            r1[e] = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
            r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);
        }
        o1.u[j] = __builtin_amdgcn_perm(__float_as_uint(v[2 * j + 1]), __float_as_uint(v[2 * j]), 0x07060302u);
        o2.u[j] = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
        o3.u[j] = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
    }
    p1 = o1.v; p2 = o2.v; p3 = o3.v;
}

// packed[piece][n][KP] (bf16), KP = K rounded up to a whole slab: one thread per 8 consecutive k of one row
__global__ void __launch_bounds__(256) pack_bf16x3_kernel(const float *__restrict__ W, int N, int K, int KP, uint4 *__restrict__ packed)
{
    const long total = (long)N * (KP / 8);
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int n = (int)(e / (KP / 8)), k0 = (int)(e % (KP / 8)) * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = k0 + j < K ? W[(size_t)n * K + k0 + j] : 0.0f;
        union { bf16x8 v; uint4 q; } p1, p2, p3;
        gb_split(v, p1.v, p2.v, p3.v);
        const size_t piece = (size_t)N * (KP / 8);
        packed[e] = p1.q;
        packed[piece + e] = p2.q;
        packed[2 * piece + e] = p3.q;
    }
}

// WR: 32-row tiles per wave.  A wave reads the W fragments of a slab from LDS once per row tile group, and with one row tile the LDS
// moved 112 bytes per cycle of its 128 (4 waves x (2 KiB of x + 9 KiB of W fragments) per 16 k, plus the staging writes, against 18
// MFMAs of 32 cycles): the kernel was bound by the LDS port, not by the matrix pipe (0.28 of its peak).  With two row tiles the W
// fragments serve twice the MFMAs.
// With two row tiles the slab after the one being multiplied is requested before the multiplication and staged behind the barrier that
// ends it (the one-tile form requests a slab, waits, stages and multiplies it, and three workgroups per CU fill each other's waits:
// pipelined as well it needs 24 registers more than three waves per SIMD leave and was slower, 1.86 against 1.37 ms).
// 819200 x 1056 -> 96: 1.37 ms one tile, 1.43 two tiles unpipelined, 1.12 two tiles pipelined.
__device__ __forceinline__ float gb_dact(float v, int act)
{
    switch (act) {                                       // activation.py:8-57, as act_backward_kernel
    case SLK_ACT_TANH: return 1.0f - v * v;
    case SLK_ACT_SIGMOID: return v * (1.0f - v);
    case SLK_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
    case SLK_ACT_ELU: return v > 0.0f ? 1.0f : v + 1.0f;
    default: return 1.0f;
    }
}

template <int NT, int ACT, int WR>
__global__ void __launch_bounds__(256) gemm_bf16x6_kernel(const float *__restrict__ x, long ldx, const uint4 *__restrict__ wp,
                                                          const float *__restrict__ bias, float *__restrict__ y, long ldy, long M, int K,
                                                          int KP, int N, int ntile_n, const float *__restrict__ dref, long lddref, int dact)
{
    constexpr int BN = 32 * NT;
    constexpr int BM = GB_BM * WR;
    __shared__ __attribute__((aligned(16))) float xs[BM * GB_XLD];
    __shared__ __attribute__((aligned(16))) unsigned short ws[3 * BN * GB_WLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const long bid = blockIdx.x;
    const long m0 = (bid / ntile_n) * BM;
    const int n0 = (int)(bid % ntile_n) * BN;
    const size_t piece = (size_t)N * (KP / 8);           // uint4 per piece

    f32x16 acc[WR][NT];
#pragma unroll
    for (int w = 0; w < WR; w++)
#pragma unroll
        for (int i = 0; i < NT; i++)
#pragma unroll
            for (int j = 0; j < 16; j++) acc[w][i][j] = 0.0f;

    constexpr int XL = (BM * GB_BK / 4) / 256;           // float4 loads per thread for x: 4 per row tile
    constexpr int CPR = GB_BK / 8;                       // 16-byte chunks per row of a slab (x: twice as many)
    constexpr int WCH = 3 * BN * CPR;                    // 16-byte chunks of the W slab: [piece][row][CPR]
    constexpr int WL = (WCH + 255) / 256;
    constexpr bool PIPE = WR > 1;
    // (LDS counter only: __syncthreads() would also wait for the slab that has just been requested)
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // all loads back to back from clamped (always valid) addresses, masked when they are stored (gemm.hip)
    auto request = [&](int k0, float4 (&xv)[XL], uint4 (&wv)[WL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XL; i++) {
            const int idx = tid + 256 * i, row = idx / (2 * CPR), c4 = (idx % (2 * CPR)) * 4;
            const long gr = m0 + row;
            const int gk = k0 + c4;
            xv[i] = *reinterpret_cast<const float4 *>(x + (gr < M ? gr : M - 1) * ldx + (gk < K ? gk : 0));
        }
#pragma unroll
        for (int i = 0; i < WL; i++) {
            const int idx = min(tid + 256 * i, WCH - 1), p = idx / (BN * CPR), rem = idx % (BN * CPR), row = rem / CPR, ch = rem % CPR;
            const int gn = min(n0 + row, N - 1);
            wv[i] = wp[p * piece + (size_t)gn * (KP / 8) + min(k0 >> 3, KP / 8 - CPR) + ch];
        }
    };
    auto stage = [&](int k0, const float4 (&xv)[XL], const uint4 (&wv)[WL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < XL; i++) {
            const int idx = tid + 256 * i, row = idx / (2 * CPR), c4 = (idx % (2 * CPR)) * 4;
            const bool in = m0 + row < M && k0 + c4 < K;                              // K is a multiple of 4
            *reinterpret_cast<float4 *>(&xs[row * GB_XLD + c4]) = in ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < WL; i++) {
            const int idx = tid + 256 * i, p = idx / (BN * CPR), rem = idx % (BN * CPR), row = rem / CPR, ch = rem % CPR;
            if (idx < WCH) *reinterpret_cast<uint4 *>(&ws[(p * BN + row) * GB_WLD + 8 * ch]) = wv[i];   // (columns past N: never stored)
        }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < GB_BK / 16; ks++) {
            bf16x8 a1[WR], a2[WR], a3[WR];
#pragma unroll
            for (int w = 0; w < WR; w++) {
                float v[8];
                const float *ap = &xs[(32 * (WR * wave + w) + r) * GB_XLD + 16 * ks + 8 * h];
                const float4 lo4 = *reinterpret_cast<const float4 *>(ap), hi4 = *reinterpret_cast<const float4 *>(ap + 4);
                v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
                gb_split(v, a1[w], a2[w], a3[w]);
            }
            bf16x8 b[NT][3];
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int p = 0; p < 3; p++)
                    b[t][p] = *reinterpret_cast<const bf16x8 *>(&ws[(p * BN + 32 * t + r) * GB_WLD + 16 * ks + 8 * h]);
            // small terms first; term-major: consecutive MFMAs go to different accumulators
#define GB_TERM(PA, PB)                                                              \
            _Pragma("unroll") for (int w = 0; w < WR; w++) _Pragma("unroll") for (int t = 0; t < NT; t++) \
                acc[w][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PA[w], b[t][PB], acc[w][t], 0, 0, 0);
            GB_TERM(a1, 2) GB_TERM(a2, 1) GB_TERM(a3, 0) GB_TERM(a1, 1) GB_TERM(a2, 0) GB_TERM(a1, 0)
#undef GB_TERM
        }
    };
    if constexpr (PIPE) {
        float4 xv[XL];
        uint4 wv[WL];
        request(0, xv, wv);
        for (int k0 = 0; k0 < K; k0 += GB_BK) {
            stage(k0, xv, wv);
            lds_barrier();
            request(k0 + GB_BK, xv, wv);                 // (past the last slab: clamped addresses, never stored)
            multiply();
            lds_barrier();
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += GB_BK) {
            // (written out, not through request() / stage(): as arguments of those the arrays of this branch stayed in scratch memory)
            float4 xv[XL];
            uint4 wv[WL];
#pragma unroll
            for (int i = 0; i < XL; i++) {
                const int idx = tid + 256 * i, row = idx / (2 * CPR), c4 = (idx % (2 * CPR)) * 4;
                const long gr = m0 + row;
                const int gk = k0 + c4;
                xv[i] = *reinterpret_cast<const float4 *>(x + (gr < M ? gr : M - 1) * ldx + (gk < K ? gk : 0));
            }
#pragma unroll
            for (int i = 0; i < WL; i++) {
                const int idx = min(tid + 256 * i, WCH - 1), p = idx / (BN * CPR), rem = idx % (BN * CPR), row = rem / CPR, ch = rem % CPR;
                const int gn = min(n0 + row, N - 1);
                wv[i] = wp[p * piece + (size_t)gn * (KP / 8) + (k0 >> 3) + ch];
            }
#pragma unroll
            for (int i = 0; i < XL; i++) {
                const int idx = tid + 256 * i, row = idx / (2 * CPR), c4 = (idx % (2 * CPR)) * 4;
                const bool in = m0 + row < M && k0 + c4 < K;                              // K is a multiple of 4
                *reinterpret_cast<float4 *>(&xs[row * GB_XLD + c4]) = in ? xv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < WL; i++) {
                const int idx = tid + 256 * i, p = idx / (BN * CPR), rem = idx % (BN * CPR), row = rem / CPR, ch = rem % CPR;
                if (idx < WCH) *reinterpret_cast<uint4 *>(&ws[(p * BN + row) * GB_WLD + 8 * ch]) = wv[i];   // (columns past N: never stored)
            }
            __syncthreads();
            multiply();
            __syncthreads();
        }
    }
    // ---- epilogue: D[row = (reg&3) + 8*(reg>>2) + 4*h][col = lane&31] ----
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = n0 + 32 * nt + r;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.0f;
        // dL/dx of a layer whose input is the output `dref` of an element-wise activation: times fun'(.) written in terms of that
        // output (csrc/train.hip act_backward_kernel), so that the layer below receives dL/d(pre-activation).  All of a tile's
        // reference values are requested before the first store (from clamped rows: no branch around the loads).
        float dv[WR][16];
        if (dref) {
#pragma unroll
            for (int w = 0; w < WR; w++)
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    const long row = m0 + 32 * (WR * wave + w) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    dv[w][reg] = dref[(row < M ? row : M - 1) * lddref + col];
                }
        }
#pragma unroll
        for (int w = 0; w < WR; w++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const long row = m0 + 32 * (WR * wave + w) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float v = slk_act_t<ACT>(acc[w][nt][reg] + bv);
                if (dref) v *= gb_dact(dv[w][reg], dact);
                if (row < M) y[row * ldy + col] = v;
            }
    }
}

// include/sloika_amd.h
extern "C" size_t slk_pack_bf16x3_bytes(int N, int K)
{
    if (N < 1 || K < 1) return 0;
    return (size_t)3 * N * ((K + GB_BK - 1) / GB_BK * GB_BK) * 2;
}

extern "C" int slk_pack_bf16x3_f32(const float *W, int N, int K, void *packed, slk_stream_t stream)
{
    if (!W || !packed || N < 1 || K < 1) return SLK_ERR_INVALID_ARG;
    const int KP = (K + GB_BK - 1) / GB_BK * GB_BK;
    long blocks = ((long)N * (KP / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), W, N, K, KP,
                       static_cast<uint4 *>(packed));
    return slk_launch_status();
}

template <int NT>
static int gb_launch(const float *x, long ldx, const uint4 *wp, const float *bias, float *y, long ldy, long M, int K, int N, int act,
                     hipStream_t s, const float *dref = nullptr, long lddref = 0, int dact = 0)
{
    // two row tiles per wave for the 96-column products of a large batch (measured: 819200 x 1056 -> 96 1.37 -> 1.12 ms, x 288 -> 96
    // 0.39 -> 0.38; 64 columns 0.26 -> 0.34, so those and everything small stay with one tile)
    constexpr int WRMAX = NT == 3 ? GB_WR : 1;
    const bool wide = WRMAX > 1 && M >= (long)GB_BM * WRMAX * 1024 && K >= 512;    // (short rows: 0.38-0.44 against 0.39-0.40 ms, no gain)
    const int ntile_n = (N + 32 * NT - 1) / (32 * NT), KP = (K + GB_BK - 1) / GB_BK * GB_BK, bm = GB_BM * (wide ? WRMAX : 1);
    const long blocks = ((M + bm - 1) / bm) * ntile_n;
    if (blocks > 0x7fffffffL) return SLK_ERR_UNSUPPORTED;
#define GB_LAUNCH(AC)                                                                                                            \
    if (wide) hipLaunchKernelGGL((gemm_bf16x6_kernel<NT, AC, WRMAX>), dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, wp, bias, y, ldy, M, K, KP, N, \
                       ntile_n, dref, lddref, dact);                                                                             \
    else hipLaunchKernelGGL((gemm_bf16x6_kernel<NT, AC, 1>), dim3((unsigned)blocks), dim3(256), 0, s, x, ldx, wp, bias, y, ldy, M, K, KP, N, \
                       ntile_n, dref, lddref, dact)
    switch (act) {
    case SLK_ACT_LINEAR: GB_LAUNCH(SLK_ACT_LINEAR); break;
    case SLK_ACT_TANH: GB_LAUNCH(SLK_ACT_TANH); break;
    case SLK_ACT_SIGMOID: GB_LAUNCH(SLK_ACT_SIGMOID); break;
    default: return SLK_ERR_UNSUPPORTED;
    }
#undef GB_LAUNCH
    return slk_launch_status();
}

static int gb_entry(const float *x, long ldx, const void *packed, const float *bias, float *y, long ldy, long M, int K, int N, int act,
                    const float *dref, long lddref, int dact, slk_stream_t stream)
{
    if (!x || !packed || !y || M < 0 || K < 1 || N < 1 || ldx < K || ldy < N || !slk_act_valid(act)) return SLK_ERR_INVALID_ARG;
    if (dref && (lddref < N || !slk_act_valid(dact))) return SLK_ERR_INVALID_ARG;
    if (M == 0) return SLK_OK;
    if ((K & 3) || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(packed) & 15))
        return SLK_ERR_UNSUPPORTED;
    // the column-tile count that wastes the fewest MFMA columns (ties -> wider tile), as gemm.hip
    int best = 4, best_cost = 1 << 30;
    for (int nt = 4; nt >= 1; nt--) {                    // (four tiles: 128 columns in one block, x read once)
        const int cost = ((N + 32 * nt - 1) / (32 * nt)) * nt;
        if (cost < best_cost) { best_cost = cost; best = nt; }
    }
    hipStream_t s = slk_stream(stream);
    const uint4 *wp = static_cast<const uint4 *>(packed);
    switch (best) {
    case 1: return gb_launch<1>(x, ldx, wp, bias, y, ldy, M, K, N, act, s, dref, lddref, dact);
    case 2: return gb_launch<2>(x, ldx, wp, bias, y, ldy, M, K, N, act, s, dref, lddref, dact);
    case 3: return gb_launch<3>(x, ldx, wp, bias, y, ldy, M, K, N, act, s, dref, lddref, dact);
    default: return gb_launch<4>(x, ldx, wp, bias, y, ldy, M, K, N, act, s, dref, lddref, dact);
    }
}

extern "C" int slk_gemm_bias_act_bf16x6(const float *x, long ldx, const void *packed, const float *bias, float *y, long ldy, long M,
                                        int K, int N, int act, slk_stream_t stream)
{
    return gb_entry(x, ldx, packed, bias, y, ldy, M, K, N, act, nullptr, 0, 0, stream);
}

// out = (x . W^T) * fun'(.) with fun' written in terms of the OUTPUT `yref` of the activation below (tanh 1 - y^2, sigmoid y (1 - y),
// relu [y > 0], elu y > 0 ? 1 : y + 1, linear 1): dL/dx of a layer and slk_act_backward_f32 of the layer below it in one pass
// (the training step's Gru-over-Convolution boundary: one write and one read of [M][N] less).
extern "C" int slk_gemm_dact_bf16x6(const float *x, long ldx, const void *packed, const float *yref, long ldyref, int dact, float *out,
                                    long ldo, long M, int K, int N, slk_stream_t stream)
{
    if (!yref) return SLK_ERR_INVALID_ARG;
    if (dact != SLK_ACT_TANH && dact != SLK_ACT_SIGMOID && dact != SLK_ACT_RELU && dact != SLK_ACT_ELU && dact != SLK_ACT_LINEAR)
        return SLK_ERR_UNSUPPORTED;
    return gb_entry(x, ldx, packed, nullptr, out, ldo, M, K, N, SLK_ACT_LINEAR, yref, ldyref, dact, stream);
}
