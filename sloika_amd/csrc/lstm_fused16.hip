// lstm_fused16.hip -- a whole Lstm layer (sloika/layers.py:677-697) of up to 64 units and up to 64 inputs in one kernel: the scan
// of lstm_scan16.hip with the input projection vW = x.iW^T + b computed inside, four steps at a time, instead of being read from HBM.
//
// Why: models/baseline_lstm.py runs 4000 event steps per chunk; its four projections write and re-read 4 x 4.2 GB at batch 1024
// (the largest stage after the scans themselves), while x itself is 12 or 64 floats per step -- and the chain leaves the matrix pipe
// three quarters idle (16 MFMAs per step).
//
// Steps come in groups of four; while group G runs, the projection of group G+1 is made:
//   * step 0 requests the x rows of group G+2 (16 rows = 4 steps x 4 chunks, one float4 per thread) and fetches the operand image of
//     group G+1 from LDS together with the state;
//   * steps 1..3 each issue one term of the three-term split -- 4 gate tiles x KBLK MFMAs of iW (A operands in registers, rows scaled)
//     with that image -- right behind the barrier, where the matrix pipe otherwise waits for the LDS reads of the state;
//   * the tail of step 3 writes vW of the wave's 16 units for the four steps of group G+1 into an LDS buffer
//     [step][chunk][unit][gate] (a chain lane fetches its four gate inputs with one 16-byte read), then stages the x rows of group
//     G+2: row maximum over the 16 lanes of a row (DPP), power-of-two scale, fp16 hi / lo halves into the operand image in LDS whose
//     16 columns are (step, chunk);
//   * everything else is lstm_scan16.hip's step: one barrier, two MFMAs per recurrent product, cell state in registers.
#include <limits.h>

#include "bar16_common.h"

// An asm load the compiler does not track.  "+v": the destination counts as read by every request, so the register stays reserved
// while earlier requests into it are still in flight (a destination nobody reads afterwards would be handed to another value and
// overwritten when the data arrives).
__device__ __forceinline__ void lf_gload4(f32x4 &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// Two workgroups per CU (<= 256 registers): the two directions of a birnn, or two batches in flight, share the SIMDs and fill each
// other's barrier and LDS waits.
#ifndef LF_OCC
#define LF_OCC 2
#endif

template <int KBLK>
__global__ void __launch_bounds__(256, LF_OCC) lstm_fused16_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ iW,
                                                              const float *__restrict__ bias, const float *__restrict__ sW,
                                                              const float *__restrict__ peep, float *__restrict__ h_out, long ldh, int T,
                                                              int B, int insize, int n, int reverse, const int *__restrict__ lens)
{
    constexpr int N = 64, KBS = N / 32;
    constexpr int XIMG = KBLK * 4 * 16 * 4;              // dwords of one x operand image: [k block][k group][column = 4 step + chunk][4 dwords]

    // (the lo image 32 banks behind the hi image: gru_bar16.hip)
    constexpr int LO = 2 * N + (2 * N % 64 == 32 ? 0 : 32);
    __shared__ __attribute__((aligned(16))) unsigned h_img[2][LO + 2 * N];         // lstm_scan16.hip
    __shared__ __attribute__((aligned(16))) unsigned xop[2][2 * XIMG];              // [group parity][hi image | lo image]
    __shared__ __attribute__((aligned(16))) float xinv[2][16];                      // inverse row scales of the staged x rows
    __shared__ __attribute__((aligned(16))) float vbuf[2][4 * 4 * N * 4];           // [group parity][step][chunk][unit][gate]
    __shared__ __attribute__((aligned(16))) float prow_inv[4][N], prow_bias[4][N];  // input rows: inverse scale and bias, [gate][unit]

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < LO + 2 * N; i += 256) { h_img[0][i] = 0u; h_img[1][i] = 0u; }
    for (int i = tid; i < 2 * XIMG; i += 256) { xop[0][i] = 0u; xop[1][i] = 0u; }
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // ---- weights: recurrent (lstm_scan16.hip) and input (same tile order: row = 4 * unit + gate), rows scaled to [1, 2) ----
    half8 w_hi[4][KBS], w_lo[4][KBS], p_hi[4][KBLK], p_lo[4][KBLK];
    float inv[4];                                        // recurrent rows: inverse scale of MY unit's row
    {
        const int unit = 16 * w + (lane & 15);
        const bool uk = unit < n;
#pragma unroll
        for (int gt = 0; gt < 4; gt++) {
            const float *row = sW + (size_t)(4 * (uk ? unit : 0) + gt) * n;
            float v[KBS][8];
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBS; kb++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                    v[kb][j] = (uk && k < n) ? row[k] : 0.0f;
                    m = fmaxf(m, fabsf(v[kb][j]));
                }
            float iv;
            const float sc = pow2_scale(kgroup_max(m), iv);
            inv[gt] = __shfl(iv, 4 * g + q);
#pragma unroll
            for (int kb = 0; kb < KBS; kb++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a = v[kb][j] * sc;
                    const _Float16 h = (_Float16)a;
                    w_hi[gt][kb][j] = h;
                    w_lo[gt][kb][j] = (_Float16)(a - (float)h);
                }
            // input weights of the same row
            const float *prow = iW + (size_t)(4 * (uk ? unit : 0) + gt) * insize;
            float pv[KBLK][8];
            float pm = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                    pv[kb][j] = (uk && k < insize) ? prow[k] : 0.0f;
                    pm = fmaxf(pm, fabsf(pv[kb][j]));
                }
            float piv;
            const float psc = pow2_scale(kgroup_max(pm), piv);
            const float bv = (uk && bias) ? bias[4 * unit + gt] : 0.0f;
            if (g == 0) { prow_inv[gt][unit] = piv; prow_bias[gt][unit] = bv; }
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a = pv[kb][j] * psc;
                    const _Float16 h = (_Float16)a;
                    p_hi[gt][kb][j] = h;
                    p_lo[gt][kb][j] = (_Float16)(a - (float)h);
                }
        }
    }
    int moff[KBS];
#pragma unroll
    for (int kb = 0; kb < KBS; kb++) moff[kb] = (q >> 1) * LO + ((kb * 4 + g) * 4 + c) * 4;
    const int u0 = 16 * w + 4 * g + q;
    const bool uok = u0 < n;
    const float p0 = (peep && uok) ? peep[u0] : 0.0f, p1 = (peep && uok) ? peep[n + u0] : 0.0f, p2 = (peep && uok) ? peep[2 * n + u0] : 0.0f;
    const int wdw = (((w >> 1) * 4 + g) * 4 + c) * 4 + q;

    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
    const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
    float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldh + (uok ? u0 : 0);

    // ---- staging role: thread = (row = tid >> 4 = 4 * step + chunk, piece i = tid & 15: floats 4i .. 4i+3 of that x row) ----
    const int srow = tid >> 4, sj = srow >> 2, sc_ = srow & 3, si = tid & 15;
    const bool sact = 4 * si < insize;                   // my piece exists
    const int sb = min(b0 + sc_, B - 1);
    const int sTc = (lens && b0 + sc_ < B) ? min(max(lens[b0 + sc_], 1), T) : T;
    // x row of scan step s for chunk sb: time (reverse ? sTc-1-s : s), clamped into the chunk (steps past its end are never stored)
    auto xoff = [&](int s) {
        const int sc = min(s, sTc - 1);
        const unsigned t = reverse ? sTc - 1 - sc : sc;  // (the launcher refuses inputs of 4 GiB and more)
        return ((t * (unsigned)B + (unsigned)sb) * (unsigned)ldx + (sact ? 4u * si : 0u)) * 4u;
    };
    // my four halves of the image: k = 4 si + e -> k block si >> 3, half (si >> 2) & 1, k group si & 3, dword r = e
    const int xw = ((((si >> 3) * 4 + (si & 3)) * 16 + srow) * 4) * 2 + ((si >> 2) & 1);     // in halves; + 2 e
    const int xb = (g * 16 + (lane & 15)) * 4;            // projection B operand of K block kb: + kb * 256 dwords

    auto stage = [&](const f32x4 &xv, int par) {         // x rows of a group -> operand image `par`
        f32x4 v = sact ? xv : f32x4{0.f, 0.f, 0.f, 0.f};
        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x128, 0xf, 0xf, false)));   // row_ror:8
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x124, 0xf, 0xf, false)));   // row_ror:4
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x122, 0xf, 0xf, false)));   // row_ror:2
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m), 0x121, 0xf, 0xf, false)));   // row_ror:1
        float iv;
        const float sc = pow2_scale(m, iv);
        if (si == 0) xinv[par][srow] = iv;
        if (4 * si < 32 * KBLK) {
            unsigned short *ih = reinterpret_cast<unsigned short *>(&xop[par][0]) + xw;
            unsigned short *il = reinterpret_cast<unsigned short *>(&xop[par][XIMG]) + xw;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float a = v[e] * sc;
                asm volatile("" : "+v"(a));              // split2's note on v_fma_mixlo_f16 applies
                const _Float16 hh = (_Float16)a;
                const _Float16 hl = (_Float16)(a - (float)hh);
                ih[2 * e] = __builtin_bit_cast(unsigned short, hh);
                il[2 * e] = __builtin_bit_cast(unsigned short, hl);
            }
        }
    };
    // vW of a group from its operand image: accumulators -> [step][chunk][unit][gate] in LDS (lane (g, j, c): column 4j + c, rows 4g + r)
    f32x4 pacc[4];
    half8 xh[KBLK], xl[KBLK];
    float xs = 0.0f;
    auto project_fetch = [&](int par) {                  // operand image and my column's inverse scale -> registers
#pragma unroll
        for (int kb = 0; kb < KBLK; kb++) { xh[kb] = ldH(&xop[par][0], xb + 256 * kb); xl[kb] = ldH(&xop[par][XIMG], xb + 256 * kb); }
        xs = xinv[par][lane & 15];
    };
    auto project_term = [&](auto TC) {                   // term 0: hi.lo (starts the sums), 1: lo.hi, 2: hi.hi
        constexpr int term = decltype(TC)::value;
#pragma unroll
        for (int kb = 0; kb < KBLK; kb++)
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                if constexpr (term == 0) {
                    if (kb == 0) pacc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    pacc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[gt][kb], xl[kb], pacc[gt], 0, 0, 0);
                } else if constexpr (term == 1) {
                    pacc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_lo[gt][kb], xh[kb], pacc[gt], 0, 0, 0);
                } else {
                    pacc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p_hi[gt][kb], xh[kb], pacc[gt], 0, 0, 0);
                }
            }
        // the last MFMA that reads an operand must not be given that operand's registers as its destination
        // (tools/mfma_overlap_scan.py): the operand stays live until the sums exist
        if constexpr (term == 0) {
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++)
                asm volatile("" : "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3]) : "v"(xl[kb]));
        }
        if constexpr (term == 2) {
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++)
                asm volatile("" : "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3]) : "v"(xh[kb]));
        }
    };
    auto project_out = [&](int par) {
        const int col = lane & 15;                       // = 4 step + chunk
        float *dst = &vbuf[par][(((col >> 2) * 4 + (col & 3)) * N + 16 * w + 4 * g) * 4];
        f32x4 o[4];
#pragma unroll
        for (int gt = 0; gt < 4; gt++) {                 // my rows are the units 16w + 4g + r
            const f32x4 pinv = *reinterpret_cast<const f32x4 *>(&prow_inv[gt][16 * w + 4 * g]);
            const f32x4 pbias = *reinterpret_cast<const f32x4 *>(&prow_bias[gt][16 * w + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; r++) o[r][gt] = fmaf(pacc[gt][r] * xs, pinv[r], pbias[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) *reinterpret_cast<f32x4 *>(dst + 4 * r) = o[r];
    };

    // ---- groups 0 (projected) and 1 (staged) before the loop.  Later groups: ONE register, requested FOUR times in a row (the same
    // address) at step 0 of group G-2 and staged at the end of its step 3 behind s_waitcnt vmcnt(3): loads complete in order, so "at
    // most three outstanding" means the first of the four has delivered the rows whatever the h stores issued since do (the counter
    // counts them too, and they may complete late), and with up to three stores in flight the wait needs nothing younger than those
    // four loads to have completed; the other three rewrite the same values. ----
    f32x4 xr = {0.f, 0.f, 0.f, 0.f};
    lf_gload4(xr, xoff(sj), x);                          // group 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(xr));
    __syncthreads();                                     // LDS initialised
    stage(xr, 0);
    lf_gload4(xr, xoff(4 + sj), x);                      // group 1
    __syncthreads();
    project_fetch(0);
    project_term(ic<0>{});
    project_term(ic<1>{});
    project_term(ic<2>{});
    asm volatile("" : "+v"(pacc[0]), "+v"(pacc[1]), "+v"(pacc[2]), "+v"(pacc[3]) : "v"(xl[0]), "v"(xl[KBLK - 1]), "v"(xh[0]), "v"(xh[KBLK - 1]));
    project_out(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(xr));
    stage(xr, 1);

    float cell = 0.0f;
    auto step = [&](auto PHC, const int s) {
        constexpr int ph = decltype(PHC)::value;
        constexpr int par = ph & 1;
        const int G = s >> 2, gp = G & 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bm[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) bm[kb] = ldH(h_img[par], moff[kb]);
        const f32x4 cur = *reinterpret_cast<const f32x4 *>(&vbuf[gp][((ph * 4 + c) * N + u0) * 4]);
        if constexpr (ph == 0) {
            project_fetch(gp ^ 1);                       // group G+1, staged at the end of the previous group
            const unsigned xo = xoff(4 * (G + 2) + sj);  // group G+2
#pragma unroll
            for (int k = 0; k < 4; k++) lf_gload4(xr, xo, x);
        } else {
            project_term(ic<ph - 1>{});                  // in the shadow of the LDS reads above
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 acc[4];
#pragma unroll
        for (int gt = 0; gt < 4; gt++) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
            for (int gt = 0; gt < 4; gt++) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[gt][kb], bm[kb], acc[gt], 0, 0, 0);
#pragma unroll
            for (int gt = 0; gt < 4; gt++) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[gt][kb], bm[kb], acc[gt], 0, 0, 0);
        }
        // pick_mix reads the accumulators from asm; the state operands stay live until the sums exist (see project_term)
        // (and the projection operand last read just before these MFMAs until they have their destinations)
        if constexpr (ph == 1)
            asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                         : "v"(bm[0]), "v"(bm[KBS - 1]), "v"(xl[0]), "v"(xl[KBLK - 1]));
        else if constexpr (ph == 3)
            asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                         : "v"(bm[0]), "v"(bm[KBS - 1]), "v"(xh[0]), "v"(xh[KBLK - 1]));
        else
            asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : "v"(bm[0]), "v"(bm[KBS - 1]));
        // layers.py:686-691
        const float a0 = fmaf(pick_mix(acc[0]), inv[0], cur[0]), a1 = fmaf(pick_mix(acc[1]), inv[1], cur[1]);
        const float a2 = fmaf(pick_mix(acc[2]), inv[2], cur[2]), a3 = fmaf(pick_mix(acc[3]), inv[3], cur[3]);
        const float fg = sigmoid4(fmaf(cell, p1, a2));
        const float ig = sigmoid4(fmaf(cell, p0, a1));
        const float cn = uok ? fmaf(fg, cell, tanh5(a0) * ig) : 0.0f;
        const float og = sigmoid4(fmaf(cn, p2, a3));
        const float hn = uok ? tanh5(cn) * og : 0.0f;
        cell = cn;
        {
            float hv = hn;
            asm volatile("" : "+v"(hv));
            const _Float16 hh = (_Float16)hv;
            const _Float16 hl = (_Float16)(hv - (float)hh);
            reinterpret_cast<unsigned short *>(&h_img[par ^ 1][wdw])[w & 1] = __builtin_bit_cast(unsigned short, hh);
            reinterpret_cast<unsigned short *>(&h_img[par ^ 1][LO + wdw])[w & 1] = __builtin_bit_cast(unsigned short, hl);
        }
        if constexpr (ph == 3) {
            project_out(gp ^ 1);                         // vW of group G+1: visible to everybody behind the next barrier
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            asm volatile("" : "+v"(xr));
            stage(xr, gp);                               // x rows of group G+2 (that image was last read a whole group ago)
        }
        if (live && s < Tc && uok) hp[0] = hn;
        hp += hstep;
    };
    for (int s = 0; s < T; s += 4) {
        step(ic<0>{}, s);
        step(ic<1>{}, s + 1);                            // (steps past T compute on clamped rows and store nothing)
        step(ic<2>{}, s + 2);
        step(ic<3>{}, s + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LF_OCC 1 only: dynamic LDS that keeps a second workgroup off the CU (a launch of <= 256 workgroups then takes one CU each)
template <int KBLK>
static size_t lstm_fused16_exclusive_lds()
{
    if (LF_OCC > 1) return 0;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(lstm_fused16_kernel<KBLK>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_fused16_kernel<KBLK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

// include/sloika_amd.h
extern "C" int slk_lstm_fused16_f32(const float *x, long ldx, const float *iW, const float *sW, const float *bias, const float *p, float *y,
                                    long ldy, int T, int B, int insize, int n, int reverse, int act, int gate_act, const int32_t *lens,
                                    slk_stream_t stream)
{
    if (!x || !iW || !sW || !y || T < 1 || B < 1 || n < 1 || insize < 1 || ldx < insize || ldy < n) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if (n % 16 || n > 64 || insize > 64 || insize % 4 || ldx % 4 || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;
    if ((unsigned long long)T * B * ldx * sizeof(float) >= (1ull << 32)) return SLK_ERR_UNSUPPORTED;          // 32-bit lane offsets
    hipStream_t s = slk_stream(stream);
    if (insize <= 32) {
        const size_t dyn = SLK_PER_DEVICE(size_t, lstm_fused16_exclusive_lds<1>());
        hipLaunchKernelGGL((lstm_fused16_kernel<1>), dim3((B + 3) / 4), dim3(256), dyn, s, x, ldx, iW, bias, sW, p, y, ldy, T, B, insize, n,
                           reverse & 1, lens);
    } else {
        const size_t dyn = SLK_PER_DEVICE(size_t, lstm_fused16_exclusive_lds<2>());
        hipLaunchKernelGGL((lstm_fused16_kernel<2>), dim3((B + 3) / 4), dim3(256), dyn, s, x, ldx, iW, bias, sW, p, y, ldy, T, B, insize, n,
                           reverse & 1, lens);
    }
    return slk_launch_status();
}
