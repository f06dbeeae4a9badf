// gru_scan16.hip -- the scan of Gru.step (sloika/layers.py:1010-1021) for layers too wide for the fused kernels (n = 112 / 128:
// models/pretrained.pkl, models/raw_1.00_rGr.py), on the execution plan of gru_bar16.hip: four waves per workgroup, one per
// SIMD, two s_barrier per step, recurrent products as 3-term fp16 splits on v_mfma_f32_16x16x32_f16, states exchanged as packed
// hi/lo halves through LDS, a lane owning one (neuron, chunk) pair per tile.
//
// What is different: at n = 128 all four waves are chain waves (32 neurons each, four 32-wide K blocks), so nobody is left to
// compute the input projection -- it comes from HBM (vI = x.iW^T + b, written by the row GEMM), read by the lane that needs it
// two steps ahead -- and the weights of a wave (2 tiles x 3 gates x 4 K blocks x hi/lo = 192 registers) do not fit next to
// its working set: the update gate's and the candidate's live in accumulation registers and their MFMAs are asm (bar16_common.h).
// Sizes below 128 (a multiple of 16) run with zero weights for the missing neurons: their state stays exactly 0.
//
// n = 144 (the middle layer of models/pretrained.pkl; 142 padded, models/raw_1.00_rGr.py) is the N = 160 instantiation: five K
// blocks, the same four chain waves for neurons 0..127, and a NINTH tile (neurons 128..143) whose three gates are spread over
// three waves -- wave 0 the reset gate (it writes r*h of those neurons), wave 1 the update gate (its z goes to wave 3 through
// LDS, in time because z only needs h(s-1)), wave 3 the candidate and the new state (which it hands back to wave 0 as float32).
// The tile's weights (3 gates x 5 K blocks x hi/lo, 30 KB) live in LDS and are fetched as A operands when they are needed, every
// register being taken.  105 MFMAs per step on the three waves with a share, 90 on wave 2.
#include <limits.h>
#include <stdlib.h>

#include "bar16_common.h"

// 1: the update gate's products with the wave's own K block right behind the reset gate's, inside the LDS round trip (0: behind the
// reset gate's other blocks; measured equal)
// timing experiments (results garbage): 1 = do not wait for the projection loads, 2 = no stores, 4 = no projection loads at all
#ifndef SCAN16_ABL
#define SCAN16_ABL 0
#endif
#ifndef SCAN16_Z0_EARLY
#define SCAN16_Z0_EARLY 1
#endif
// 1: two MFMAs per recurrent product, the state's hi and lo halves in different column groups (bar16_common.h, mfma2x2 / pick_mix);
// 0: the three-term sequence of round 2 (every column group a copy of the hi half, the lo half a second operand)
#ifndef SCAN16_MIX
#define SCAN16_MIX 1
#endif
// one float per lane from HBM, not tracked by the compiler: the caller counts (s_waitcnt vmcnt(n), then pin_f)
__device__ __forceinline__ void gload1(float &dst, const float *src) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(src) : "memory"); }
// the same with the row's base address in scalar registers and the lane's part as an unsigned 32-bit byte offset: nothing but one
// add per step is left of the address arithmetic
__device__ __forceinline__ void gload1_s(float &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void pin_f(float &v) { asm volatile("" : "+v"(v)); }

template <int N>
__global__ void __launch_bounds__(256, 1) gru_scan16_kernel(const float *__restrict__ vI, long ldv, const float *__restrict__ sW,
                                                            const float *__restrict__ sW2, float *__restrict__ h_out, long ldh, int T,
                                                            int B, int n, int reverse, const int *__restrict__ lens)
{
    static_assert(N == 128 || N == 160, "four chain waves of 32 neurons (+ a ninth tile on wave 3)");
    constexpr int KBS = N / 32;
    constexpr bool T9 = N == 160;
    constexpr int NV = T9 ? 3 : 2;                       // (neuron, chunk) pairs a lane requests per step

    constexpr bool MIX = SCAN16_MIX != 0;
    // hi image, then lo image 32 banks behind it (gru_bar16.hip)
    constexpr int LO = 2 * N + (2 * N % 64 == 32 ? 0 : 32);
    __shared__ __attribute__((aligned(16))) unsigned h_img[LO + 2 * N], rh_img[LO + 2 * N];
    unsigned *const h_hi = h_img, *const h_lo = h_img + LO, *const rh_hi = rh_img, *const rh_lo = rh_img + LO;
    // the ninth tile's A operands: [gate r, z, c][K block][hi, lo][lane] x 16 bytes
    __shared__ __attribute__((aligned(16))) unsigned w9[T9 ? 3 * KBS * 2 * 64 * 4 : 4];
    __shared__ float h9f[64], z9f[64];                   // ninth tile: h(s-1) for the reset-gate wave, z(s) for the owner

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < 2 * N; i += 256) { h_hi[i] = 0u; h_lo[i] = 0u; rh_hi[i] = 0u; rh_lo[i] = 0u; }      // h(-1) = 0
    if (tid < 64) { h9f[tid] = 0.0f; z9f[tid] = 0.0f; }
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // recurrent weights: A operands, K blocks in the rotated order w, w+1, ... (element (g, j) of block kb is neuron
    // 32 kb + 16 (j&1) + 4 g + (j>>1), the order the owners' packed writes create), rows scaled to [1, 2)
    half8 wz_hi[2][KBS], wz_lo[2][KBS], wr_hi[2][KBS], wr_lo[2][KBS], wc_hi[2][KBS], wc_lo[2][KBS];
    float inv_z[2], inv_r[2], inv_c[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const int row = 32 * w + 16 * p + (lane & 15);
        const bool rok = row < n;
        float vz[KBS][8], vr[KBS][8], vc[KBS][8];
        float mz = 0.0f, mr = 0.0f, mc = 0.0f;
#pragma unroll
        for (int i = 0; i < KBS; i++) {
            const int kb = (w + i) % KBS;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                const bool ok = rok && k < n;
                vz[i][j] = ok ? sW[(size_t)row * n + k] : 0.0f;
                vr[i][j] = ok ? sW[(size_t)(n + row) * n + k] : 0.0f;
                vc[i][j] = ok ? sW2[(size_t)row * n + k] : 0.0f;
                mz = fmaxf(mz, fabsf(vz[i][j])); mr = fmaxf(mr, fabsf(vr[i][j])); mc = fmaxf(mc, fabsf(vc[i][j]));
            }
        }
        float iz, ir, ic_;
        const float sz = pow2_scale(kgroup_max(mz), iz), sr = pow2_scale(kgroup_max(mr), ir), sc = pow2_scale(kgroup_max(mc), ic_);
        inv_z[p] = __shfl(iz, 4 * g + q); inv_r[p] = __shfl(ir, 4 * g + q); inv_c[p] = __shfl(ic_, 4 * g + q);
#pragma unroll
        for (int i = 0; i < KBS; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float az = vz[i][j] * sz, ar = vr[i][j] * sr, ac = vc[i][j] * sc;
                const _Float16 hz = (_Float16)az, hr = (_Float16)ar, hc = (_Float16)ac;
                wz_hi[p][i][j] = hz; wz_lo[p][i][j] = (_Float16)(az - (float)hz);
                wr_hi[p][i][j] = hr; wr_lo[p][i][j] = (_Float16)(ar - (float)hr);
                wc_hi[p][i][j] = hc; wc_lo[p][i][j] = (_Float16)(ac - (float)hc);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int i = 0; i < KBS; i++) {
            wz_hi[p][i] = to_acc_regs(wz_hi[p][i]); wz_lo[p][i] = to_acc_regs(wz_lo[p][i]);
            wc_hi[p][i] = to_acc_regs(wc_hi[p][i]); wc_lo[p][i] = to_acc_regs(wc_lo[p][i]);
            if constexpr (T9) {     // five K blocks: the reset gate's weights too (240 of the 256), or in-flight vI registers get spilled
                wr_hi[p][i] = to_acc_regs(wr_hi[p][i]); wr_lo[p][i] = to_acc_regs(wr_lo[p][i]);
            }
        }
    }
    // duty of this wave for the ninth tile: gate 0 (r) on wave 0, 1 (z) on wave 1, 2 (c) on wave 3; -1: none
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int duty = !T9 ? -1 : wu == 0 ? 0 : wu == 1 ? 1 : wu == 3 ? 2 : -1;
    const bool w3 = wu == 3;
    float inv9 = 1.0f;
    auto w9at = [&](int gate, int kb, int hl) { return reinterpret_cast<half8 *>(w9 + (((gate * KBS + kb) * 2 + hl) * 64 + lane) * 4); };
    if constexpr (T9) {
        if (duty >= 0) {
            const int row = 128 + (lane & 15);
            const bool rok = row < n;
            const float *src = duty == 0 ? sW + (size_t)(n + row) * n : duty == 1 ? sW + (size_t)row * n : sW2 + (size_t)row * n;
            float v[KBS][8];
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                    v[kb][j] = (rok && k < n) ? src[k] : 0.0f;
                    m = fmaxf(m, fabsf(v[kb][j]));
                }
            }
            float iv;
            const float sc = pow2_scale(kgroup_max(m), iv);
            inv9 = __shfl(iv, 4 * g + q);
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
                half8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a = v[kb][j] * sc;
                    const _Float16 h = (_Float16)a;
                    hi[j] = h;
                    lo[j] = (_Float16)(a - (float)h);
                }
                *w9at(duty, kb, 0) = hi;
                *w9at(duty, kb, 1) = lo;
            }
        }
    }
    int boff[KBS];
#pragma unroll
    for (int i = 0; i < KBS; i++) boff[i] = ((((w + i) % KBS) * 4 + g) * 4 + c) * 4;        // in dwords
    int moff[KBS];                                       // MIX: my column group's image (q = 0, 1: hi; q = 2, 3: lo)
#pragma unroll
    for (int i = 0; i < KBS; i++) moff[i] = (q >> 1) * LO + boff[i];
    auto pick = [&](const f32x4 &a) { if constexpr (MIX) return pick_mix(a); else return sel4(a, q); };
    const int wd = ((w * 4 + g) * 4 + c) * 4 + q;                                           // my packed pair, in dwords
    const int n0 = 32 * w + 4 * g + q;                                                      // my neuron of tile 2w (+16: 2w+1)
    const bool nok0 = n0 < n, nok1 = n0 + 16 < n;
    const int n9 = 128 + 4 * g + q;                                                         // ninth tile: my neuron (wave 3)
    const bool nok9 = T9 && n9 < n;
    const int wd9 = ((4 * 4 + g) * 4 + c) * 4 + q;
    // my chunk's rows (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
    const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
    float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldh + n0;
    // vI of step s for my two (neuron, chunk) pairs: z | r | c blocks of n floats per row; steps past the chunk's end re-read its
    // last row (their results are never stored), neurons past n are zero
    // Four register sets, step s uses set s % 4 and requests step s + 3 into the set step s - 1 used: no copies of values still in
    // flight.  The loads are asm (the compiler would wait for ALL outstanding memory operations at the first use -- the wave also has
    // stores in flight); loads complete in order among themselves, so once at most 18 operations are outstanding (the loads of the
    // three younger steps) those of the current step have arrived, whatever the stores do.
    struct VI { float z[NV], r[NV], c[NV]; };
    VI vs[4];
    // addresses: three wave-uniform bases (z | r | c block of row 0) + this lane's byte offset of the step being requested, which
    // advances by one time step per request until the chunk's last row (requests past it re-read that row); the caller refuses
    // projections of 4 GiB and more
    const float *sb_z = vI, *sb_r = vI + n, *sb_c = vI + 2 * n;
    const unsigned tile1 = nok1 ? 64u : 0u;                                       // bytes to my neuron of tile 2w+1
    const unsigned tile9 = (nok9 && nok0) ? (unsigned)(128 - 32 * w) * 4u : 0u;  // ... of the ninth tile (every wave requests it: one count)
    unsigned voff = (unsigned)((((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldv + (nok0 ? n0 : 0)) * sizeof(float));
    const unsigned vstep = (unsigned)((size_t)B * ldv * sizeof(float));
    int vnext = 0;                                                                // step the next request is for
    auto load_vi = [&](int, VI &v) {
        gload1_s(v.z[0], voff, sb_z); gload1_s(v.r[0], voff, sb_r); gload1_s(v.c[0], voff, sb_c);
        gload1_s(v.z[1], voff + tile1, sb_z); gload1_s(v.r[1], voff + tile1, sb_r); gload1_s(v.c[1], voff + tile1, sb_c);
        if constexpr (T9) { gload1_s(v.z[2], voff + tile9, sb_z); gload1_s(v.r[2], voff + tile9, sb_r); gload1_s(v.c[2], voff + tile9, sb_c); }
        vnext++;
        if (vnext < Tc) voff = reverse ? voff - vstep : voff + vstep;
    };
    load_vi(0, vs[0]);
    load_vi(1, vs[1]);
    load_vi(2, vs[2]);

    __syncthreads();                                     // LDS initialised
    float hold[2] = {0.0f, 0.0f};
    float hold9 = 0.0f;
    // one gate of the ninth tile: sum over the K blocks of the 3-term split, A operands from LDS (slot i <-> my block order)
    auto tile9_mfma = [&](int gate, const half8 *bhh, const half8 *bll) __attribute__((always_inline)) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        half8 ah = *w9at(gate, wu, 0), al = *w9at(gate, wu, 1);                     // my operand order starts at my own K block
#pragma unroll
        for (int i = 0; i < KBS; i++) {
            half8 nh = ah, nl = al;
            if (i + 1 < KBS) {                                                      // in flight under this block's MFMAs
                const int kb = (wu + i + 1) % KBS;
                nh = *w9at(gate, kb, 0);
                nl = *w9at(gate, kb, 1);
            }
            if constexpr (MIX) {                                                    // bhh = the mixed operands, bll unused
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bhh[i], acc, 0, 0, 0);
                if (i == 0) asm volatile("" : "+v"(acc) : "v"(ah), "v"(al), "v"(bhh[0]));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bhh[i], acc, 0, 0, 0);
                ah = nh;
                al = nl;
                continue;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bll[i], acc, 0, 0, 0);
            if (i == 0) asm volatile("" : "+v"(acc) : "v"(ah), "v"(al), "v"(bll[0]), "v"(bhh[0]));     // see gemm_rows_f16x3.hip
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bhh[i], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bhh[i], acc, 0, 0, 0);
            ah = nh;
            al = nl;
        }
        return acc;
    };
    half8 oh = {0, 0, 0, 0, 0, 0, 0, 0}, ol = {0, 0, 0, 0, 0, 0, 0, 0};      // my own K block of h(s-1) as B operand
    settle(oh);
    settle(ol);
    auto step = [&](auto PHC, const int s) {
        constexpr int ph = decltype(PHC)::value;
        VI &cur = vs[ph];
        // ------------------------------ interval A ------------------------------
        if constexpr (MIX) lds_bar_1read(); else lds_bar_2reads();
        half8 bh[KBS], bl[KBS];                          // MIX: bh = the mixed operands, bl unused
        bh[0] = oh;
        bl[0] = ol;
#pragma unroll
        for (int i = 1; i < KBS; i++) {
            if constexpr (MIX) bh[i] = ldH(h_img, moff[i]);
            else { bh[i] = ldH(h_hi, boff[i]); bl[i] = ldH(h_lo, boff[i]); }
        }
        if constexpr (!(SCAN16_ABL & 4)) load_vi(s + 3, vs[(ph + 3) & 3]);                // three steps ahead
        __builtin_amdgcn_sched_barrier(0);
        f32x4 accR[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accZ[2], accC[2];
        if constexpr (MIX) {
            if constexpr (T9) z_block_mfma2<true>(accR[0], accR[1], wr_hi[0][0], wr_lo[0][0], wr_hi[1][0], wr_lo[1][0], bh[0]);
            else mfma2x2(wr_hi[0][0], wr_lo[0][0], wr_hi[1][0], wr_lo[1][0], bh[0], accR[0], accR[1]);
        } else {
            if constexpr (T9) z_block_mfma<true>(accR[0], accR[1], wr_hi[0][0], wr_lo[0][0], wr_hi[1][0], wr_lo[1][0], bh[0], bl[0]);
            else mfma3x2(wr_hi[0][0], wr_lo[0][0], wr_hi[1][0], wr_lo[1][0], bh[0], bl[0], accR[0], accR[1]);
        }
#if SCAN16_Z0_EARLY
        if constexpr (MIX) z_block_mfma2<true>(accZ[0], accZ[1], wz_hi[0][0], wz_lo[0][0], wz_hi[1][0], wz_lo[1][0], bh[0]);
        else z_block_mfma<true>(accZ[0], accZ[1], wz_hi[0][0], wz_lo[0][0], wz_hi[1][0], wz_lo[1][0], bh[0], bl[0]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 1; i < KBS; i++) { keep(bh[i]); if constexpr (!MIX) keep(bl[i]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < KBS; i++) {
            if constexpr (MIX) {
                if constexpr (T9) z_block_mfma2<false>(accR[0], accR[1], wr_hi[0][i], wr_lo[0][i], wr_hi[1][i], wr_lo[1][i], bh[i]);
                else mfma2x2(wr_hi[0][i], wr_lo[0][i], wr_hi[1][i], wr_lo[1][i], bh[i], accR[0], accR[1]);
            } else {
                if constexpr (T9) z_block_mfma<false>(accR[0], accR[1], wr_hi[0][i], wr_lo[0][i], wr_hi[1][i], wr_lo[1][i], bh[i], bl[i]);
                else mfma3x2(wr_hi[0][i], wr_lo[0][i], wr_hi[1][i], wr_lo[1][i], bh[i], bl[i], accR[0], accR[1]);
            }
        }
        f32x4 acc9 = {0.f, 0.f, 0.f, 0.f};
        float h9prev = 0.0f;
        if constexpr (T9) {
            if (duty == 0) h9prev = h9f[lane];                                     // written by wave 3 before the barrier that ended step s-1
            if (duty == 0 || duty == 1) acc9 = tile9_mfma(duty, bh, bl);
        }
        __builtin_amdgcn_sched_barrier(0);
        // z products of blocks 0 .. KBS-2 under the r epilogue
#if !SCAN16_Z0_EARLY
        if constexpr (MIX) z_block_mfma2<true>(accZ[0], accZ[1], wz_hi[0][0], wz_lo[0][0], wz_hi[1][0], wz_lo[1][0], bh[0]);
        else z_block_mfma<true>(accZ[0], accZ[1], wz_hi[0][0], wz_lo[0][0], wz_hi[1][0], wz_lo[1][0], bh[0], bl[0]);
#endif
        static_for<1, KBS - 1>([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            if constexpr (MIX) z_block_mfma2<false>(accZ[0], accZ[1], wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh[i]);
            else z_block_mfma<false>(accZ[0], accZ[1], wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh[i], bl[i]);
        });
        if constexpr (!(SCAN16_ABL & 5)) {                                                                  // this step's vI (see above)
            if constexpr (T9) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        }
        pin_f(cur.z[0]); pin_f(cur.z[1]); pin_f(cur.r[0]); pin_f(cur.r[1]); pin_f(cur.c[0]); pin_f(cur.c[1]);
        if constexpr (T9) { pin_f(cur.z[2]); pin_f(cur.r[2]); pin_f(cur.c[2]); }
        if constexpr (T9) { mfma_drain(accR[0]); mfma_drain(accR[1]); }       // asm MFMAs: the compiler keeps no distance for them
        float rr[2];
        rr[0] = nok0 ? sigmoid4(fmaf(pick(accR[0]), inv_r[0], cur.r[0])) : 0.0f;
        rr[1] = nok1 ? sigmoid4(fmaf(pick(accR[1]), inv_r[1], cur.r[1])) : 0.0f;
        {
            unsigned hi, lo;
            split2(rr[0] * hold[0], rr[1] * hold[1], hi, lo);
            lds_fence();
            rh_hi[wd] = hi;
            rh_lo[wd] = lo;
        }
        if constexpr (T9) {
            if (duty == 0) {
                const float rr9 = nok9 ? sigmoid4(fmaf(pick(acc9), inv9, cur.r[2])) : 0.0f;
                unsigned hi, lo;
                split2(rr9 * h9prev, 0.0f, hi, lo);
                rh_hi[wd9] = hi;
                rh_lo[wd9] = lo;
            } else if (duty == 1) {
                z9f[lane] = sigmoid4(fmaf(pick(acc9), inv9, cur.z[2]));
            }
        }
        half8 ch[KBS], cl[KBS];                          // MIX: ch = the mixed operands, cl unused
        if constexpr (MIX) {
            ch[0] = ldH(rh_img, moff[0]);                // my own block, straight back (LDS executes a wave's operations in order)
            cl[0] = ch[0];
        } else {
            ch[0] = ldH(rh_hi, boff[0]);
            cl[0] = ldH(rh_lo, boff[0]);
        }
        lds_fence();
        // ------------------------------ interval B ------------------------------
        if constexpr (MIX) lds_bar_1read(); else lds_bar_2reads();
#pragma unroll
        for (int i = 1; i < KBS; i++) {
            if constexpr (MIX) ch[i] = ldH(rh_img, moff[i]);
            else { ch[i] = ldH(rh_hi, boff[i]); cl[i] = ldH(rh_lo, boff[i]); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MIX) {
            z_block_mfma2<false>(accZ[0], accZ[1], wz_hi[0][KBS - 1], wz_lo[0][KBS - 1], wz_hi[1][KBS - 1], wz_lo[1][KBS - 1], bh[KBS - 1]);
            z_block_mfma2<true>(accC[0], accC[1], wc_hi[0][0], wc_lo[0][0], wc_hi[1][0], wc_lo[1][0], ch[0]);
            asm volatile("" ::"v"(bh[KBS - 1]));         // the fresh accumulators must not take over the operand the z MFMAs still read
        } else {
            z_block_mfma<false>(accZ[0], accZ[1], wz_hi[0][KBS - 1], wz_lo[0][KBS - 1], wz_hi[1][KBS - 1], wz_lo[1][KBS - 1], bh[KBS - 1],
                                bl[KBS - 1]);
            z_block_mfma<true>(accC[0], accC[1], wc_hi[0][0], wc_lo[0][0], wc_hi[1][0], wc_lo[1][0], ch[0], cl[0]);
            asm volatile("" ::"v"(bh[KBS - 1]), "v"(bl[KBS - 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 1; i < KBS; i++) { keep(ch[i]); if constexpr (!MIX) keep(cl[i]); }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MIX) z_block_mfma2<false>(accC[0], accC[1], wc_hi[0][1], wc_lo[0][1], wc_hi[1][1], wc_lo[1][1], ch[1]);
        else z_block_mfma<false>(accC[0], accC[1], wc_hi[0][1], wc_lo[0][1], wc_hi[1][1], wc_lo[1][1], ch[1], cl[1]);
        // the z accumulators: twelve (MIX: eight) MFMAs have been issued since their last one
        asm volatile("" : "+v"(accZ[0]), "+v"(accZ[1]));
        float zz[2], omz[2], zh[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            zz[p] = sigmoid4(fmaf(pick(accZ[p]), inv_z[p], cur.z[p]));
            omz[p] = 1.0f - zz[p];
            zh[p] = zz[p] * hold[p];
            asm volatile("" : "+v"(zh[p]), "+v"(omz[p]));                 // pinned here: not sunk to the blend below
        }
        static_for<2, KBS>([&](auto IC) {
            constexpr int i = decltype(IC)::value;
            if constexpr (MIX) z_block_mfma2<false>(accC[0], accC[1], wc_hi[0][i], wc_lo[0][i], wc_hi[1][i], wc_lo[1][i], ch[i]);
            else z_block_mfma<false>(accC[0], accC[1], wc_hi[0][i], wc_lo[0][i], wc_hi[1][i], wc_lo[1][i], ch[i], cl[i]);
        });
        f32x4 acc9c = {0.f, 0.f, 0.f, 0.f};
        float z9 = 0.0f;
        if constexpr (T9) {
            if (w3) {
                z9 = z9f[lane];                                                     // wave 1 wrote it before the barrier that opened this interval
                acc9c = tile9_mfma(2, ch, cl);
            }
        }
        mfma_drain(accC[0]);
        mfma_drain(accC[1]);
        float hn[2];
        {
            const float hb0 = tanh5(fmaf(pick(accC[0]), inv_c[0], cur.c[0])), hb1 = tanh5(fmaf(pick(accC[1]), inv_c[1], cur.c[1]));
            hn[0] = nok0 ? fmaf(omz[0], hb0, zh[0]) : 0.0f;               // layers.py:1020
            hn[1] = nok1 ? fmaf(omz[1], hb1, zh[1]) : 0.0f;
        }
        {
            unsigned hi, lo;
            split2(hn[0], hn[1], hi, lo);
            lds_fence();
            h_hi[wd] = hi;
            h_lo[wd] = lo;
        }
        float hn9 = 0.0f;
        if constexpr (T9) {
            if (w3) {
                const float hb9 = tanh5(fmaf(pick(acc9c), inv9, cur.c[2]));
                hn9 = nok9 ? fmaf(1.0f - z9, hb9, z9 * hold9) : 0.0f;
                h9f[lane] = hn9;
                unsigned hi, lo;
                split2(hn9, 0.0f, hi, lo);
                h_hi[wd9] = hi;
                h_lo[wd9] = lo;
            }
        }
        if constexpr (MIX) {
            oh = ldH(h_img, moff[0]);
            ol = oh;
        } else {
            oh = ldH(h_hi, boff[0]);
            ol = ldH(h_lo, boff[0]);
        }
        lds_fence();
        if (live && s < Tc && !(SCAN16_ABL & 2)) {
            if (nok0) hp[0] = hn[0];
            if (nok1) hp[16] = hn[1];
            if constexpr (T9) {
                if (w3 && nok9) hp[32] = hn9;                 // neuron 128 + 4g + q = n0 + 32 for wave 3
            }
        }
        hp += hstep;
        hold[0] = hn[0];
        hold[1] = hn[1];
        hold9 = hn9;
    };
    for (int s = 0; s < T; s += 4) {
        step(ic<0>{}, s);
        if (s + 1 < T) step(ic<1>{}, s + 1);
        if (s + 2 < T) step(ic<2>{}, s + 2);
        if (s + 3 < T) step(ic<3>{}, s + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of mine may land in registers after the wave has ended
}

// One workgroup per CU: ask for enough dynamic LDS that two cannot share a CU (each wave is compiled for a whole SIMD's registers).
template <int N>
static size_t scan16_exclusive_lds()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(gru_scan16_kernel<N>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(gru_scan16_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

extern "C" int slk_gru_scan1t_launch(const float *vI, long ldv, const float *sW, const float *sW2, float *y, long ldy, int T, int B, int n,
                                     int reverse, const int32_t *lens, hipStream_t s);             // gru_scan1t.hip

// include/sloika_amd.h
extern "C" int slk_gru_scan16_f32(const float *vI, long ldv, const float *sW, const float *sW2, float *y, long ldy, int T, int B, int n,
                                  int reverse, int act, int gate_act, const int32_t *lens, slk_stream_t stream)
{
    if (!vI || !sW || !sW2 || !y || T < 1 || B < 1 || n < 1 || ldv < 3L * n || ldy < n) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if (n % 16 || n <= 96 || n > 144) return SLK_ERR_UNSUPPORTED;
    if ((unsigned long long)T * B * ldv * sizeof(float) >= (1ull << 32)) return SLK_ERR_UNSUPPORTED;       // 32-bit lane offsets
    // one tile per wave (gru_scan1t.hip) is the faster plan at every width (a layer of 112 at B = 1024, T' = 800: 1.11-1.19 ms against
    // 1.21-1.28 with its projection, 128: 1.19-1.22 against 1.23-1.26; 144, nine waves with its last 16 neurons as a K block of 16:
    // scan alone 0.94 against 1.37).  SLOIKA_AMD_SCAN1T=0 / 1: never / always, for comparisons.
    static const int plan1t = getenv("SLOIKA_AMD_SCAN1T") ? atoi(getenv("SLOIKA_AMD_SCAN1T")) : -1;
    if (plan1t == 1 || plan1t < 0) {
        const int rc = slk_gru_scan1t_launch(vI, ldv, sW, sW2, y, ldy, T, B, n, reverse, lens, slk_stream(stream));
        if (rc != SLK_ERR_UNSUPPORTED) return rc;
    }
    if (n > 128) {
        const size_t dyn = SLK_PER_DEVICE(size_t, scan16_exclusive_lds<160>());
        hipLaunchKernelGGL((gru_scan16_kernel<160>), dim3((B + 3) / 4), dim3(256), dyn, slk_stream(stream), vI, ldv, sW, sW2, y, ldy, T, B,
                           n, reverse & 1, lens);
        return slk_launch_status();
    }
    const size_t dyn = SLK_PER_DEVICE(size_t, scan16_exclusive_lds<128>());
    hipLaunchKernelGGL((gru_scan16_kernel<128>), dim3((B + 3) / 4), dim3(256), dyn, slk_stream(stream), vI, ldv, sW, sW2, y, ldy, T, B, n,
                       reverse & 1, lens);
    return slk_launch_status();
}
