// gru_bar16q.hip -- the barrier-stepped Gru kernel with SIXTEEN chunks per workgroup (sloika/layers.py:1010-1021): every column of
// the recurrent 16x16x32 tiles is a different chunk, no copies left.  gru_bar16.hip gives the matrix pipe four chunks (four copies
// each), gru_bar16d.hip eight (two copies); here the same 60 MFMAs per chain wave and step serve sixteen, and a lane keeps all
// four rows of its tile: lane (g, n) owns neurons 4g .. 4g+3 of tiles 2w and 2w+1 for chunk n.  For batches of more than eight
// chunks per CU (four batches handed over as one call).
//
// What changes with it:
//   * the gate arithmetic is four values per lane and tile (eight independent chains per lane);
//   * the exchange is one 16-byte LDS write per image and step (four packed pairs), the state leaves as 16-byte stores;
//   * the time-parallel projection works on TWO steps x EIGHT chunks per 16-column tile (two sets of eight chunks), a group is
//     two steps = four intervals: the rings of vI and of the operand images are as large as gru_bar16d.hip's;
//   * the reset gate's AND the candidate's weights live in accumulation registers (asm MFMAs, bar16_common.h: both come in one run
//     in front of their epilogue anyway); the update gate's stay in ordinary registers, its MFMAs are the compiler's to interleave
//     with the reset gate's epilogue, all in interval A, and its own epilogue covers the LDS round trip of interval B.
// Arithmetic per (neuron, chunk) is that of the other two plans, instruction for instruction: results are bit-identical.
#include <limits.h>
#include <stdlib.h>

#include "bar16_common.h"

// timing experiments (tools/build_bar16q_variants.sh; results are garbage): 1 = service waves only keep the barriers, 2 = chain waves
// skip their share of the projection, 4 = no stores to h_out
#ifndef BAR16Q_ABL
#define BAR16Q_ABL 0
#endif
// first tile of interval k (of four) when a service wave has st tiles per group and set; the leader splits set 0 in interval 0 and
// set 1 in interval 2
__host__ __device__ constexpr int tile_first_q(int st, int k)
{
#ifndef BAR16Q_W
#define BAR16Q_W 2, 2, 2, 3
#endif
    constexpr int w[4] = {BAR16Q_W};
    int tot = 0, acc = 0;
    for (int i = 0; i < 4; i++) tot += w[i];
    for (int i = 0; i < k && i < 4; i++) acc += w[i];
    return k >= 4 ? st : (acc * st + tot / 2) / tot < st ? (acc * st + tot / 2) / tot : st;
}

template <int I, int N, bool SAVE>
__global__ void __launch_bounds__(256, 1) gru_bar16q_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ iW,
                                                            const float *__restrict__ bias, const float *__restrict__ sW,
                                                            const float *__restrict__ sW2, float *__restrict__ h_out, long ldh,
                                                            int T, int B, int reverse, const int *__restrict__ lens,
                                                            float *__restrict__ zr_out)
{
    static_assert(I % 16 == 0 && N % 32 == 0 && N <= 96, "unsupported size for the barrier-stepped GRU kernel");
    constexpr int NCW = N / 32;                          // chain waves = 32-wide K blocks of the recurrent products
    constexpr int KBS = N / 32;
    constexpr int NSW = 4 - NCW;                         // service waves
    constexpr int NT = N / 16;                           // tiles per gate
    constexpr int NT16 = 3 * NT;                         // tiles of vI rows (z | r | c)
    constexpr int KBLK = (I + 31) / 32;
    constexpr int GS = 2;                                // steps per projection group (16 MFMA columns = 2 steps x 8 chunks of a set)
    constexpr int R = 2 * GS;                            // vI ring: group G+1 is written while group G is consumed
#ifndef BAR16Q_CT
#define BAR16Q_CT 3
#endif
    constexpr int CT = NCW == 3 ? BAR16Q_CT : 0;         // projection tiles of a chain wave (weights in accumulation registers)
    constexpr int ST = (NT16 - NCW * CT) / NSW;          // ... of a service wave
    constexpr int NACAP = 240 / (8 * KBLK);              // 256 accumulation registers, 2 * KBLK * 4 per tile
    constexpr int NA = ST < NACAP ? ST : NACAP;
    static_assert(NCW * CT + NSW * ST == NT16, "tile assignment");
    static_assert(KBLK <= 4 && ST <= 21, "interval plan");
    constexpr int OPIMG = GS * KBLK * 128;               // dwords of one operand image: [step][k block][k group][chunk of 8][8 halves]
    constexpr int VSTEP = NT16 * 128;                    // floats of one step's vI of a set: [tile][g][chunk of 8][r]
    constexpr int IMG = KBS * 4 * 16 * 4;                // dwords of a state image: [k block][g][chunk of 16][4 packed pairs]

    __shared__ __attribute__((aligned(16))) unsigned xop_hi[2 * 2 * OPIMG], xop_lo[2 * 2 * OPIMG];      // [group & 1][set]
    __shared__ __attribute__((aligned(16))) float xinv_lds[2 * 2 * 16];
    __shared__ __attribute__((aligned(16))) float vbuf[R * 2 * VSTEP];                                   // [step % R][set]
    __shared__ __attribute__((aligned(16))) unsigned h_hi[IMG], h_lo[IMG], rh_hi[IMG], rh_lo[IMG];
    __shared__ __attribute__((aligned(16))) float bias_lds[3 * N], invw_lds[3 * N];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 16;

    for (int i = tid; i < IMG; i += 256) { h_hi[i] = 0u; h_lo[i] = 0u; }               // h(-1) = 0
    for (int i = tid; i < 3 * N; i += 256) bias_lds[i] = bias ? bias[i] : 0.0f;

    // ---------------- projection pieces shared by both kinds of wave ----------------
    const int pcol = lane & 15, kg = lane >> 4;          // operand row / column and k group of this lane
    const int pstep = pcol >> 3, pc = pcol & 7;          // as a B column: (step in group, chunk of the set)
    const int poff = pstep * (KBLK * 128) + kg * 32 + pc * 4;           // + 128 kb: my 16 bytes of an operand image, in dwords
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };
    auto opimg = [&](int grp, int set) { return ((grp & 1) * 2 + set) * OPIMG + poff; };
    // iW tile -> A operands (lane: row pcol of the tile, k = 32 kb + 8 kg + 0..7), row scale remembered in invw_lds
    auto load_tile = [&](int tile, half8 *hi, half8 *lo) {
        const int row = 16 * tile + pcol;
        float u[KBLK][8];
        float m = 0.0f;
#pragma unroll
        for (int kb = 0; kb < KBLK; kb++) {
            const int k0 = 32 * kb + 8 * kg;
            const bool kok = (I % 32 == 0) || k0 < I;
            const float *src = iW + (size_t)row * I + (kok ? k0 : 0);
            const float4 u0 = *reinterpret_cast<const float4 *>(src), u1 = *reinterpret_cast<const float4 *>(src + 4);
            const float t[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                u[kb][j] = kok ? t[j] : 0.0f;
                m = fmaxf(m, fabsf(u[kb][j]));
            }
        }
        float inv;
        const float ws = pow2_scale(kgroup_max(m), inv);
        if (kg == 0) invw_lds[row] = inv;
#pragma unroll
        for (int kb = 0; kb < KBLK; kb++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float v = u[kb][j] * ws;
                const _Float16 h = (_Float16)v;
                hi[kb][j] = h;
                lo[kb][j] = (_Float16)(v - (float)h);
            }
        }
    };
    // accumulator of a tile for group G1 of a set -> vI ring: lane holds rows 4 kg + r of column (pstep, pc)
    auto proj_out = [&](int tile, const f32x4 &acc, int G1, int set) {
        const float xin = xinv_lds[((G1 & 1) * 2 + set) * 16 + pcol];
        const f32x4 iw = *reinterpret_cast<const f32x4 *>(&invw_lds[16 * tile + 4 * kg]);
        const f32x4 bs = *reinterpret_cast<const f32x4 *>(&bias_lds[16 * tile + 4 * kg]);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = fmaf(acc[r] * xin, iw[r], bs[r]);
        const int st = GS * G1 + pstep;
        *reinterpret_cast<f32x4 *>(&vbuf[((st % R) * 2 + set) * VSTEP + ((tile * 4 + kg) * 8 + pc) * 4]) = o;
    };
    const int NG = (T + GS - 1) / GS;

    if (wave < NCW) {
        // =================================================================================================
        // chain waves
        // =================================================================================================
        const int w = wave;
        const int ck = lane & 15, g = lane >> 4;         // my chunk of the sixteen, my row group
        const int set = ck >> 3, pc8 = ck & 7;
        // recurrent weights: A operands, K blocks in the rotated order w, w+1, ... (element (g, j) of block kb is neuron
        // 32 kb + 16 (j&1) + 4 g + (j>>1), the order the owners' packed writes create), rows scaled to [1, 2)
        half8 wz_hi[2][KBS], wz_lo[2][KBS], wr_hi[2][KBS], wr_lo[2][KBS], wc_hi[2][KBS], wc_lo[2][KBS];
        float inv_z[2][4], inv_r[2][4], inv_c[2][4];
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int row = 32 * w + 16 * p + (lane & 15);
            float vz[KBS][8], vr[KBS][8], vc[KBS][8];
            float mz = 0.0f, mr = 0.0f, mc = 0.0f;
#pragma unroll
            for (int i = 0; i < KBS; i++) {
                const int kb = (w + i) % KBS;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                    vz[i][j] = sW[(size_t)row * N + k];
                    vr[i][j] = sW[(size_t)(N + row) * N + k];
                    vc[i][j] = sW2[(size_t)row * N + k];
                    mz = fmaxf(mz, fabsf(vz[i][j])); mr = fmaxf(mr, fabsf(vr[i][j])); mc = fmaxf(mc, fabsf(vc[i][j]));
                }
            }
            float iz, ir, ic_;
            const float sz = pow2_scale(kgroup_max(mz), iz), sr = pow2_scale(kgroup_max(mr), ir), sc = pow2_scale(kgroup_max(mc), ic_);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                inv_z[p][j] = __shfl(iz, 4 * g + j); inv_r[p][j] = __shfl(ir, 4 * g + j); inv_c[p][j] = __shfl(ic_, 4 * g + j);
            }
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
                wr_hi[p][i] = to_acc_regs(wr_hi[p][i]); wr_lo[p][i] = to_acc_regs(wr_lo[p][i]);
                wc_hi[p][i] = to_acc_regs(wc_hi[p][i]); wc_lo[p][i] = to_acc_regs(wc_lo[p][i]);
            }
        }
        constexpr int CTA = CT > 0 ? CT : 1;
        half8 pw_hi[CTA][KBLK], pw_lo[CTA][KBLK];
        f32x4 pacc[2][CTA];
        if constexpr (CT > 0) {
#pragma unroll
            for (int t = 0; t < CT; t++) {
                load_tile(w * CT + t, pw_hi[t], pw_lo[t]);
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) { pw_hi[t][kb] = to_acc_regs(pw_hi[t][kb]); pw_lo[t][kb] = to_acc_regs(pw_lo[t][kb]); }
            }
        }
        int boff[KBS];
#pragma unroll
        for (int i = 0; i < KBS; i++) boff[i] = ((((w + i) % KBS) * 4 + g) * 16 + ck) * 4;     // in dwords
        const int wd = ((w * 4 + g) * 16 + ck) * 4;                                             // my four packed pairs, in dwords
        const int n0 = 32 * w + 4 * g;                                                          // my neurons n0 .. n0+3 of tile 2w (+16: 2w+1)
        const int voff = (g * 8 + pc8) * 4;                                                     // my four elements of a vI tile
        // my chunk's rows of h_out (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
        const int bc = b0 + ck;
        const bool live = bc < B;
        const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
        const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
        float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + (live ? bc : 0)) * ldh + n0;
        const long zstep = (reverse ? -1L : 1L) * (long)B * 2 * N;
        float *zp = SAVE ? zr_out + ((size_t)(reverse ? Tc - 1 : 0) * B + (live ? bc : 0)) * (2 * N) + n0 : nullptr;

        __syncthreads();                                 // LDS initialised, every wave's invw_lds rows written
        lds_bar();                                       // x operand images of groups 0 and 1 (service leader)
        const half8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
        half8 pxh[2] = {hzero, hzero}, pxl[2] = {hzero, hzero};      // x operands of the coming slot of the projection share
        settle(pxh[0]); settle(pxh[1]); settle(pxl[0]); settle(pxl[1]);
        if constexpr (CT > 0) {                          // vI of group 0
#pragma unroll
            for (int sset = 0; sset < 2; sset++) {
                half8 xh0[KBLK], xl0[KBLK];
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) { xh0[kb] = ldH(xop_hi, opimg(0, sset) + 128 * kb); xl0[kb] = ldH(xop_lo, opimg(0, sset) + 128 * kb); }
#pragma unroll
                for (int t = 0; t < CT; t++) {
                    pacc[sset][t] = tile_mfma_acc<KBLK>(pw_hi[t], pw_lo[t], xh0, xl0);
                    mfma_drain(pacc[sset][t]);
                    proj_out(w * CT + t, pacc[sset][t], 0, sset);
                }
                pxh[sset] = ldH(xop_hi, opimg(1, sset));             // slot (step 0, interval A) projects K block 0 of group 1
                pxl[sset] = ldH(xop_lo, opimg(1, sset));
            }
        }
        lds_bar();                                       // vI of group 0 complete

        float hold[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        half8 oh = hzero, ol = hzero;                    // my own K block of h(s-1) as B operand (read back right after I wrote it)
        settle(oh);
        settle(ol);
        // The chain waves' share of the projection: a group is two steps = four slots (ph, interval), slot k = 2 ph + interval
        // projects K block k of the NEXT group (when k < KBLK); the operands of a slot are fetched one slot earlier
        auto proj_slot = [&](auto KC) {
            constexpr int k = decltype(KC)::value;
            if constexpr (CT > 0 && k < KBLK && !(BAR16Q_ABL & 2)) {
#pragma unroll
                for (int sset = 0; sset < 2; sset++) {
#pragma unroll
                    for (int t = 0; t < CT; t++) block_mfma_acc<k == 0>(pacc[sset][t], pw_hi[t][k], pw_lo[t][k], pxh[sset], pxl[sset]);
                }
            }
        };
        auto fetch_slot = [&](auto KC, int G, half8 *xh, half8 *xl) {       // operands of slot k of group G's successor
            constexpr int k = decltype(KC)::value;
            if constexpr (CT > 0 && k < KBLK) {
#pragma unroll
                for (int sset = 0; sset < 2; sset++) {
                    xh[sset] = ldH(xop_hi, opimg(G, sset) + 128 * k);
                    xl[sset] = ldH(xop_lo, opimg(G, sset) + 128 * k);
                }
            }
        };
        auto step = [&](auto PHC, const int s, const int G) {
            constexpr int ph = decltype(PHC)::value;
            // ------------------------------ interval A ------------------------------
            lds_bar_2reads();
            half8 bh[KBS], bl[KBS];
            bh[0] = oh;
            bl[0] = ol;
#pragma unroll
            for (int i = 1; i < KBS; i++) { bh[i] = ldH(h_hi, boff[i]); bl[i] = ldH(h_lo, boff[i]); }
            if (s > 0) {                                 // h(s-1), still in `hold` (gru_bar16.hip: stored behind the barrier, not in front of it)
                if (live && s - 1 < Tc && !(BAR16Q_ABL & 4)) {
                    *reinterpret_cast<f32x4 *>(hp) = f32x4{hold[0][0], hold[0][1], hold[0][2], hold[0][3]};
                    *reinterpret_cast<f32x4 *>(hp + 16) = f32x4{hold[1][0], hold[1][1], hold[1][2], hold[1][3]};
                }
                hp += hstep;
            }
            // vI(s): complete since the previous barrier at the latest
            const float *vcur = vbuf + ((s % R) * 2 + set) * VSTEP + voff;
            f32x4 vz[2], vr[2], vc[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                vr[p] = *reinterpret_cast<const f32x4 *>(vcur + 128 * (NT + 2 * w + p));
                vz[p] = *reinterpret_cast<const f32x4 *>(vcur + 128 * (2 * w + p));
                vc[p] = *reinterpret_cast<const f32x4 *>(vcur + 128 * (2 * NT + 2 * w + p));
            }
            half8 xhA[2], xlA[2];                        // operands of this step's interval-B slot (2 ph + 1)
            fetch_slot(ic<2 * ph + 1>{}, G + 1, xhA, xlA);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accR[2], accC[2], accZ[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            z_block_mfma<true>(accR[0], accR[1], wr_hi[0][0], wr_lo[0][0], wr_hi[1][0], wr_lo[1][0], bh[0], bl[0]);
            proj_slot(ic<2 * ph>{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 1; i < KBS; i++) { keep(bh[i]); keep(bl[i]); }
            if constexpr (CT > 0 && 2 * ph + 1 < KBLK) {
#pragma unroll
                for (int sset = 0; sset < 2; sset++) { keep(xhA[sset]); keep(xlA[sset]); pxh[sset] = xhA[sset]; pxl[sset] = xlA[sset]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // tile 0 first, then tile 1: tile 1's MFMAs are the wait states between tile 0's last MFMA and the arithmetic that reads its
            // sum (the hardware does not interlock that read: tools/probes/mfma_read_hazard_probe.hip), and they run while that
            // arithmetic does; tile 1's own sum is handed on behind tile 0's sigmoids (below)
            // (narrower layers -- two workgroups per CU, three MFMAs per tile -- keep the interleaved order and the drain: measured
            //  0.81 against 0.91 ms for 64 -> 64 at B = 4096)
            if constexpr (KBS >= 3) {
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    tile_block_mfma<false>(accR[0], wr_hi[0][i], wr_lo[0][i], bh[i], bl[i]);
                });
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    tile_block_mfma<false>(accR[1], wr_hi[1][i], wr_lo[1][i], bh[i], bl[i]);
                });
                // tile 0's sum is handed on BEHIND tile 1's MFMAs: the arithmetic that reads it must not be placed in front of them
                asm volatile("" : "+v"(accR[0]), "+v"(accR[1]));
            } else {
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    z_block_mfma<false>(accR[0], accR[1], wr_hi[0][i], wr_lo[0][i], wr_hi[1][i], wr_lo[1][i], bh[i], bl[i]);
                });
                mfma_drain2(accR[0], accR[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ALL z products under the r epilogue (they only need h(s-1), like r)
#pragma unroll
            for (int i = 0; i < KBS; i++)
                mfma3x2(wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh[i], bl[i], accZ[0], accZ[1]);
            float rr[2][4];
#pragma unroll
            for (int j = 0; j < 4; j++) rr[0][j] = sigmoid4(fmaf(accR[0][j], inv_r[0][j], vr[0][j]));
            // (twenty instructions of tile 0's gate lie in front of the first read of tile 1's sum)
            asm volatile("" : "+v"(accR[1]) : "v"(rr[0][0]), "v"(rr[0][1]), "v"(rr[0][2]), "v"(rr[0][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) rr[1][j] = sigmoid4(fmaf(accR[1][j], inv_r[1][j], vr[1][j]));
            {
                uint4 hi, lo;
                split2(rr[0][0] * hold[0][0], rr[1][0] * hold[1][0], hi.x, lo.x);
                split2(rr[0][1] * hold[0][1], rr[1][1] * hold[1][1], hi.y, lo.y);
                split2(rr[0][2] * hold[0][2], rr[1][2] * hold[1][2], hi.z, lo.z);
                split2(rr[0][3] * hold[0][3], rr[1][3] * hold[1][3], hi.w, lo.w);
                lds_fence();
                *reinterpret_cast<uint4 *>(&rh_hi[wd]) = hi;
                *reinterpret_cast<uint4 *>(&rh_lo[wd]) = lo;
            }
            half8 ch[KBS], cl[KBS];
            ch[0] = ldH(rh_hi, boff[0]);                 // my own block, straight back (LDS executes a wave's operations in order)
            cl[0] = ldH(rh_lo, boff[0]);
            lds_fence();
            // one MFMA, then up to four VALU instructions, for as long as both last
#ifndef BAR16Q_ZV
#define BAR16Q_ZV 4
#endif
#pragma unroll
            for (int i = 0; i < 6 * KBS; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, BAR16Q_ZV, 0);
            }
            const bool store = live && s < Tc && !(BAR16Q_ABL & 4);
            if constexpr (SAVE) {
                if (store) {
                    *reinterpret_cast<f32x4 *>(zp + N) = f32x4{rr[0][0], rr[0][1], rr[0][2], rr[0][3]};
                    *reinterpret_cast<f32x4 *>(zp + N + 16) = f32x4{rr[1][0], rr[1][1], rr[1][2], rr[1][3]};
                }
            }
            // ------------------------------ interval B ------------------------------
            lds_bar_2reads();
#pragma unroll
            for (int i = 1; i < KBS; i++) { ch[i] = ldH(rh_hi, boff[i]); cl[i] = ldH(rh_lo, boff[i]); }
            half8 xhB[2], xlB[2];                        // operands of the next step's interval-A slot: K block 2 of this group's
            constexpr int nk = ph == 0 ? 2 : 0;          // successor, or K block 0 of the one after
            fetch_slot(ic<nk>{}, G + (ph == 0 ? 1 : 2), xhB, xlB);
            __builtin_amdgcn_sched_barrier(0);
            z_block_mfma<true>(accC[0], accC[1], wc_hi[0][0], wc_lo[0][0], wc_hi[1][0], wc_lo[1][0], ch[0], cl[0]);
            // (the fresh accumulators must not take over the operands the update gate's MFMAs just issued still read)
#pragma unroll
            for (int i = 0; i < KBS; i++) asm volatile("" ::"v"(bh[i]), "v"(bl[i]));
            proj_slot(ic<2 * ph + 1>{});
            // the update gate's epilogue while the other waves' r*h is on its way
            float zz[2][4];
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int j = 0; j < 4; j++) zz[p][j] = sigmoid4(fmaf(accZ[p][j], inv_z[p][j], vz[p][j]));
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 1; i < KBS; i++) { keep(ch[i]); keep(cl[i]); }
            if constexpr (CT > 0 && nk < KBLK) {
#pragma unroll
                for (int sset = 0; sset < 2; sset++) { keep(xhB[sset]); keep(xlB[sset]); pxh[sset] = xhB[sset]; pxl[sset] = xlB[sset]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KBS >= 3) {
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    tile_block_mfma<false>(accC[0], wc_hi[0][i], wc_lo[0][i], ch[i], cl[i]);
                });
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    tile_block_mfma<false>(accC[1], wc_hi[1][i], wc_lo[1][i], ch[i], cl[i]);
                });
            } else {
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    z_block_mfma<false>(accC[0], accC[1], wc_hi[0][i], wc_lo[0][i], wc_hi[1][i], wc_lo[1][i], ch[i], cl[i]);
                });
            }
            if constexpr (CT > 0 && ph == 1) {           // vI of group G+1: its last MFMAs are at least a candidate block old, or drained
#pragma unroll
                for (int sset = 0; sset < 2; sset++) {
#pragma unroll
                    for (int t = 0; t < CT; t++) {
                        if constexpr (KBLK == 4 || KBS == 1) mfma_drain(pacc[sset][t]);
                        else asm volatile("" : "+v"(pacc[sset][t]));
                        proj_out(w * CT + t, pacc[sset][t], G + 1, sset);
                    }
                }
            }
            if constexpr (KBS >= 3) asm volatile("" : "+v"(accC[0]), "+v"(accC[1]));
            else mfma_drain2(accC[0], accC[1]);
            __builtin_amdgcn_sched_barrier(0);
            float hn[2][4];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                // tile 1's sum behind tile 0's candidate (see interval A)
                if (p == 1) asm volatile("" : "+v"(accC[1]) : "v"(hn[0][0]), "v"(hn[0][1]), "v"(hn[0][2]), "v"(hn[0][3]));
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float hbar = tanh5(fmaf(accC[p][j], inv_c[p][j], vc[p][j]));
                    const float omz = 1.0f - zz[p][j], zh = zz[p][j] * hold[p][j];
                    hn[p][j] = fmaf(omz, hbar, zh);                           // layers.py:1020
                }
            }
            {
                uint4 hi, lo;
                split2(hn[0][0], hn[1][0], hi.x, lo.x);
                split2(hn[0][1], hn[1][1], hi.y, lo.y);
                split2(hn[0][2], hn[1][2], hi.z, lo.z);
                split2(hn[0][3], hn[1][3], hi.w, lo.w);
                lds_fence();
                *reinterpret_cast<uint4 *>(&h_hi[wd]) = hi;
                *reinterpret_cast<uint4 *>(&h_lo[wd]) = lo;
            }
            oh = ldH(h_hi, boff[0]);
            ol = ldH(h_lo, boff[0]);
            lds_fence();
            if constexpr (SAVE) {
                if (store) {
                    *reinterpret_cast<f32x4 *>(zp) = f32x4{zz[0][0], zz[0][1], zz[0][2], zz[0][3]};
                    *reinterpret_cast<f32x4 *>(zp + 16) = f32x4{zz[1][0], zz[1][1], zz[1][2], zz[1][3]};
                }
                zp += zstep;
            }
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int j = 0; j < 4; j++) hold[p][j] = hn[p][j];
            }
        };
        for (int G = 0; G < NG; G++) {
            const int s = GS * G;
            step(ic<0>{}, s, G);
            if (s + 1 < T) step(ic<1>{}, s + 1, G);
        }
        if (live && T - 1 < Tc && !(BAR16Q_ABL & 4)) {   // h of the last step
            *reinterpret_cast<f32x4 *>(hp) = f32x4{hold[0][0], hold[0][1], hold[0][2], hold[0][3]};
            *reinterpret_cast<f32x4 *>(hp + 16) = f32x4{hold[1][0], hold[1][1], hold[1][2], hold[1][3]};
        }
    } else {
        // =================================================================================================
        // service waves: the rest of the projection; the leader (first of them) also loads and splits x
        // =================================================================================================
        const int sw = wave - NCW;
        const bool leader = sw == 0;
        const int tile0 = NCW * CT + sw * ST;
        constexpr int NV = ST - NA > 0 ? ST - NA : 1;
        half8 pa_hi[NA][KBLK], pa_lo[NA][KBLK];          // tiles 0..NA-1: accumulation registers
        half8 pw_hi[NV][KBLK], pw_lo[NV][KBLK];          // the rest: ordinary registers
#pragma unroll
        for (int t = 0; t < NA; t++) {
            load_tile(tile0 + t, pa_hi[t], pa_lo[t]);
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) { pa_hi[t][kb] = to_acc_regs(pa_hi[t][kb]); pa_lo[t][kb] = to_acc_regs(pa_lo[t][kb]); }
        }
#pragma unroll
        for (int t = NA; t < ST; t++) load_tile(tile0 + t, pw_hi[t - NA], pw_lo[t - NA]);

        // x of a group and set: the leader's lane (row pcol = (step, chunk), k group kg) loads ITS eight floats of every K block
        // straight into registers a group ahead of the split (ordinary loads: the compiler waits where the split first uses them)
        int xbc[2], xTc[2];
#pragma unroll
        for (int sset = 0; sset < 2; sset++) {
            xbc[sset] = min(b0 + 8 * sset + pc, B - 1);
            xTc[sset] = lens ? min(max(lens[xbc[sset]], 1), T) : T;
        }
        f32x4 xr[2][KBLK][2];
        auto load_x = [&](int G2, auto SC) {
            constexpr int sset = decltype(SC)::value;
            // steps past the chunk's end re-read its last valid row (their results are never stored)
            const int ss = min(G2 * GS + pstep, xTc[sset] - 1);
            const int tt = reverse ? xTc[sset] - 1 - ss : ss;
            const float *row = x + ((size_t)tt * B + xbc[sset]) * ldx;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kg;
                const bool kok = (I % 32 == 0) || k0 < I;
                const float *src = row + (kok ? k0 : 0);
                xr[sset][kb][0] = *reinterpret_cast<const f32x4 *>(src);
                xr[sset][kb][1] = *reinterpret_cast<const f32x4 *>(src + 4);
            }
        };
        // the split of a set's rows is spread over two intervals (the scale and the first K block, then the others): done in one, it
        // made that interval the longest of the group and the chain waves waited for it at the barrier
        float raw[KBLK][8], xs = 1.0f;
        auto split_scale = [&](int G2, auto SC) {        // the row's power-of-two scale
            constexpr int sset = decltype(SC)::value;
            float amax = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kg;
                const bool kok = (I % 32 == 0) || k0 < I;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    raw[kb][j] = kok ? xr[sset][kb][0][j] : 0.0f;
                    raw[kb][4 + j] = kok ? xr[sset][kb][1][j] : 0.0f;
                    amax = fmaxf(amax, fmaxf(fabsf(raw[kb][j]), fabsf(raw[kb][4 + j])));
                }
            }
            float xinv;
            xs = pow2_scale(kgroup_max(amax), xinv);
            if (kg == 0) xinv_lds[((G2 & 1) * 2 + sset) * 16 + pcol] = xinv;
        };
        auto split_block = [&](int G2, int sset, int kb) {         // K block kb -> operand images
            unsigned ahi[4], alo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) split2(raw[kb][2 * j] * xs, raw[kb][2 * j + 1] * xs, ahi[j], alo[j]);
            const int ob = opimg(G2, sset) + 128 * kb;
            *reinterpret_cast<uint4 *>(xop_hi + ob) = make_uint4(ahi[0], ahi[1], ahi[2], ahi[3]);
            *reinterpret_cast<uint4 *>(xop_lo + ob) = make_uint4(alo[0], alo[1], alo[2], alo[3]);
        };
        auto split_set = [&](int G2, auto SC) {
            split_scale(G2, SC);
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) split_block(G2, decltype(SC)::value, kb);
        };

        __syncthreads();
        if (leader) {
            static_for<0, 2>([&](auto GC) {
                static_for<0, 2>([&](auto SC) {
                    load_x(decltype(GC)::value, SC);
                    split_set(decltype(GC)::value, SC);
                });
            });
            load_x(2, ic<0>{});                          // group 2: split during group 0
            load_x(2, ic<1>{});
        }
        lds_bar();
        half8 xh[2][KBLK], xl[2][KBLK];
        auto load_operands = [&](int G1) {
#pragma unroll
            for (int sset = 0; sset < 2; sset++) {
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) {
                    const int ob = opimg(G1, sset) + 128 * kb;
                    xh[sset][kb] = ldH(xop_hi, ob);
                    xl[sset][kb] = ldH(xop_lo, ob);
                }
            }
        };
        // one tile for both sets: the output constants are requested first, then all MFMAs, one drain, then the outputs
        auto no_hook = [](auto) {};
        auto project_tile = [&](auto T0, int G1) {
            constexpr int t0 = decltype(T0)::value;
            const float xin0 = xinv_lds[((G1 & 1) * 2 + 0) * 16 + pcol], xin1 = xinv_lds[((G1 & 1) * 2 + 1) * 16 + pcol];
            const f32x4 iw0 = *reinterpret_cast<const f32x4 *>(&invw_lds[16 * (tile0 + t0) + 4 * kg]);
            const f32x4 bs0 = *reinterpret_cast<const f32x4 *>(&bias_lds[16 * (tile0 + t0) + 4 * kg]);
            f32x4 a0, a1;
            if constexpr (t0 < NA) {
                tile2_mfma_acc<KBLK>(a0, a1, pa_hi[t0], pa_lo[t0], xh[0], xl[0], xh[1], xl[1], no_hook);
            } else {                                     // weights in ordinary registers: builtins, scheduled by the compiler
                a0 = f32x4{0.f, 0.f, 0.f, 0.f};
                a1 = a0;
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) {
                    a0 = mfma3(pw_hi[t0 - NA][kb], pw_lo[t0 - NA][kb], xh[0][kb], xl[0][kb], a0);
                    a1 = mfma3(pw_hi[t0 - NA][kb], pw_lo[t0 - NA][kb], xh[1][kb], xl[1][kb], a1);
                }
            }
            mfma_drain2(a0, a1);
            const int st = GS * G1 + pstep;
            float *dst = &vbuf[(st % R) * 2 * VSTEP + 128 * (tile0 + t0) + (kg * 8 + pc) * 4];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = fmaf(a0[r] * xin0, iw0[r], bs0[r]);
            *reinterpret_cast<f32x4 *>(dst) = o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = fmaf(a1[r] * xin1, iw0[r], bs0[r]);
            *reinterpret_cast<f32x4 *>(dst + VSTEP) = o;
        };
        // interval k of group G: the tiles of group G1 = G + 1; the leader splits set 0 of group G + 2 in intervals 0 and 1, set 1 in
        // intervals 2 and 3, and requests the same set of group G + 3 right behind
        auto project_interval = [&](auto KC, int G1) {
            constexpr int k = decltype(KC)::value;
            constexpr int lo = tile_first_q(ST, k), hi = tile_first_q(ST, k + 1);
            static_for<lo, hi>([&](auto TC) { project_tile(TC, G1); });
            if (leader && G1 > 0) {
                constexpr int sset = k / 2;
                if constexpr (NCW < 3) {                 // narrow layers (two workgroups per CU): the whole set in one interval, as measured
                    if constexpr ((k & 1) == 0) {
                        split_set(G1 + 1, ic<sset>{});
                        load_x(G1 + 2, ic<sset>{});
                    }
                } else if constexpr ((k & 1) == 0) {
                    split_scale(G1 + 1, ic<sset>{});
                    split_block(G1 + 1, sset, 0);
                } else {
#pragma unroll
                    for (int kb = 1; kb < KBLK; kb++) split_block(G1 + 1, sset, kb);
                    load_x(G1 + 2, ic<sset>{});
                }
            }
        };
        load_operands(0);
        static_for<0, 4>([&](auto KC) { project_interval(KC, 0); });
        lds_bar();                                       // vI of group 0 complete

        auto interval = [&](auto KC, const int G) {
            constexpr int k = decltype(KC)::value;
            lds_bar();
            if constexpr (BAR16Q_ABL & 1) return;
            if constexpr (k == 0) load_operands(G + 1);
            project_interval(KC, G + 1);
        };
        for (int G = 0; G < NG; G++) {
            const int s = GS * G;
            interval(ic<0>{}, G); interval(ic<1>{}, G);
            if (s + 1 < T) { interval(ic<2>{}, G); interval(ic<3>{}, G); }
        }
    }
}

template <int I, int N>
static int launch_bar16q(const float *x, long ldx, const float *iW, const float *bias, const float *sW, const float *sW2,
                         float *y, long ldy, int T, int B, int reverse, const int *lens, float *zr_out, hipStream_t s)
{
    if (zr_out)
        hipLaunchKernelGGL((gru_bar16q_kernel<I, N, true>), dim3((B + 15) / 16), dim3(256), 0, s, x, ldx, iW, bias, sW, sW2, y, ldy, T,
                           B, reverse & 1, lens, zr_out);
    else
        hipLaunchKernelGGL((gru_bar16q_kernel<I, N, false>), dim3((B + 15) / 16), dim3(256), 0, s, x, ldx, iW, bias, sW, sW2, y, ldy, T,
                           B, reverse & 1, lens, zr_out);
    return slk_launch_status();
}

// The sixteen-chunk plan behind slk_gru_bar16_f32 (same contract); SLK_ERR_UNSUPPORTED when no instantiation covers the request
// (the caller then takes the eight- or four-chunk kernel).
extern "C" int slk_gru_bar16q_launch(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias,
                                     float *y, long ldy, int T, int B, int insize, int n, int reverse, const int32_t *lens,
                                     float *zr_out, hipStream_t s)
{
    if ((ldy & 3) || (reinterpret_cast<uintptr_t>(y) & 15)) return SLK_ERR_UNSUPPORTED;         // 16-byte state stores
#define BAR16Q(II, NN) \
    if (insize == II && n == NN) return launch_bar16q<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse, lens, zr_out, s);
    BAR16Q(96, 96) BAR16Q(64, 64) BAR16Q(32, 96) BAR16Q(128, 96) BAR16Q(64, 96) BAR16Q(48, 32) BAR16Q(16, 64)
#undef BAR16Q
    return SLK_ERR_UNSUPPORTED;
}
