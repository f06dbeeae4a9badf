// gru_bar16d.hip -- the barrier-stepped Gru kernel of gru_bar16.hip with EIGHT chunks per workgroup (sloika/layers.py:1010-1021).
//
// gru_bar16.hip gives the matrix pipe four chunks and fills the other twelve columns of a 16-column tile with three more
// copies of them, so that every lane ends up owning one (neuron, chunk) pair.  A batch of more than 1024 chunks (two
// batches in flight, long reads cut into many chunks) needs more than one workgroup per CU that way, one after the other.
// Here a workgroup takes two sets of four chunks: column groups 0 and 1 carry set 0, groups 2 and 3 set 1.  The recurrent
// MFMAs are the SAME instructions as for four chunks -- 60 per chain wave and step, now with two copies of each chunk instead
// of four -- and a lane owns two neighbouring neurons of its tile for one chunk: the gate arithmetic doubles (two independent
// values per lane, which also fills the issue gaps of the dependent chain), the matrix work per chunk halves.
//   lane (g, q, c): set q>>1, chunk c of the set, rows 4g + 2(q&1) + {0, 1} of tiles 2w and 2w+1
// The time-parallel projection has no copies to give up: it runs once per set (4 steps x 4 chunks = 16 columns each), so the
// service waves and the chain waves' share of it do twice the work per group; x arrives in blocks of one group (4 steps) per
// set so that both sets' rings fit into LDS.
// Everything else -- operand images, range scaling, the two barriers per step, own-block read-back, states stored straight
// to h_out -- is gru_bar16.hip's; see there and DESIGN.md section 3.1.
#include <limits.h>
#include <stdlib.h>

#include "bar16_common.h"

typedef float f32x2d __attribute__((ext_vector_type(2)));

// timing experiments (tools/build_bar16d_variants.sh; results are garbage): 1 = service waves only keep the barriers, 2 = the leader
// does not split x, 4 = chain waves skip their share of the projection, 8 = no stores to h_out
//   16 = stamps: cycles each wave of workgroup 0 works (barrier exit -> next barrier entry) and waits per interval of a group
#ifndef BAR16D_ABL
#define BAR16D_ABL 0
#endif
// 1: the leader's split of x is cut into pieces issued between the MFMAs of a tile (measured: slower, 3320 vs 3010 cycles per step)
#ifndef BAR16D_HOOKS
#define BAR16D_HOOKS 0
#endif
#ifndef BAR16D_CT
#define BAR16D_CT 2
#endif
// 1: the reset gate's epilogue in pieces between the update gate's asm MFMAs (interval A); measured: 2763 against 2697 cycles per step
#ifndef BAR16D_ZHOOK
#define BAR16D_ZHOOK 0
#endif
// which gate's weights live in accumulation registers (its MFMAs are asm, placed where the source puts them): 0 = the update gate,
// 1 = the candidate (its MFMAs come in one run in front of its epilogue anyway; the update gate's stay the compiler's to interleave):
// measured 2912 against 2782 cycles per step
// 1: two MFMAs per recurrent product, the two copies of a chunk carrying the hi and the lo half of the state (bar16_common.h: mfma2x2,
// pick_mix_d) -- the same arithmetic as gru_bar16.hip's, bit for bit; 0: round 2's three-term sequence
#ifndef BAR16D_MIX
#define BAR16D_MIX 1
#endif
#ifndef BAR16D_CACC
#define BAR16D_CACC 0
#endif
__device__ unsigned long long slk_dbg_bar16d[4][16];
#ifdef SLK_DIAG                          /* tools/build_diag_lib.sh */
extern "C" SLK_API int slk_debug_read_bar16d(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_bar16d), sizeof(unsigned long long) * 64) == hipSuccess ? SLK_OK
                                                                                                                 : SLK_ERR_LAUNCH;
}
#endif
// DSTAMP_IN(k): about to enter the barrier that opens interval k; DSTAMP_OUT(k): through it
#define DSTAMP_IN(k)                                                                  \
    if constexpr (BAR16D_ABL & 16) {                                                  \
        unsigned long long tnow;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        dwork[((k) + 7) & 7] += tnow - tprev;                                         \
        tprev = tnow;                                                                 \
    }
#define DSTAMP_OUT(k)                                                                 \
    if constexpr (BAR16D_ABL & 16) {                                                  \
        unsigned long long tnow;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        dwait[(k) & 7] += tnow - tprev;                                               \
        tprev = tnow;                                                                 \
    }

// first tile of interval k when a service wave has st tiles per group and set (k = 8: st); the leader splits x of set 0 in
// intervals 1..KBLK and of set 1 in intervals 4.. (or 5..), so the load is even except for interval 0 (operand fetch)
__host__ __device__ constexpr int tile_first_d(int st, int k)
{
    constexpr int w[8] = {1, 2, 2, 2, 2, 2, 2, 3};
    int tot = 0, acc = 0;
    for (int i = 0; i < 8; i++) tot += w[i];
    for (int i = 0; i < k && i < 8; i++) acc += w[i];
    return k >= 8 ? st : (acc * st + tot / 2) / tot < st ? (acc * st + tot / 2) / tot : st;
}

template <int I, int N, bool SAVE>
__global__ void __launch_bounds__(256, 1) gru_bar16d_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ iW,
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
    constexpr int GS = 4;                                // steps per projection group (16 MFMA columns = 4 steps x 4 chunks of a set)
    constexpr int R = 2 * GS;                            // vI ring: group G+1 is written while group G is consumed
    constexpr int CT = NCW == 3 ? (KBLK == 4 ? 3 : BAR16D_CT) : 0;   // projection tiles of a chain wave (KBLK = 4: the service wave's registers hold 9 tiles, not 12) (weights in accumulation registers)
    constexpr int ST = (NT16 - NCW * CT) / NSW;          // ... of a service wave
    constexpr int NACAP = 240 / (8 * KBLK);              // 256 accumulation registers, 2 * KBLK * 4 per tile
    constexpr int NA = ST < NACAP ? ST : NACAP;
    static_assert(NCW * CT + NSW * ST == NT16, "tile assignment");
    static_assert(KBLK <= 4 && ST <= 21, "interval plan");
    // dwords of one operand image: [step][k block][k group][chunk][8 halves]; the steps 32 banks apart (gru_bar16.hip: the 16-lane groups
    // of a ds_read_b128 -- lanes of two k groups and two steps each -- then find their pieces on different banks)
    constexpr int OPSTEP = KBLK * 64 + 32;
    constexpr int OPIMG = GS * OPSTEP;
    constexpr int VSTEP = NT16 * 64;                     // floats of one step's vI of a set: [tile][g][chunk][r]

    __shared__ __attribute__((aligned(16))) unsigned xop_hi[2 * 2 * OPIMG], xop_lo[2 * 2 * OPIMG];      // [group & 1][set]
    __shared__ __attribute__((aligned(16))) float xinv_lds[2 * 2 * 16];
    __shared__ __attribute__((aligned(16))) float vbuf[R * 2 * VSTEP];                                   // [step % R][set]
    constexpr bool MIX = BAR16D_MIX != 0;
    static_assert(!MIX || (!BAR16D_ZHOOK && !BAR16D_CACC), "the two-term products are written for the default schedule only");
    // hi image [set], then lo image [set], the lo image 32 banks behind the hi image (gru_bar16.hip: on the same banks a ds_read_b128 of
    // the mixed operand, whose lane quartets read both, takes two passes -- LDSBankConflict 7.8 % of this kernel's LDS cycles in round 4)
    constexpr int IMG = 4 * N + (4 * N % 64 == 0 ? 32 : 4 * N % 64 == 32 ? 0 : 4);
    __shared__ __attribute__((aligned(16))) unsigned h_img[2 * IMG], rh_img[2 * IMG];
    unsigned *const h_hi = h_img, *const h_lo = h_img + IMG, *const rh_hi = rh_img, *const rh_lo = rh_img + IMG;
    __shared__ __attribute__((aligned(16))) float bias_lds[3 * N], invw_lds[3 * N];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 8;

    for (int i = tid; i < 4 * N; i += 256) { h_hi[i] = 0u; h_lo[i] = 0u; }             // h(-1) = 0
    for (int i = tid; i < 3 * N; i += 256) bias_lds[i] = bias ? bias[i] : 0.0f;

    // ---------------- projection pieces shared by both kinds of wave ----------------
    const int pcol = lane & 15, kg = lane >> 4;          // operand row / column and k group of this lane
    const int pstep = pcol >> 2, pc = pcol & 3;          // as a B column: (step in group, chunk of the set)
    const int poff = pstep * OPSTEP + kg * 16 + pc * 4;                 // + 64 kb: my 16 bytes of an operand image, in dwords
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
        *reinterpret_cast<f32x4 *>(&vbuf[((st % R) * 2 + set) * VSTEP + ((tile * 4 + kg) * 4 + pc) * 4]) = o;
    };
    const int NG = (T + GS - 1) / GS;

    if (wave < NCW) {
        // =================================================================================================
        // chain waves
        // =================================================================================================
        const int w = wave;
        const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
        const int set = q >> 1, qh = q & 1;
        // recurrent weights: A operands, K blocks in the rotated order w, w+1, ... (element (g, j) of block kb is neuron
        // 32 kb + 16 (j&1) + 4 g + (j>>1), the order the owners' packed writes create), rows scaled to [1, 2)
        half8 wz_hi[2][KBS], wz_lo[2][KBS], wr_hi[2][KBS], wr_lo[2][KBS], wc_hi[2][KBS], wc_lo[2][KBS];
        float inv_z[2][2], inv_r[2][2], inv_c[2][2];
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
            for (int j = 0; j < 2; j++) {
                inv_z[p][j] = __shfl(iz, 4 * g + 2 * qh + j); inv_r[p][j] = __shfl(ir, 4 * g + 2 * qh + j);
                inv_c[p][j] = __shfl(ic_, 4 * g + 2 * qh + j);
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
        // the update gate's weights live in accumulation registers (its MFMAs are asm, off the serial path): the ordinary
        // registers they free hold the second value of every gate
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
            for (int i = 0; i < KBS; i++) {
                if constexpr (BAR16D_CACC) { wc_hi[p][i] = to_acc_regs(wc_hi[p][i]); wc_lo[p][i] = to_acc_regs(wc_lo[p][i]); }
                else { wz_hi[p][i] = to_acc_regs(wz_hi[p][i]); wz_lo[p][i] = to_acc_regs(wz_lo[p][i]); }
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
        for (int i = 0; i < KBS; i++) boff[i] = set * 2 * N + ((((w + i) % KBS) * 4 + g) * 4 + c) * 4;        // in dwords
        int moff[KBS];                                   // MIX: my copy's image (q & 1 = 0: hi, 1: lo)
#pragma unroll
        for (int i = 0; i < KBS; i++) moff[i] = qh * IMG + boff[i];
        const int wd = set * 2 * N + ((w * 4 + g) * 4 + c) * 4 + 2 * qh;                         // my two packed pairs, in dwords
        const int n0 = 32 * w + 4 * g + 2 * qh;                                                  // my neurons n0, n0+1 of tile 2w (+16: 2w+1)
        const int voff = (g * 4 + c) * 4 + 2 * qh;                                               // my two elements of a vI tile
        // my chunk's rows of h_out (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
        const int bc = b0 + 4 * set + c;
        const bool live = bc < B;
        const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
        const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
        float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + (live ? bc : 0)) * ldh + n0;
        const long zstep = (reverse ? -1L : 1L) * (long)B * 2 * N;
        float *zp = SAVE ? zr_out + ((size_t)(reverse ? Tc - 1 : 0) * B + (live ? bc : 0)) * (2 * N) + n0 : nullptr;

        __syncthreads();                                 // LDS initialised, every wave's invw_lds rows written
        lds_bar();                                       // x operand images of groups 0 and 1 (service leader)
        const half8 hzero = {0, 0, 0, 0, 0, 0, 0, 0};
        half8 pxh[2] = {hzero, hzero}, pxl[2] = {hzero, hzero};      // x operands of the coming step's share of the projection
        settle(pxh[0]); settle(pxh[1]); settle(pxl[0]); settle(pxl[1]);
        if constexpr (CT > 0) {                          // vI of group 0
#pragma unroll
            for (int sset = 0; sset < 2; sset++) {
                half8 xh0[KBLK], xl0[KBLK];
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) { xh0[kb] = ldH(xop_hi, opimg(0, sset) + 64 * kb); xl0[kb] = ldH(xop_lo, opimg(0, sset) + 64 * kb); }
#pragma unroll
                for (int t = 0; t < CT; t++) {
                    pacc[sset][t] = tile_mfma_acc<KBLK>(pw_hi[t], pw_lo[t], xh0, xl0);
                    mfma_drain(pacc[sset][t]);
                    proj_out(w * CT + t, pacc[sset][t], 0, sset);
                }
                pxh[sset] = ldH(xop_hi, opimg(1, sset));             // step 0 projects K block 0 of group 1
                pxl[sset] = ldH(xop_lo, opimg(1, sset));
            }
        }
        lds_bar();                                       // vI of group 0 complete

        unsigned long long dwork[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dwait[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        if constexpr (BAR16D_ABL & 16) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
        float hold[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
        [[maybe_unused]] float zkeep[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};      // SAVE: the update gates of the step before
        // carried from step to step: my own K block of h(s-1) as B operand (read back right after I wrote it)
        half8 oh = hzero, ol = hzero;
        settle(oh);
        settle(ol);
        // rows 4g + 2qh + j of a tile's accumulator
        auto pick = [&](const f32x4 &a, int j) {
            if constexpr (MIX) return pick_mix_d(a, j);
            else return qh ? a[2 + j] : a[j];
        };
        auto mfma_z = [&](auto FIRSTC, int i, const half8 &bh_, const half8 &bl_, f32x4 &a0, f32x4 &a1) {
            if constexpr (MIX) z_block_mfma2<decltype(FIRSTC)::value != 0>(a0, a1, wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh_);
            else z_block_mfma<decltype(FIRSTC)::value != 0>(a0, a1, wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh_, bl_);
        };
        // r / c products of one K block (MIX: bh_ = the mixed operand, bl_ unused)
        auto mfma_rc = [&](const half8 (&wh)[2][KBS], const half8 (&wl)[2][KBS], int i, const half8 &bh_, const half8 &bl_, f32x4 &a0, f32x4 &a1) {
            if constexpr (MIX) mfma2x2(wh[0][i], wl[0][i], wh[1][i], wl[1][i], bh_, a0, a1);
            else mfma3x2(wh[0][i], wl[0][i], wh[1][i], wl[1][i], bh_, bl_, a0, a1);
        };
        // One step = two intervals, each opened by a barrier (see gru_bar16.hip for the plan of a step)
        auto step = [&](auto PHC, const int s, const int G) {
            constexpr int ph = decltype(PHC)::value;
            constexpr bool PROJ = CT > 0 && ph < KBLK && !(BAR16D_ABL & 4);
            // ------------------------------ interval A ------------------------------
            DSTAMP_IN(2 * ph)
            if constexpr (BAR16D_ABL & 16) lds_bar(); else if constexpr (MIX) lds_bar_1read(); else lds_bar_2reads();
            DSTAMP_OUT(2 * ph)
            half8 bh[KBS], bl[KBS];                      // MIX: bh = the mixed operands, bl unused
            bh[0] = oh;
            bl[0] = ol;
#pragma unroll
            for (int i = 1; i < KBS; i++) {
                if constexpr (MIX) bh[i] = ldH(h_img, moff[i]);
                else { bh[i] = ldH(h_hi, boff[i]); bl[i] = ldH(h_lo, boff[i]); }
            }
            if (s > 0) {                                 // h(s-1), still in `hold` (gru_bar16.hip: stored behind the barrier, not in front of it)
                if (live && s - 1 < Tc && !(BAR16D_ABL & 8)) {
                    *reinterpret_cast<f32x2d *>(hp) = f32x2d{hold[0][0], hold[0][1]};
                    *reinterpret_cast<f32x2d *>(hp + 16) = f32x2d{hold[1][0], hold[1][1]};
                    if constexpr (SAVE) {
                        *reinterpret_cast<f32x2d *>(zp) = f32x2d{zkeep[0][0], zkeep[0][1]};
                        *reinterpret_cast<f32x2d *>(zp + 16) = f32x2d{zkeep[1][0], zkeep[1][1]};
                    }
                }
                hp += hstep;
                if constexpr (SAVE) zp += zstep;
            }
            // vI(s): complete since the previous barrier at the latest (the service waves use every interval)
            const float *vcur = vbuf + ((s % R) * 2 + set) * VSTEP + voff;
            f32x2d vz[2], vr[2], vc[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                vr[p] = *reinterpret_cast<const f32x2d *>(vcur + 64 * (NT + 2 * w + p));
                vz[p] = *reinterpret_cast<const f32x2d *>(vcur + 64 * (2 * w + p));
                vc[p] = *reinterpret_cast<const f32x2d *>(vcur + 64 * (2 * NT + 2 * w + p));
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accR[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accZ[2];
            mfma_rc(wr_hi, wr_lo, 0, bh[0], bl[0], accR[0], accR[1]);
            if constexpr (PROJ) {                        // my tiles of the projection, K block ph, both sets: inside the LDS round trip
#pragma unroll
                for (int sset = 0; sset < 2; sset++) {
#pragma unroll
                    for (int t = 0; t < CT; t++) block_mfma_acc<ph == 0>(pacc[sset][t], pw_hi[t][ph], pw_lo[t][ph], pxh[sset], pxl[sset]);
                }
            }
            if constexpr (CT > 0 && ph == 3) {           // vI of group G+1 (its last MFMAs were issued a step ago unless KBLK = 4)
#pragma unroll
                for (int sset = 0; sset < 2; sset++) {
#pragma unroll
                    for (int t = 0; t < CT; t++) {
                        if constexpr (KBLK == 4) mfma_drain(pacc[sset][t]);
                        proj_out(w * CT + t, pacc[sset][t], G + 1, sset);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KBS > 1) {
                asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");       // the six vI reads may still be on their way
#pragma unroll
                for (int i = 1; i < KBS; i++) { keep(bh[i]); if constexpr (!MIX) keep(bl[i]); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 1; i < KBS; i++) mfma_rc(wr_hi, wr_lo, i, bh[i], bl[i], accR[0], accR[1]);
                // the r products in FRONT of the update gate's asm MFMAs, which supply the wait states between them and the asm reads of
                // pick_mix_d (instruction selection places an MFMA anywhere its operands allow, sched_barrier or not; volatile
                // statements keep their order, and this one hands the accumulators on)
                asm volatile("" : "+v"(accR[0]), "+v"(accR[1]));
                __builtin_amdgcn_sched_barrier(0);
            }
            // z products of all blocks but the last INSIDE the r epilogue: the asm MFMAs are not the compiler's to place (it put
            // them behind the sigmoids, back to back in front of the LDS write the other waves wait for), so the epilogue is cut
            // into pieces that follow every other MFMA: sigmoid(r) per value, r*h, the two splits, the write
            float rr[2][2];
            uint2 rhi, rlo;
            auto r_piece = [&](auto HC) {
                constexpr int i = decltype(HC)::value;
                constexpr int NH = 6 * (KBS - 1);                     // hooks available (0: the whole epilogue afterwards)
                constexpr int stride = NH >= 12 ? 2 : 1;
                if constexpr (NH >= 6) {
                    if constexpr (i == 0 * stride) rr[0][0] = sigmoid4(fmaf(pick(accR[0], 0), inv_r[0][0], vr[0][0]));
                    if constexpr (i == 1 * stride) rr[1][0] = sigmoid4(fmaf(pick(accR[1], 0), inv_r[1][0], vr[1][0]));
                    if constexpr (i == 2 * stride) rr[0][1] = sigmoid4(fmaf(pick(accR[0], 1), inv_r[0][1], vr[0][1]));
                    if constexpr (i == 3 * stride) rr[1][1] = sigmoid4(fmaf(pick(accR[1], 1), inv_r[1][1], vr[1][1]));
                    if constexpr (i == 4 * stride) split2(rr[0][0] * hold[0][0], rr[1][0] * hold[1][0], rhi.x, rlo.x);
                    if constexpr (i == 5 * stride) split2(rr[0][1] * hold[0][1], rr[1][1] * hold[1][1], rhi.y, rlo.y);
                }
            };
            if constexpr (KBS > 1 && BAR16D_ZHOOK) {
                static_for<0, KBS - 1>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    z_block_mfma_hooked<i == 0, 6 * i>(accZ[0], accZ[1], wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh[i], bl[i], r_piece);
                });
            } else {
                if constexpr (BAR16D_CACC) {
                    accZ[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    accZ[1] = accZ[0];
#pragma unroll
                    for (int i = 0; i < KBS - 1; i++)
                        mfma3x2(wz_hi[0][i], wz_lo[0][i], wz_hi[1][i], wz_lo[1][i], bh[i], bl[i], accZ[0], accZ[1]);
                } else if constexpr (KBS > 1) {
                    mfma_z(ic<1>{}, 0, bh[0], bl[0], accZ[0], accZ[1]);
#pragma unroll
                    for (int i = 1; i < KBS - 1; i++) mfma_z(ic<0>{}, i, bh[i], bl[i], accZ[0], accZ[1]);
                }
                // pick_mix_d reads the accumulators from asm, where the compiler keeps no distance to the MFMAs that wrote them: with
                // more than one K block the z products lie in between, otherwise let the pipe drain
                if constexpr (MIX && KBS == 1) mfma_drain2(accR[0], accR[1]);
#pragma unroll
                for (int p = 0; p < 2; p++) {
#pragma unroll
                    for (int j = 0; j < 2; j++) rr[p][j] = sigmoid4(fmaf(pick(accR[p], j), inv_r[p][j], vr[p][j]));
                }
                split2(rr[0][0] * hold[0][0], rr[1][0] * hold[1][0], rhi.x, rlo.x);
                split2(rr[0][1] * hold[0][1], rr[1][1] * hold[1][1], rhi.y, rlo.y);
                if constexpr (BAR16D_CACC) {             // one MFMA, then up to four VALU instructions, for as long as both last
#pragma unroll
                    for (int i = 0; i < 6 * (KBS - 1); i++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                }
            }
            lds_fence();
            *reinterpret_cast<uint2 *>(&rh_hi[wd]) = rhi;
            *reinterpret_cast<uint2 *>(&rh_lo[wd]) = rlo;
            half8 ch[KBS], cl[KBS];                      // MIX: ch = the mixed operands, cl unused
            if constexpr (MIX) {
                ch[0] = ldH(rh_img, moff[0]);            // my own block, straight back (LDS executes a wave's operations in order)
                cl[0] = ch[0];
            } else {
                ch[0] = ldH(rh_hi, boff[0]);
                cl[0] = ldH(rh_lo, boff[0]);
            }
            lds_fence();
            const bool store = live && s < Tc && !(BAR16D_ABL & 8);
            if constexpr (SAVE) {
                if (store) {
                    *reinterpret_cast<f32x2d *>(zp + N) = f32x2d{rr[0][0], rr[0][1]};
                    *reinterpret_cast<f32x2d *>(zp + N + 16) = f32x2d{rr[1][0], rr[1][1]};
                }
            }
            // ------------------------------ interval B ------------------------------
            DSTAMP_IN(2 * ph + 1)
            if constexpr (BAR16D_ABL & 16) lds_bar(); else if constexpr (MIX) lds_bar_1read(); else lds_bar_2reads();
            DSTAMP_OUT(2 * ph + 1)
#pragma unroll
            for (int i = 1; i < KBS; i++) {
                if constexpr (MIX) ch[i] = ldH(rh_img, moff[i]);
                else { ch[i] = ldH(rh_hi, boff[i]); cl[i] = ldH(rh_lo, boff[i]); }
            }
            constexpr int nph = (ph + 1) & 3;            // the next step projects K block nph of the group after ITS group
            constexpr bool NPROJ = CT > 0 && nph < KBLK;
            half8 xh[2], xl[2];
            if constexpr (NPROJ) {
#pragma unroll
                for (int sset = 0; sset < 2; sset++) {
                    const int ob = opimg(G + (ph == 3 ? 2 : 1), sset) + 64 * nph;
                    xh[sset] = ldH(xop_hi, ob);
                    xl[sset] = ldH(xop_lo, ob);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accC[2];
            if constexpr (BAR16D_CACC) {
                if constexpr (KBS == 1) { accZ[0] = f32x4{0.f, 0.f, 0.f, 0.f}; accZ[1] = accZ[0]; }
                mfma3x2(wz_hi[0][KBS - 1], wz_lo[0][KBS - 1], wz_hi[1][KBS - 1], wz_lo[1][KBS - 1], bh[KBS - 1], bl[KBS - 1], accZ[0], accZ[1]);
                z_block_mfma<true>(accC[0], accC[1], wc_hi[0][0], wc_lo[0][0], wc_hi[1][0], wc_lo[1][0], ch[0], cl[0]);
            } else {
                if constexpr (KBS > 1) mfma_z(ic<0>{}, KBS - 1, bh[KBS - 1], bl[KBS - 1], accZ[0], accZ[1]);
                else mfma_z(ic<1>{}, 0, bh[0], bl[0], accZ[0], accZ[1]);
                accC[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                accC[1] = accC[0];
                mfma_rc(wc_hi, wc_lo, 0, ch[0], cl[0], accC[0], accC[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 1; i < KBS; i++) { keep(ch[i]); if constexpr (!MIX) keep(cl[i]); }
            if constexpr (NPROJ) {
#pragma unroll
                for (int sset = 0; sset < 2; sset++) { keep(xh[sset]); keep(xl[sset]); pxh[sset] = xh[sset]; pxl[sset] = xl[sset]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BAR16D_CACC) {
                // the candidate's products with the other waves' blocks: asm, in source order, the update gate's epilogue behind the first
                static_for<1, KBS>([&](auto IC) {
                    constexpr int i = decltype(IC)::value;
                    z_block_mfma<false>(accC[0], accC[1], wc_hi[0][i], wc_lo[0][i], wc_hi[1][i], wc_lo[1][i], ch[i], cl[i]);
                });
            } else {
                if constexpr (KBS > 1) {
                    mfma_rc(wc_hi, wc_lo, 1, ch[1], cl[1], accC[0], accC[1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the z accumulators came from asm MFMAs the compiler does not know as such: twelve MFMAs (or the drain) have been
                // issued since the last of them, and nothing that reads them may move above this point
                // (... and the candidate's MFMAs issued so far lie between them and those reads: the statement hands their accumulators on too)
                if constexpr (KBS == 1) mfma_drain2(accZ[0], accZ[1]);
                else asm volatile("" : "+v"(accZ[0]), "+v"(accZ[1]), "+v"(accC[0]), "+v"(accC[1]));
#pragma unroll
                for (int i = 2; i < KBS; i++) mfma_rc(wc_hi, wc_lo, i, ch[i], cl[i], accC[0], accC[1]);
            }
            float zz[2][2], omz[2][2], zh[2][2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    zz[p][j] = sigmoid4(fmaf(pick(accZ[p], j), inv_z[p][j], vz[p][j]));
                    omz[p][j] = 1.0f - zz[p][j];
                    zh[p][j] = zz[p][j] * hold[p][j];
                    asm volatile("" : "+v"(zh[p][j]), "+v"(omz[p][j]));        // pinned here: not sunk to the blend below
                }
            }
            if constexpr (BAR16D_CACC) {
                __builtin_amdgcn_sched_barrier(0);
                mfma_drain(accC[0]);
                mfma_drain(accC[1]);
            } else {                                     // one MFMA, then up to four VALU instructions, for as long as both last
#pragma unroll
                for (int i = 0; i < 6 * (KBS - 2); i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MIX) mfma_drain2(accC[0], accC[1]);                  // the candidate's last MFMAs were issued just above
            float hn[2][2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const float hbar = tanh5(fmaf(pick(accC[p], j), inv_c[p][j], vc[p][j]));
                    hn[p][j] = fmaf(omz[p][j], hbar, zh[p][j]);               // layers.py:1020
                }
            }
            {
                uint2 hi, lo;
                split2(hn[0][0], hn[1][0], hi.x, lo.x);
                split2(hn[0][1], hn[1][1], hi.y, lo.y);
                lds_fence();
                *reinterpret_cast<uint2 *>(&h_hi[wd]) = hi;
                *reinterpret_cast<uint2 *>(&h_lo[wd]) = lo;
            }
            if constexpr (MIX) {
                oh = ldH(h_img, moff[0]);
                ol = oh;
            } else {
                oh = ldH(h_hi, boff[0]);
                ol = ldH(h_lo, boff[0]);
            }
            lds_fence();
            if constexpr (SAVE) {
#pragma unroll
                for (int p = 0; p < 2; p++) { zkeep[p][0] = zz[p][0]; zkeep[p][1] = zz[p][1]; }
            }
#pragma unroll
            for (int p = 0; p < 2; p++) { hold[p][0] = hn[p][0]; hold[p][1] = hn[p][1]; }
        };
        for (int G = 0; G < NG; G++) {
            const int s = GS * G;
            step(ic<0>{}, s, G);
            if (s + 1 < T) step(ic<1>{}, s + 1, G);
            if (s + 2 < T) step(ic<2>{}, s + 2, G);
            if (s + 3 < T) step(ic<3>{}, s + 3, G);
        }
        if (live && T - 1 < Tc && !(BAR16D_ABL & 8)) {   // h (and z) of the last step
            *reinterpret_cast<f32x2d *>(hp) = f32x2d{hold[0][0], hold[0][1]};
            *reinterpret_cast<f32x2d *>(hp + 16) = f32x2d{hold[1][0], hold[1][1]};
            if constexpr (SAVE) {
                *reinterpret_cast<f32x2d *>(zp) = f32x2d{zkeep[0][0], zkeep[0][1]};
                *reinterpret_cast<f32x2d *>(zp + 16) = f32x2d{zkeep[1][0], zkeep[1][1]};
            }
        }
        if constexpr (BAR16D_ABL & 16) {
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 8; i++) { slk_dbg_bar16d[wave][i] = dwork[i]; slk_dbg_bar16d[wave][8 + i] = dwait[i]; }
        }
    } else {
        // =================================================================================================
        // service waves: the rest of the projection; the leader (first of them) also runs the x DMA and splits x
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
        // straight into registers (the four lanes of a row and K block cover one 128-byte line) a group ahead of the split --
        // no staging in LDS, no LDS-DMA (measured here: ~150 cycles of the wave per 1-KiB request).  Ordinary loads: the
        // compiler waits for them where the split first uses them, a group later.
        int xbc[2], xTc[2];
#pragma unroll
        for (int sset = 0; sset < 2; sset++) {
            xbc[sset] = min(b0 + 4 * sset + pc, B - 1);
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
        // split of a set's rows
        float xs = 1.0f;
        float raw[KBLK][8];                              // the rows as read for the scale, kept for the split
        auto split_scale = [&](int G2, auto SC) {        // pass 1: the row's power-of-two scale
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
        half8 ahi, alo;                                  // the operand pieces of the K block being split
        auto split_piece = [&](int kb, int j) {          // pass 2, element j of K block kb
            const float v = raw[kb][j] * xs;
            const _Float16 h = (_Float16)v;
            ahi[j] = h;
            alo[j] = (_Float16)(v - (float)h);
        };
        auto split_store = [&](int G2, int sset, int kb) {
            const int ob = opimg(G2, sset) + 64 * kb;
            *reinterpret_cast<half8 *>(xop_hi + ob) = ahi;
            *reinterpret_cast<half8 *>(xop_lo + ob) = alo;
        };
        auto split_block = [&](int G2, int sset, int kb) {         // pass 2: K block kb -> operand images
            unsigned bhi[4], blo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) split2(raw[kb][2 * j] * xs, raw[kb][2 * j + 1] * xs, bhi[j], blo[j]);
            const int ob = opimg(G2, sset) + 64 * kb;
            *reinterpret_cast<uint4 *>(xop_hi + ob) = make_uint4(bhi[0], bhi[1], bhi[2], bhi[3]);
            *reinterpret_cast<uint4 *>(xop_lo + ob) = make_uint4(blo[0], blo[1], blo[2], blo[3]);
        };

        __syncthreads();
        if (leader) {
            static_for<0, 2>([&](auto GC) {
                static_for<0, 2>([&](auto SC) {
                    load_x(decltype(GC)::value, SC);
                    split_scale(decltype(GC)::value, SC);
#pragma unroll
                    for (int kb = 0; kb < KBLK; kb++) split_block(decltype(GC)::value, decltype(SC)::value, kb);
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
                    const int ob = opimg(G1, sset) + 64 * kb;
                    xh[sset][kb] = ldH(xop_hi, ob);
                    xl[sset][kb] = ldH(xop_lo, ob);
                }
            }
        };
        // one tile for both sets: the output constants are requested first, then all MFMAs (the matrix pipe stays busy through
        // the LDS round trip) with `hook` in their issue gaps, one drain, then the outputs
        auto project_tile = [&](auto T0, int G1, auto &&hook) {
            constexpr int t0 = decltype(T0)::value;
            const float xin0 = xinv_lds[((G1 & 1) * 2 + 0) * 16 + pcol], xin1 = xinv_lds[((G1 & 1) * 2 + 1) * 16 + pcol];
            const f32x4 iw0 = *reinterpret_cast<const f32x4 *>(&invw_lds[16 * (tile0 + t0) + 4 * kg]);
            const f32x4 bs0 = *reinterpret_cast<const f32x4 *>(&bias_lds[16 * (tile0 + t0) + 4 * kg]);
            f32x4 a0, a1;
            if constexpr (t0 < NA) {
                tile2_mfma_acc<KBLK>(a0, a1, pa_hi[t0], pa_lo[t0], xh[0], xl[0], xh[1], xl[1], hook);
            } else {                                     // weights in ordinary registers: builtins, scheduled by the compiler
                a0 = f32x4{0.f, 0.f, 0.f, 0.f};
                a1 = a0;
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) {
                    a0 = mfma3(pw_hi[t0 - NA][kb], pw_lo[t0 - NA][kb], xh[0][kb], xl[0][kb], a0);
                    a1 = mfma3(pw_hi[t0 - NA][kb], pw_lo[t0 - NA][kb], xh[1][kb], xl[1][kb], a1);
                }
                static_for<0, 6 * KBLK>([&](auto HC) { hook(HC); });
            }
            mfma_drain2(a0, a1);
            const int st = GS * G1 + pstep;
            float *dst = &vbuf[(st % R) * 2 * VSTEP + 64 * (tile0 + t0) + (kg * 4 + pc) * 4];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = fmaf(a0[r] * xin0, iw0[r], bs0[r]);
            *reinterpret_cast<f32x4 *>(dst) = o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = fmaf(a1[r] * xin1, iw0[r], bs0[r]);
            *reinterpret_cast<f32x4 *>(dst + VSTEP) = o;
        };
        auto no_hook = [](auto) {};
        constexpr int K0 = KBLK == 4 ? 0 : 1;            // set 0 is split in intervals K0 .. K0 + KBLK - 1, set 1 from 4 on
        // interval k of group G: the tiles of group G1 = G + 1; the leader splits one K block of group G + 2 in the issue gaps
        // of its first tile's MFMAs (elements after MFMAs 1, 3, 5, ..., the store after the last)
        auto project_interval = [&](auto KC, int G1) {
            constexpr int k = decltype(KC)::value;
            constexpr int lo = tile_first_d(ST, k), hi = tile_first_d(ST, k + 1);
            constexpr bool S0 = k >= K0 && k < K0 + KBLK, S1 = k >= 4 && k < 4 + KBLK;
            constexpr int sset = S1 ? 1 : 0, kb = S1 ? k - 4 : k - K0;
            if constexpr ((S0 || S1) && !(BAR16D_ABL & 2) && (hi == lo || !BAR16D_HOOKS)) {   // the split on its own, then the tiles
                static_for<lo, hi>([&](auto TC) { project_tile(TC, G1, no_hook); });
                if (leader && G1 > 0) {
                    if constexpr (kb == 0) split_scale(G1 + 1, ic<sset>{});
                    split_block(G1 + 1, sset, kb);
                    if constexpr (kb == KBLK - 1) load_x(G1 + 2, ic<sset>{});
                }
            } else if constexpr ((S0 || S1) && !(BAR16D_ABL & 2)) {
                if (leader && G1 > 0) {
                    if constexpr (kb == 0) split_scale(G1 + 1, ic<sset>{});
                    project_tile(ic<lo>{}, G1, [&](auto HC) {
                        constexpr int i = decltype(HC)::value;
                        if constexpr ((i & 1) && i < 16) split_piece(kb, i >> 1);
                        if constexpr (i == 6 * KBLK - 1) {
                            if constexpr (6 * KBLK < 16) {
#pragma unroll
                                for (int j = 3 * KBLK; j < 8; j++) split_piece(kb, j);
                            }
                            split_store(G1 + 1, sset, kb);
                        }
                    });
                    if constexpr (kb == KBLK - 1) load_x(G1 + 2, ic<sset>{});
                } else {
                    project_tile(ic<lo>{}, G1, no_hook);
                }
                static_for<lo + 1, hi>([&](auto TC) { project_tile(TC, G1, no_hook); });
            } else {
                static_for<lo, hi>([&](auto TC) { project_tile(TC, G1, no_hook); });
            }
        };
        load_operands(0);
        static_for<0, 8>([&](auto KC) { project_interval(KC, 0); });
        lds_bar();                                       // vI of group 0 complete

        // interval k = 0..7 of group G (two per step, each opened by the barrier the chain waves open theirs with)
        unsigned long long dwork[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dwait[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        if constexpr (BAR16D_ABL & 16) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
        auto interval = [&](auto KC, const int G) {
            constexpr int k = decltype(KC)::value;
            DSTAMP_IN(k)
            lds_bar();
            DSTAMP_OUT(k)
            if constexpr (BAR16D_ABL & 1) return;
            if constexpr (k == 0) load_operands(G + 1);
            project_interval(KC, G + 1);
        };
        for (int G = 0; G < NG; G++) {
            const int s = GS * G;
            interval(ic<0>{}, G); interval(ic<1>{}, G);
            if (s + 1 < T) { interval(ic<2>{}, G); interval(ic<3>{}, G); }
            if (s + 2 < T) { interval(ic<4>{}, G); interval(ic<5>{}, G); }
            if (s + 3 < T) { interval(ic<6>{}, G); interval(ic<7>{}, G); }
        }
        if constexpr (BAR16D_ABL & 16) {
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 8; i++) { slk_dbg_bar16d[wave][i] = dwork[i]; slk_dbg_bar16d[wave][8 + i] = dwait[i]; }
        }
    }
}

template <int I, int N>
static int launch_bar16d(const float *x, long ldx, const float *iW, const float *bias, const float *sW, const float *sW2,
                         float *y, long ldy, int T, int B, int reverse, const int *lens, float *zr_out, hipStream_t s)
{
    if (zr_out)
        hipLaunchKernelGGL((gru_bar16d_kernel<I, N, true>), dim3((B + 7) / 8), dim3(256), 0, s, x, ldx, iW, bias, sW, sW2, y, ldy, T,
                           B, reverse & 1, lens, zr_out);
    else
        hipLaunchKernelGGL((gru_bar16d_kernel<I, N, false>), dim3((B + 7) / 8), dim3(256), 0, s, x, ldx, iW, bias, sW, sW2, y, ldy, T,
                           B, reverse & 1, lens, zr_out);
    return slk_launch_status();
}

// The eight-chunk plan behind slk_gru_bar16_f32 (same contract); SLK_ERR_UNSUPPORTED when no instantiation covers the request
// (the caller then takes the four-chunk kernel).
extern "C" int slk_gru_bar16d_launch(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias, float *y,
                          long ldy, int T, int B, int insize, int n, int reverse, const int32_t *lens, float *zr_out, hipStream_t s)
{
    if ((ldy & 1) || (reinterpret_cast<uintptr_t>(y) & 7)) return SLK_ERR_UNSUPPORTED;          // 8-byte state stores
#define BAR16D(II, NN) \
    if (insize == II && n == NN) return launch_bar16d<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse, lens, zr_out, s);
    BAR16D(96, 96) BAR16D(64, 64) BAR16D(32, 96) BAR16D(128, 96) BAR16D(64, 96) BAR16D(48, 32) BAR16D(16, 64)
#undef BAR16D
    return SLK_ERR_UNSUPPORTED;
}
