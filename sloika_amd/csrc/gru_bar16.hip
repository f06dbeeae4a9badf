// gru_bar16.hip -- a whole Gru layer (sloika/layers.py:1010-1021) in one persistent kernel of FOUR waves per workgroup,
// one per SIMD, in lock step: the same arithmetic as gru_fused16.hip (3-term fp16 split products, float32 accumulation,
// rows scaled by powers of two) on a different execution plan.
//
// gru_fused16.hip runs eight waves coupled by progress counters in LDS; a step of its chain costs two counter round trips
// (poll, ballot, retry), shares its SIMDs with the projection waves, and lives in 256 registers.  Measured there: of
// ~2200 cycles per step only ~860 are the chain's MFMAs.  Here:
//   * one wave per SIMD (256 threads, 512 registers each): nobody competes with the chain for issue slots;
//   * the two exchanges of a step (r*h, then h) are two s_barrier -- LDS data written before the barrier is simply there
//     after it, no counters, no retries;
//   * the time-parallel projection vI = x.iW^T + b (four steps at a time) is cut into pieces that the waves execute at
//     fixed places of the step: a chain wave issues its share (CT tiles, one K block per step) right after a barrier,
//     while its LDS reads of the state are in flight; the service waves (the SIMDs without a chain wave) take the rest,
//     split x into fp16 halves ONCE per group for everybody (operand images in LDS), and run the x DMA one request at a
//     time so that they always reach the next barrier before the chain does;
//   * every chain lane owns (neuron, chunk) pairs and stores its new state straight to h_out (a lane quartet writes 16
//     consecutive bytes, a wave a whole 128-byte line per chunk and step): no staging ring, no copy-out pass.
// Layout of the 16x16x32 tiles (weights = A, state of the 4 chunks in all four column groups = B, lane (c, q, g) keeps
// neuron 4g+q of its tile) and the packed operand images are those of gru_fused16.hip.
#include <limits.h>
#include <stdlib.h>

#include "bar16_common.h"

// Recurrent products take TWO MFMAs each: the state's hi and lo halves ride in different column groups (bar16_common.h: pick_mix).
//
// Diagnostics (per-section shader-clock stamps, ablation launches, workgroup clocks) exist only in builds with -DSLK_DIAG
// (tools/build_diag_lib.sh; readers: tools/bar16_check.py, tools/bar16_wg_times.py).  ABL bits (results are then garbage):
// 1 = no s_barrier, 2 = chain waves issue no MFMAs, 4 = cheap activations, 8 = service waves only keep the barriers,
// 16 = no stores to h_out, 32 = every workgroup records where it ran and for how long.
#ifdef SLK_DIAG
__device__ unsigned long long slk_dbg_bar16[4][16];
extern "C" SLK_API int slk_debug_read_bar16(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_bar16), sizeof(unsigned long long) * 64) == hipSuccess ? SLK_OK
                                                                                                                : SLK_ERR_LAUNCH;
}
__device__ unsigned long long slk_dbg_bar16_wg[1024][4];
extern "C" SLK_API int slk_debug_read_bar16_wg(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_bar16_wg), sizeof(unsigned long long) * 4096) == hipSuccess ? SLK_OK
                                                                                                                    : SLK_ERR_LAUNCH;
}
#define BSTAMP(i)                                                                     \
    if constexpr (DIAG) {                                                             \
        unsigned long long tnow;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        sacc[i] += tnow - tprev;                                                      \
        tprev = tnow;                                                                 \
    }
#else
#define BSTAMP(i)
#endif

// first tile of interval k when a service wave has st tiles per group (k = 8: st)
__host__ __device__ constexpr int tile_first(int st, int k)
{
    // share per interval: lighter where the leader also splits x (intervals 1..4) and fetches operands (0)
    constexpr int w[8] = {1, 1, 1, 1, 2, 2, 2, 2};
    int tot = 0, acc = 0;
    for (int i = 0; i < 8; i++) tot += w[i];
    for (int i = 0; i < k && i < 8; i++) acc += w[i];
    return k >= 8 ? st : (acc * st + tot / 2) / tot < st ? (acc * st + tot / 2) / tot : st;
}

template <int I, int N, bool SAVE, bool DIAG = false, int ABL = 0>
__global__ void __launch_bounds__(256, 1) gru_bar16_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ iW,
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
    constexpr int GS = 4;                                // steps per projection group (16 MFMA columns = 4 steps x 4 chunks)
    constexpr int R = 2 * GS;                            // vI ring: group G+1 is written while group G is consumed
    // projection tiles of a chain wave (weights in accumulation registers); three where four K blocks of the input would not leave
    // the service wave room for the tiles it keeps in ordinary registers next to the x rows it loads (128 -> 96 spilled)
    constexpr int CT = NCW == 3 ? (KBLK == 4 ? 3 : 2) : 0;
    constexpr int ST = (NT16 - NCW * CT) / NSW;                         // ... of a service wave
    constexpr int NACAP = 240 / (8 * KBLK);                              // 256 accumulation registers, 2 * KBLK * 4 per tile
    constexpr int NA = ST < NACAP ? ST : NACAP;                               // of which this many keep their weights in accumulation registers
    static_assert(NCW * CT + NSW * ST == NT16, "tile assignment");
    static_assert(KBLK <= 4 && ST <= 21, "interval plan");
    // dwords of one operand image: [step][k block][k group][chunk][8 halves]; the steps lie 32 banks apart so that the 16-lane
    // groups of a ds_read_b128 (lanes of two k groups and two steps each) find their pieces on different banks
    constexpr int OPSTEP = KBLK * 64 + 32;
    constexpr int OPIMG = GS * OPSTEP;
    // floats of one step's vI: [tile][g][chunk][r], row = 16 tile + 4g + r; + 16: the steps of a projection tile (one per lane
    // quartet of its 16-byte writes) on different banks
    constexpr int VSTEP = NT16 * 64 + 16;

    __shared__ __attribute__((aligned(16))) unsigned xop_hi[2 * OPIMG], xop_lo[2 * OPIMG];
    __shared__ __attribute__((aligned(16))) float xinv_lds[2 * 16];
    __shared__ __attribute__((aligned(16))) float vbuf[R * VSTEP];
    // The lo image lies 32 banks behind the hi image: a ds_read_b128 of the mixed operand serves lane quartets of both in one pass
    // (2N + 4 put the two on the same banks: SQ_LDS_BANK_CONFLICT was 37 % of the kernel's LDS cycles)
    constexpr int IMG = 2 * N + (2 * N % 64 == 0 ? 32 : 2 * N % 64 == 32 ? 0 : 4);
    __shared__ __attribute__((aligned(16))) unsigned h_img[2 * IMG], rh_img[2 * IMG];             // hi image, then lo image
    unsigned *const h_hi = h_img, *const h_lo = h_img + IMG, *const rh_hi = rh_img, *const rh_lo = rh_img + IMG;
    __shared__ __attribute__((aligned(16))) float bias_lds[3 * N], invw_lds[3 * N];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
#ifdef SLK_DIAG
    unsigned long long wg_t0 = 0, wg_r0 = 0;
    if constexpr (ABL & 32) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0), "=s"(wg_r0)::"memory");
#endif

    for (int i = tid; i < 2 * N; i += 256) { h_hi[i] = 0u; h_lo[i] = 0u; }             // h(-1) = 0
    for (int i = tid; i < 3 * N; i += 256) bias_lds[i] = bias ? bias[i] : 0.0f;

    // ---------------- projection pieces shared by both kinds of wave ----------------
    const int pcol = lane & 15, kg = lane >> 4;          // operand row / column and k group of this lane
    const int pstep = pcol >> 2, pc = pcol & 3;          // as a B column: (step in group, chunk)
    const int poff = pstep * OPSTEP + kg * 16 + pc * 4;                 // + 64 kb: my 16 bytes of an operand image, in dwords
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };
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
    // accumulator of a tile for group G1 -> vI ring: lane holds rows 4 kg + r of column (pstep, pc)
    auto proj_out = [&](int tile, const f32x4 &acc, int G1) {
        const float xin = xinv_lds[(G1 & 1) * 16 + pcol];
        const f32x4 iw = *reinterpret_cast<const f32x4 *>(&invw_lds[16 * tile + 4 * kg]);
        const f32x4 bs = *reinterpret_cast<const f32x4 *>(&bias_lds[16 * tile + 4 * kg]);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = fmaf(acc[r] * xin, iw[r], bs[r]);
        const int st = GS * G1 + pstep;
        *reinterpret_cast<f32x4 *>(&vbuf[(st % R) * VSTEP + ((tile * 4 + kg) * 4 + pc) * 4]) = o;
    };
    const int NG = (T + GS - 1) / GS;

    if (wave < NCW) {
        // =================================================================================================
        // chain waves
        // =================================================================================================
        const int w = wave;
        const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
        // recurrent weights: A operands, K blocks in the rotated order w, w+1, ... (element (g, j) of block kb is neuron
        // 32 kb + 16 (j&1) + 4 g + (j>>1), the order the owners' packed writes create), rows scaled to [1, 2)
        half8 wz_hi[2][KBS], wz_lo[2][KBS], wr_hi[2][KBS], wr_lo[2][KBS], wc_hi[2][KBS], wc_lo[2][KBS];
        float inv_z[2], inv_r[2], inv_c[2];
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
        constexpr int CTA = CT > 0 ? CT : 1;
        half8 pw_hi[CTA][KBLK], pw_lo[CTA][KBLK];
        f32x4 pacc[CTA];
        if constexpr (CT > 0) {
#pragma unroll
            for (int t = 0; t < CT; t++) {
                load_tile(w * CT + t, pw_hi[t], pw_lo[t]);
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) { pw_hi[t][kb] = to_acc_regs(pw_hi[t][kb]); pw_lo[t][kb] = to_acc_regs(pw_lo[t][kb]); }
            }
        }
        // my 16 bytes of K block (w + i) % KBS in MY column group's image (q = 0, 1: hi; q = 2, 3: lo), in dwords
        int moff[KBS];
#pragma unroll
        for (int i = 0; i < KBS; i++) moff[i] = (q >> 1) * IMG + ((((w + i) % KBS) * 4 + g) * 4 + c) * 4;
        const int wd = ((w * 4 + g) * 4 + c) * 4 + q;                                           // my packed pair, in dwords
        const int n0 = 32 * w + 4 * g + q;                                                      // my neuron of tile 2w (+16: 2w+1)
        const int voff = (g * 4 + c) * 4 + q;                                                   // my element of a vI tile
        // my chunk's rows of h_out (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
        const int bc = b0 + c;
        const bool live = bc < B;
        const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
        const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
        float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + (live ? bc : 0)) * ldh + n0;
        const long zstep = (reverse ? -1L : 1L) * (long)B * 2 * N;
        float *zp = SAVE ? zr_out + ((size_t)(reverse ? Tc - 1 : 0) * B + (live ? bc : 0)) * (2 * N) + n0 : nullptr;

        __syncthreads();                                 // LDS initialised, every wave's invw_lds rows written
        lds_bar();                                       // x operand images of groups 0 and 1 (service leader)
        half8 pxh = {0, 0, 0, 0, 0, 0, 0, 0}, pxl = pxh;    // x operands of the coming step's share of the projection
        settle(pxh);
        settle(pxl);
        if constexpr (CT > 0) {                          // vI of group 0
            half8 xh0[KBLK], xl0[KBLK];
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) { xh0[kb] = ldH(xop_hi, poff + 64 * kb); xl0[kb] = ldH(xop_lo, poff + 64 * kb); }
#pragma unroll
            for (int t = 0; t < CT; t++) {
                pacc[t] = tile_mfma_acc<KBLK>(pw_hi[t], pw_lo[t], xh0, xl0);
                mfma_drain(pacc[t]);
                proj_out(w * CT + t, pacc[t], 0);
            }
            pxh = ldH(xop_hi, OPIMG + poff);             // step 0 projects K block 0 of group 1
            pxl = ldH(xop_lo, OPIMG + poff);
        }
        lds_bar();                                       // vI of group 0 complete

        [[maybe_unused]] unsigned long long sacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        if constexpr (DIAG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        float hold[2] = {0.0f, 0.0f};
        [[maybe_unused]] float zkeep[2] = {0.0f, 0.0f};  // SAVE: the update gate of the step before, stored with its h
        // carried from step to step: my own K block of h(s-1) as B operand, read back right after I wrote it
        half8 oh = {0, 0, 0, 0, 0, 0, 0, 0};
        settle(oh);
        // registers the asm statements of a step write, kept from step to step (bar16_common.h: pick_mix_kept)
        float pk0 = 0.0f, pk1 = 0.0f;
        unsigned sp_hi = 0u, sp_lo = 0u;
        // One step = two intervals, each opened by a barrier; MFMAs are issued in an order that keeps the matrix pipe busy
        // through every LDS round trip and every stretch of gate arithmetic (an MFMA occupies the pipe for 16 cycles and
        // the issuing wave for 4):
        //   A  [others' h(s-1) visible]  request the other K blocks, then vI(s); r products with my own block (already in
        //      registers) and this step's share of the projection while they fly; r products with the others, tile 0 first;
        //      z products (all but the last block) under sigmoid(r), r*h, split, write, own block read back
        //   B  [others' r*h visible]     request the other K blocks, then the x operands of the next step's share of the
        //      projection; last z block and candidate products with my own block while they fly; candidate products with
        //      the others (tile 0 first), sigmoid(z) in their shadow; tanh, blend, split, write, own block read back, store
        // The gate arithmetic reads the accumulators from inline asm (pick_mix), where hipcc inserts no wait states, and the
        // hardware does not interlock a vector read of an MFMA result: seven wait states must lie between a
        // v_mfma_f32_16x16x32_f16 and the read (tools/probes/mfma_read_hazard_probe.hip).  Every tile's last MFMA is therefore
        // pinned (sched_barrier) in front of at least eight wait states of other instructions: tile 1's MFMAs for tile 0 --
        // which also hides their latency behind tile 0's arithmetic -- and tile 0's pick for tile 1.
        auto mfma2 = [](const half8 &w_hi, const half8 &w_lo, const half8 &bm, f32x4 &acc) {
            if constexpr (ABL & 2) {
                half8 a = w_hi, b = bm;
                asm volatile("" : "+v"(a), "+v"(b), "+v"(acc));
            } else {
                ::mfma2(w_hi, w_lo, bm, acc);
            }
        };
        // wait states in front of tile 0's pick when 2 (KBS - 1) MFMAs of tile 1 (at least two: its own block) follow tile 0's last
        constexpr int WS0 = KBS > 1 ? (8 - 2 * (KBS - 1) > 2 ? 8 - 2 * (KBS - 1) : 2) : 6;
        constexpr int WS1 = (8 - WS0 - 4) > 2 ? (8 - WS0 - 4) : 2;         // ... of tile 1's, behind tile 0's pick (WS0 + four reads)
        auto step = [&](auto PHC, const int s, const int G) {
            constexpr int ph = decltype(PHC)::value;
            constexpr bool PROJ = CT > 0 && ph < KBLK;
            // ------------------------------ interval A ------------------------------
            if constexpr (DIAG) lds_bar(); else lds_bar_1read<!(ABL & 1)>();
            BSTAMP(0)
            half8 bh[KBS];
            bh[0] = oh;
#pragma unroll
            for (int i = 1; i < KBS; i++) bh[i] = ldH(h_img, moff[i]);       // what the step waits for is requested first
            __builtin_amdgcn_sched_barrier(0);
            if (s > 0) {                                 // h(s-1): still in `hold`
                if (live && s - 1 < Tc && !(ABL & 16)) {
                    hp[0] = hold[0];
                    hp[16] = hold[1];
                    if constexpr (SAVE) { zp[0] = zkeep[0]; zp[16] = zkeep[1]; }
                }
                hp += hstep;
                if constexpr (SAVE) zp += zstep;
            }
            __builtin_amdgcn_sched_barrier(0);
            // vI(s): complete since the previous barrier at the latest (the service waves use every interval)
            const float *vcur = vbuf + (s % R) * VSTEP + voff;
            float vz[2], vr[2], vc[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                vr[p] = vcur[64 * (NT + 2 * w + p)];
                vz[p] = vcur[64 * (2 * w + p)];
                vc[p] = vcur[64 * (2 * NT + 2 * w + p)];
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 accR[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accZ[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma2(wr_hi[0][0], wr_lo[0][0], bh[0], accR[0]);
            mfma2(wr_hi[1][0], wr_lo[1][0], bh[0], accR[1]);
            if constexpr (PROJ) {                        // my tile of the projection, K block ph: inside the LDS round trip
                if constexpr (!(ABL & 2)) {
#pragma unroll
                    for (int t = 0; t < CT; t++) block_mfma_acc<ph == 0>(pacc[t], pw_hi[t][ph], pw_lo[t][ph], pxh, pxl);
                }
            }
            if constexpr (CT > 0 && ph == 3) {           // vI of group G+1 (its last MFMAs were issued a step ago unless KBLK = 4)
#pragma unroll
                for (int t = 0; t < CT; t++) {
                    if constexpr (KBLK == 4) mfma_drain(pacc[t]);
                    proj_out(w * CT + t, pacc[t], G + 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            BSTAMP(1)
#pragma unroll
            for (int i = 1; i < KBS; i++) mfma2(wr_hi[0][i], wr_lo[0][i], bh[i], accR[0]);
            // tile 0 complete BEFORE tile 1's remaining MFMAs are issued (WS0 counts on them).  Instruction selection places an MFMA
            // anywhere its operands allow, sched_barrier or not; these statements (volatile: they keep their order) hand the
            // accumulators on, so the MFMAs in front of one and behind the next cannot change sides.
            asm volatile("" : "+v"(accR[0]));
            asm volatile("" : "+v"(accR[1]));
#pragma unroll
            for (int i = 1; i < KBS; i++) mfma2(wr_hi[1][i], wr_lo[1][i], bh[i], accR[1]);
            asm volatile("" : "+v"(accR[1]));
            asm volatile("" : "+v"(accZ[0]), "+v"(accZ[1]));                 // the z products: behind the r products
            __builtin_amdgcn_sched_barrier(0);
            BSTAMP(2)
#pragma unroll
            for (int i = 0; i < KBS - 1; i++) {
                mfma2(wz_hi[0][i], wz_lo[0][i], bh[i], accZ[0]);
                mfma2(wz_hi[1][i], wz_lo[1][i], bh[i], accZ[1]);
            }
            float rr[2];
            pick_mix_kept<WS0>(accR[0], pk0, accR[1][0]);                     // behind tile 1's MFMAs
            pick_mix_kept<WS1>(accR[1], pk1, pk0);                            // behind tile 0's pick
            rr[0] = (ABL & 4) ? fmaf(pk0, inv_r[0], vr[0]) * 0.01f : sigmoid4(fmaf(pk0, inv_r[0], vr[0]));
            rr[1] = (ABL & 4) ? fmaf(pk1, inv_r[1], vr[1]) * 0.01f : sigmoid4(fmaf(pk1, inv_r[1], vr[1]));
            split2_kept(rr[0] * hold[0], rr[1] * hold[1], sp_hi, sp_lo);
            lds_fence();
            rh_hi[wd] = sp_hi;
            rh_lo[wd] = sp_lo;
            // the accumulators stay allocated until here: a value that moved into their registers right behind the picks would
            // make the compiler pad for the MFMAs it knows wrote them (it counts an asm statement as one wait state)
            asm volatile("" ::"v"(accR[0]), "v"(accR[1]));
            half8 ch[KBS];
            ch[0] = ldH(rh_img, moff[0]);                // my own block, straight back (LDS executes a wave's operations in order)
            lds_fence();
            // one MFMA, then up to three VALU instructions, for as long as both last
#pragma unroll
            for (int i = 0; i < 4 * (KBS - 1); i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            const bool store = live && s < Tc && !(ABL & 16);
            // ------------------------------ interval B ------------------------------
            if constexpr (DIAG) { BSTAMP(3) lds_bar(); } else lds_bar_1read<!(ABL & 1)>();
            BSTAMP(4)
#pragma unroll
            for (int i = 1; i < KBS; i++) ch[i] = ldH(rh_img, moff[i]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SAVE) {                        // (behind the barrier, like h: while the requested operands are on their way)
                if (store) { zp[N] = rr[0]; zp[N + 16] = rr[1]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            constexpr int nph = (ph + 1) & 3;            // the next step projects K block nph of the group after ITS group
            constexpr bool NPROJ = CT > 0 && nph < KBLK;
            if constexpr (NPROJ) {
                const int ob = ((G + (ph == 3 ? 2 : 1)) & 1) * OPIMG + poff + 64 * nph;
                pxh = ldH(xop_hi, ob);
                pxl = ldH(xop_lo, ob);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma2(wz_hi[0][KBS - 1], wz_lo[0][KBS - 1], bh[KBS - 1], accZ[0]);
            mfma2(wz_hi[1][KBS - 1], wz_lo[1][KBS - 1], bh[KBS - 1], accZ[1]);
            f32x4 accC[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma2(wc_hi[0][0], wc_lo[0][0], ch[0], accC[0]);
            mfma2(wc_hi[1][0], wc_lo[1][0], ch[0], accC[1]);
            // everything below -- sigmoid(z) and the other blocks' candidate products -- behind these eight MFMAs
            asm volatile("" : "+v"(accZ[0]), "+v"(accZ[1]), "+v"(accC[0]), "+v"(accC[1]));
            __builtin_amdgcn_sched_barrier(0);
            BSTAMP(5)
#pragma unroll
            for (int i = 1; i < KBS; i++) mfma2(wc_hi[0][i], wc_lo[0][i], ch[i], accC[0]);
            asm volatile("" : "+v"(accC[0]));                                // as above: tile 0's sum before tile 1's
            asm volatile("" : "+v"(accC[1]));
#pragma unroll
            for (int i = 1; i < KBS; i++) mfma2(wc_hi[1][i], wc_lo[1][i], ch[i], accC[1]);
            asm volatile("" : "+v"(accC[1]));
            // sigmoid(z): its last MFMAs were issued in front of the candidate's own-block products (four MFMAs ago at least)
            float zz[2], omz[2], zh[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                float &pz = p ? pk1 : pk0;
                pick_mix_kept<4>(accZ[p], pz, accZ[1][0]);
                zz[p] = (ABL & 4) ? fmaf(pz, inv_z[p], vz[p]) * 0.01f : sigmoid4(fmaf(pz, inv_z[p], vz[p]));
                omz[p] = 1.0f - zz[p];
                zh[p] = zz[p] * hold[p];
                asm volatile("" : "+v"(zh[p]), "+v"(omz[p]));                 // pinned here: not sunk to the blend below
            }
#pragma unroll
            for (int i = 0; i < 4 * (KBS - 1); i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            BSTAMP(6)
            float hn[2];
            {
                pick_mix_kept<WS0>(accC[0], pk0, accC[1][0]);
                pick_mix_kept<WS1>(accC[1], pk1, pk0);
                const float h0 = (ABL & 4) ? fmaf(pk0, inv_c[0], vc[0]) * 0.01f : tanh5(fmaf(pk0, inv_c[0], vc[0]));
                const float h1 = (ABL & 4) ? fmaf(pk1, inv_c[1], vc[1]) * 0.01f : tanh5(fmaf(pk1, inv_c[1], vc[1]));
                hn[0] = fmaf(omz[0], h0, zh[0]);                              // layers.py:1020
                hn[1] = fmaf(omz[1], h1, zh[1]);
            }
            split2_kept(hn[0], hn[1], sp_hi, sp_lo);
            lds_fence();
            h_hi[wd] = sp_hi;
            h_lo[wd] = sp_lo;
            asm volatile("" ::"v"(accC[0]), "v"(accC[1]), "v"(accZ[0]), "v"(accZ[1]));
            oh = ldH(h_img, moff[0]);
            lds_fence();
            // (h(s) is stored by the NEXT step, behind its first barrier, while that step waits for the state it has requested from LDS;
            //  here the stores were two more instructions between the last write of the state and the barrier everybody waits at)
            if constexpr (SAVE) {
                zkeep[0] = zz[0];
                zkeep[1] = zz[1];
            }
#pragma unroll
            for (int p = 0; p < 2; p++) hold[p] = hn[p];
            BSTAMP(7)
        };
        for (int G = 0; G < NG; G++) {
            const int s = GS * G;
            step(ic<0>{}, s, G);
            if (s + 1 < T) step(ic<1>{}, s + 1, G);
            if (s + 2 < T) step(ic<2>{}, s + 2, G);
            if (s + 3 < T) step(ic<3>{}, s + 3, G);
        }
        if (live && T - 1 < Tc && !(ABL & 16)) {         // h (and z) of the last step
            hp[0] = hold[0];
            hp[16] = hold[1];
            if constexpr (SAVE) { zp[0] = zkeep[0]; zp[16] = zkeep[1]; }
        }
#ifdef SLK_DIAG
        if constexpr (DIAG) {
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 16; i++) slk_dbg_bar16[wave][i] = sacc[i];
        }
        if constexpr (ABL & 32) {
            unsigned long long t1, r1;
            unsigned hwid, xcc;
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_getreg_b32 %2, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %3, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                         : "=s"(t1), "=s"(r1), "=s"(hwid), "=s"(xcc)::"memory");
            if (wave == 0 && lane == 0 && blockIdx.x < 1024) {
                slk_dbg_bar16_wg[blockIdx.x][0] = t1 - wg_t0;
                slk_dbg_bar16_wg[blockIdx.x][1] = r1 - wg_r0;
                slk_dbg_bar16_wg[blockIdx.x][2] = ((unsigned long long)xcc << 32) | hwid;
                slk_dbg_bar16_wg[blockIdx.x][3] = wg_r0;
            }
        }
#endif
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

        // x of a group: the leader's lane (row pcol = (step, chunk), k group kg) loads ITS eight floats of every K block straight
        // into registers (the four lanes of a row and K block cover one 128-byte line) a group ahead of the split -- no staging
        // in LDS, no LDS-DMA (a 1-KiB request kept the issuing wave ~150 cycles, twelve of sixteen intervals carried one).
        // Ordinary loads: the compiler waits for them where the split first uses them, a group later.
        const int xbc = min(b0 + pc, B - 1);
        const int xTc = lens ? min(max(lens[xbc], 1), T) : T;
        f32x4 xr[KBLK][2];
        auto load_x = [&](int G2) {
            // steps past the chunk's end re-read its last valid row (their results are never stored)
            const int ss = min(G2 * GS + pstep, xTc - 1);
            const int tt = reverse ? xTc - 1 - ss : ss;
            const float *row = x + ((size_t)tt * B + xbc) * ldx;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kg;
                const bool kok = (I % 32 == 0) || k0 < I;
                const float *src = row + (kok ? k0 : 0);
                xr[kb][0] = *reinterpret_cast<const f32x4 *>(src);
                xr[kb][1] = *reinterpret_cast<const f32x4 *>(src + 4);
            }
        };
        float xs = 1.0f;
        float raw[KBLK][8];                              // the group's rows as read for the scale, kept for the split
        auto split_scale = [&](int G2) {                 // pass 1: the row's power-of-two scale
            float amax = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kg;
                const bool kok = (I % 32 == 0) || k0 < I;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    raw[kb][j] = kok ? xr[kb][0][j] : 0.0f;
                    raw[kb][4 + j] = kok ? xr[kb][1][j] : 0.0f;
                    amax = fmaxf(amax, fmaxf(fabsf(raw[kb][j]), fabsf(raw[kb][4 + j])));
                }
            }
            float xinv;
            xs = pow2_scale(kgroup_max(amax), xinv);
            if (kg == 0) xinv_lds[(G2 & 1) * 16 + pcol] = xinv;
        };
        auto split_block = [&](int G2, int kb) {         // pass 2: K block kb -> operand images
            unsigned ahi[4], alo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) split2(raw[kb][2 * j] * xs, raw[kb][2 * j + 1] * xs, ahi[j], alo[j]);
            const int ob = (G2 & 1) * OPIMG + poff + 64 * kb;
            *reinterpret_cast<uint4 *>(xop_hi + ob) = make_uint4(ahi[0], ahi[1], ahi[2], ahi[3]);
            *reinterpret_cast<uint4 *>(xop_lo + ob) = make_uint4(alo[0], alo[1], alo[2], alo[3]);
        };

        __syncthreads();
        if (leader) {
            for (int G2 = 0; G2 < 2; G2++) {
                load_x(G2);
                split_scale(G2);
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) split_block(G2, kb);
            }
            load_x(2);                                   // group 2: split during group 0
        }
        lds_bar();
        half8 xh[KBLK], xl[KBLK];
        auto load_operands = [&](int G1) {
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int ob = (G1 & 1) * OPIMG + poff + 64 * kb;
                xh[kb] = ldH(xop_hi, ob);
                xl[kb] = ldH(xop_lo, ob);
            }
        };
        // accumulator of one tile (no drain): accumulation-register weights through asm, the rest through the builtin
        auto tile_acc = [&](auto TC) {
            constexpr int t = decltype(TC)::value;
            if constexpr (t < NA) {
                return tile_mfma_acc<KBLK>(pa_hi[t], pa_lo[t], xh, xl);
            } else {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) acc = mfma3(pw_hi[t - NA][kb], pw_lo[t - NA][kb], xh[kb], xl[kb], acc);
                return acc;
            }
        };
        // two tiles: their output constants are requested first, then all MFMAs (the matrix pipe stays busy through the LDS
        // round trip), one drain, then the outputs
        auto project_tiles = [&](auto T0, auto T1, int G1) {
            constexpr int t0 = decltype(T0)::value, t1 = decltype(T1)::value;
            const float xin = xinv_lds[(G1 & 1) * 16 + pcol];
            f32x4 iw0, bs0, iw1, bs1;
            iw0 = *reinterpret_cast<const f32x4 *>(&invw_lds[16 * (tile0 + t0) + 4 * kg]);
            bs0 = *reinterpret_cast<const f32x4 *>(&bias_lds[16 * (tile0 + t0) + 4 * kg]);
            if constexpr (t1 < ST) {
                iw1 = *reinterpret_cast<const f32x4 *>(&invw_lds[16 * (tile0 + t1) + 4 * kg]);
                bs1 = *reinterpret_cast<const f32x4 *>(&bias_lds[16 * (tile0 + t1) + 4 * kg]);
            }
            f32x4 a0 = tile_acc(T0), a1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (t1 < ST) a1 = tile_acc(ic<t1 < ST ? t1 : 0>{});
            mfma_drain2(a0, a1);
            const int st = GS * G1 + pstep;
            float *dst = &vbuf[(st % R) * VSTEP + (kg * 4 + pc) * 4];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = fmaf(a0[r] * xin, iw0[r], bs0[r]);
            *reinterpret_cast<f32x4 *>(dst + 64 * (tile0 + t0)) = o;
            if constexpr (t1 < ST) {
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] = fmaf(a1[r] * xin, iw1[r], bs1[r]);
                *reinterpret_cast<f32x4 *>(dst + 64 * (tile0 + t1)) = o;
            }
        };
        // which tiles an interval computes: the leader's intervals 1-3 carry the x split, so they get fewer
        auto project_interval = [&](auto KC, int G1) {
            constexpr int k = decltype(KC)::value;
            constexpr int lo = tile_first(ST, k), hi = tile_first(ST, k + 1);
            static_assert(hi - lo <= 3, "at most three tiles per interval");
            if constexpr (hi - lo == 1) project_tiles(ic<lo>{}, ic<ST>{}, G1);
            if constexpr (hi - lo == 2) project_tiles(ic<lo>{}, ic<lo + 1>{}, G1);
            if constexpr (hi - lo == 3) { project_tiles(ic<lo>{}, ic<lo + 1>{}, G1); project_tiles(ic<lo + 2>{}, ic<ST>{}, G1); }
        };
        load_operands(0);
        static_for<0, 8>([&](auto KC) { project_interval(KC, 0); });
        lds_bar();                                       // vI of group 0 complete

        // interval k = 0..7 of group G (two per step, each opened by the barrier the chain waves open theirs with)
        [[maybe_unused]] unsigned long long sacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        if constexpr (DIAG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        auto interval = [&](auto KC, const int G) {
            constexpr int k = decltype(KC)::value;
            lds_bar<!(ABL & 1)>();
            BSTAMP(8 + k)
            if constexpr (ABL & 8) return;
            if constexpr (k == 0) load_operands(G + 1);
            project_interval(KC, G + 1);
            if (leader) {
                if constexpr (k == 1) split_scale(G + 2);
                if constexpr (k >= 2 && k < 2 + KBLK) split_block(G + 2, k - 2);
                if constexpr (k == 1 + KBLK) load_x(G + 3);
            }
            BSTAMP(k)
        };
        for (int G = 0; G < NG; G++) {
            const int s = GS * G;
            interval(ic<0>{}, G); interval(ic<1>{}, G);
            if (s + 1 < T) { interval(ic<2>{}, G); interval(ic<3>{}, G); }
            if (s + 2 < T) { interval(ic<4>{}, G); interval(ic<5>{}, G); }
            if (s + 3 < T) { interval(ic<6>{}, G); interval(ic<7>{}, G); }
        }
#ifdef SLK_DIAG
        if constexpr (DIAG) {
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 16; i++) slk_dbg_bar16[wave][i] = sacc[i];
        }
#endif
    }
}

// One workgroup per CU: ask for enough dynamic LDS that two cannot share a CU.
template <typename K>
static size_t exclusive_cu_lds_bar(K kernel)
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(kernel)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;                         // 160 KB of LDS per CU
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

template <typename K>
static int bar16_blocks_per_cu(K kernel)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kernel), 256, 0) != hipSuccess) return 0;
    return nb;
}

// share_cu: the caller runs more four-chunk workgroups at a time than there are CUs and the instantiation fits two per CU (<= 256
// registers, <= 80 KB of LDS: the 64-wide ones): no dynamic LDS, so that two of them -- the directions of a birnn -- share a CU
template <int I, int N>
static int launch_bar16(const float *x, long ldx, const float *iW, const float *bias, const float *sW, const float *sW2,
                        float *y, long ldy, int T, int B, int reverse, const int *lens, float *zr_out, hipStream_t s, bool share_cu)
{
#ifdef SLK_DIAG
    if constexpr (I == 96 && N == 96) {
        // diagnostic launches (undocumented bits, tools/bar16_check.py; SLOIKA_AMD_BAR16_DIAG forces one for every launch of a
        // process so that tools/bar16_wg_times.py --pipeline can read the clock the kernel gets inside the whole step)
        static const int forced = getenv("SLOIKA_AMD_BAR16_DIAG") ? atoi(getenv("SLOIKA_AMD_BAR16_DIAG")) : 0;
        const int dv = forced ? forced : reverse >> 1;
#define DIAG_LAUNCH(CODE, STAMPS, ABLV)                                                                                   \
        if (dv == CODE) {                                                                                                 \
            const size_t dyn = SLK_PER_DEVICE(size_t, exclusive_cu_lds_bar(gru_bar16_kernel<I, N, false, STAMPS, ABLV>));                  \
            hipLaunchKernelGGL((gru_bar16_kernel<I, N, false, STAMPS, ABLV>), dim3((B + 3) / 4), dim3(256), dyn, s, x, ldx, \
                               iW, bias, sW, sW2, y, ldy, T, B, reverse & 1, lens, zr_out);                                \
            return slk_launch_status();                                                                                    \
        }
        DIAG_LAUNCH(1, true, 0) DIAG_LAUNCH(2, false, 1) DIAG_LAUNCH(3, false, 2) DIAG_LAUNCH(4, false, 4) DIAG_LAUNCH(5, false, 8)
        DIAG_LAUNCH(6, false, 16) DIAG_LAUNCH(7, false, 9) DIAG_LAUNCH(8, false, 3) DIAG_LAUNCH(9, false, 11) DIAG_LAUNCH(10, false, 31)
        DIAG_LAUNCH(11, false, 32) DIAG_LAUNCH(12, false, 34) DIAG_LAUNCH(13, false, 40) DIAG_LAUNCH(14, false, 42) DIAG_LAUNCH(15, false, 36)
#undef DIAG_LAUNCH
    }
#endif
    const bool shared = share_cu && N <= 64;
    if (shared) {
        // the caller's plan counts on two of these workgroups per CU: ask the runtime whether they fit (a compiler that takes more
        // than 256 registers or 80 KB of LDS for the instantiation would otherwise halve the plan's speed silently)
        const int fit = zr_out ? SLK_PER_DEVICE(int, bar16_blocks_per_cu(gru_bar16_kernel<I, N, true>))
                               : SLK_PER_DEVICE(int, bar16_blocks_per_cu(gru_bar16_kernel<I, N, false>));
        if (fit < 2) return SLK_ERR_UNSUPPORTED;         // -> slk_gru_bar16_f32 takes the eight-chunk plan
    }
    if (zr_out) {
        const size_t dyn = shared ? 0 : SLK_PER_DEVICE(size_t, exclusive_cu_lds_bar(gru_bar16_kernel<I, N, true>));
        hipLaunchKernelGGL((gru_bar16_kernel<I, N, true>), dim3((B + 3) / 4), dim3(256), dyn, s, x, ldx, iW, bias, sW, sW2, y,
                           ldy, T, B, reverse & 1, lens, zr_out);
    } else {
        const size_t dyn = shared ? 0 : SLK_PER_DEVICE(size_t, exclusive_cu_lds_bar(gru_bar16_kernel<I, N, false>));
        hipLaunchKernelGGL((gru_bar16_kernel<I, N, false>), dim3((B + 3) / 4), dim3(256), dyn, s, x, ldx, iW, bias, sW, sW2, y,
                           ldy, T, B, reverse & 1, lens, zr_out);
    }
    return slk_launch_status();
}

extern "C" int slk_gru_bar16d_launch(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias,
                                     float *y, long ldy, int T, int B, int insize, int n, int reverse, const int32_t *lens,
                                     float *zr_out, hipStream_t s);

extern "C" int slk_gru_bar16q_launch(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias,
                                     float *y, long ldy, int T, int B, int insize, int n, int reverse, const int32_t *lens,
                                     float *zr_out, hipStream_t s);

// More workgroups of four chunks than CUs would run one after the other: such batches take the eight-chunk plan of
// gru_bar16d.hip, and the sixteen-chunk plan of gru_bar16q.hip when the eight-chunk workgroups do not fit either
// (SLOIKA_AMD_GRU_DUAL=0 / 1 / 2 forces four / eight / sixteen chunks).  Returns chunks per workgroup divided by four.
static int bar16_auto_plan(int B)
{
    static const int forced = getenv("SLOIKA_AMD_GRU_DUAL") ? atoi(getenv("SLOIKA_AMD_GRU_DUAL")) : -1;
    if (forced >= 0) return forced == 0 ? 1 : (forced == 1 ? 2 : 4);
    const int ncu = SLK_PER_DEVICE(int, ([] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n > 0 ? n : 256;
    }()));
    if ((B + 3) / 4 <= ncu) return 1;
    return (B + 7) / 8 <= ncu ? 2 : 4;
}

// Same contract as slk_gru_fused16_f32 (include/sloika_amd.h); SLK_ERR_UNSUPPORTED when no instantiation covers the request.
extern "C" int slk_gru_bar16_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2,
                                 const float *bias, float *y, long ldy, int T, int B, int insize, int n, int reverse, int act,
                                 int gate_act, const int32_t *lens, float *zr_out, slk_stream_t stream)
{
    if (!x || !iW || !sW || !sW2 || !y || T < 1 || B < 1 || insize < 1 || n < 1 || ldx < insize || ldy < n)
        return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;   // 16-byte DMA pieces
    hipStream_t s = slk_stream(stream);
    const int plan = (reverse >> 8) & 3;                 // include/sloika_amd.h: 0 = by batch size, 1 / 2 / 3 = four / eight / sixteen chunks
    const bool share_cu = (reverse >> 10) & 1;           // ... bit 10: four-chunk workgroups of a 64-wide layer may share a CU
    reverse &= 0xff;
    const int per4 = plan == 1 ? 1 : plan == 2 ? 2 : plan == 3 ? 4 : ((reverse >> 1) == 0 ? bar16_auto_plan(B) : 1);
    if (per4 == 4) {
        const int rc = slk_gru_bar16q_launch(x, ldx, iW, sW, sW2, bias, y, ldy, T, B, insize, n, reverse, lens, zr_out, s);
        if (rc != SLK_ERR_UNSUPPORTED) return rc;
    }
    if (per4 >= 2) {
        const int rc = slk_gru_bar16d_launch(x, ldx, iW, sW, sW2, bias, y, ldy, T, B, insize, n, reverse, lens, zr_out, s);
        if (rc != SLK_ERR_UNSUPPORTED) return rc;
    }
#define BAR16(II, NN)                                                                                                              \
    if (insize == II && n == NN) {                                                                                                 \
        int rc = launch_bar16<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse, lens, zr_out, s, share_cu);                 \
        if (rc == SLK_ERR_UNSUPPORTED && share_cu) {     /* two workgroups per CU do not fit: eight chunks, else one per CU */     \
            rc = slk_gru_bar16d_launch(x, ldx, iW, sW, sW2, bias, y, ldy, T, B, insize, n, reverse, lens, zr_out, s);              \
            if (rc == SLK_ERR_UNSUPPORTED) rc = launch_bar16<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse, lens, zr_out, s, false); \
        }                                                                                                                          \
        return rc;                                                                                                                 \
    }
    BAR16(96, 96) BAR16(64, 64) BAR16(32, 96) BAR16(128, 96) BAR16(64, 96) BAR16(48, 32) BAR16(16, 64)
#undef BAR16
    return SLK_ERR_UNSUPPORTED;
}
