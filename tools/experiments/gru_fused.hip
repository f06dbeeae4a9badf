// gru_fused.hip -- a whole Gru layer (input projection + recurrence, sloika/layers.py:1010-1021) in ONE persistent
// kernel on gfx950.
//
// The recurrence of one 4-chunk tile is a serial chain (state exchange -> gates -> state exchange -> update) that keeps
// the MFMA pipe busy well under half of each step.  The time-parallel input projection x.iW^T + b of the same layer is
// independent work of the same size, so it is computed IN the same workgroup by four extra waves:
//
//   waves 0-3 "rec" : z|r phase, candidate phase (the roles of gru_mfma_kernel in recurrent.hip), raised priority
//   waves 4-7 "proj": vI(sp) = x(sp).iW^T + b for the tile, a few steps ahead, into an LDS ring the rec waves read;
//                     they also stream x in (LDS-DMA) and copy finished state blocks out to h_out
//
// The two groups share each SIMD's matrix pipe (two waves per SIMD).  They are NOT coupled by s_barrier: a hardware
// barrier would make the recurrence wait for whatever the projection wave happens to be doing twice per step
// (measured: ~700 of 3800 cycles per step).  Instead every wave publishes progress counters in LDS and consumers poll
// them; LDS operations of one wave execute in order, so "data write, then counter write" on the producer and "counter
// read, then data read" on the consumer is a release/acquire pair without any fence, and the consumer issues its data
// reads together with the counter read (one LDS round trip per exchange, retried in the rare case the counter was
// not there yet).  vI never exists in HBM; x arrives as an image of 16-byte pieces [k/4][chunk][k%4] that the packed
// MFMA A-operand reads without conflicts.  Exact fp32 (v_mfma_f32_4x4x1_16b_f32).
#include <limits.h>

#include "f16split.h"
#include "lds_flags.h"
#include "mfma4.h"

// Diagnostic: shader-clock cycles and 100 MHz wall ticks spent by workgroup 0 in the last fused launch
// (tools/bench_kernels.py reads it to report the clock the chip actually holds under this kernel).
__device__ unsigned long long slk_dbg_clock[2];
__device__ unsigned long long slk_dbg_stamp[16];
#define STAMP(i)                                                                                   \
    if (DIAG && (variant & 4)) {                                                                             \
        unsigned long long tnow;                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        stamp_acc[i] += tnow - tprev;                                                              \
        tprev = tnow;                                                                              \
    }

#ifdef SLK_DIAG                          /* tools/build_diag_lib.sh */
extern "C" SLK_API int slk_debug_read_stamps(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_stamp), sizeof(unsigned long long) * 16) == hipSuccess ? SLK_OK
                                                                                                                : SLK_ERR_LAUNCH;
}
#endif

#ifdef SLK_DIAG                          /* tools/build_diag_lib.sh */
extern "C" SLK_API int slk_debug_read_clock(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_clock), sizeof(unsigned long long) * 2) == hipSuccess ? SLK_OK
                                                                                                               : SLK_ERR_LAUNCH;
}
#endif

// DIAG: diagnostic instantiation (per-phase s_memtime stamps, optional skipping of the projection MFMAs); the production
// instantiation carries none of those branches -- a taken branch costs a lone wave an instruction refetch.
// SAVE (training, sloika_amd/train.py): the activated gates z | r of every step are also written to zr_out[(t*B + b)][2N]
// straight from the recurrent waves' registers (they have no loads, so the stores never make them wait); the reverse pass
// then needs no gate recompute at all.
template <int I, int N, int ACT, int GACT, bool DIAG, bool SAVE = false>
__global__ void __launch_bounds__(512, 2) gru_fused_kernel(const float *__restrict__ x, long ldx,
                                                           const float *__restrict__ iW, const float *__restrict__ bias,
                                                           const float *__restrict__ sW, const float *__restrict__ sW2,
                                                           float *__restrict__ h_out, long ldh, int T, int B, int reverse,
                                                           int act, int gate_act, int diag, const int *__restrict__ lens,
                                                           float *__restrict__ zr_out)
{
    static_assert(I % 16 == 0 && N % 16 == 0 && N <= 128, "unsupported size for the fused GRU kernel");
    // ---------------- recurrent role constants (as in gru_mfma_kernel) ----------------
    constexpr int NW = N / 4;
    constexpr int SA = (2 * NW <= 16) ? 4 : ((2 * NW <= 32) ? 2 : 1);
    constexpr int SB = (NW <= 16) ? 4 : ((NW <= 32) ? 2 : 1);
    constexpr int LPA = 64 / SA, LPB = 64 / SB;
    constexpr int MA = N / SA, MB = N / SB;
    constexpr int GA = 16 / SA, GB = 16 / SB;
    constexpr int CBA = 4 - ilog2(SA), CBB = 4 - ilog2(SB);
    constexpr int NV = N / 16;
    // ---------------- projection role constants ----------------
    // vI rows in tiles of 16, K in blocks of 32, four time steps x four chunks = the 16 rows of one
    // v_mfma_f32_16x16x32_f16; every float32 operand is split v = hi + lo into two halves and the product evaluated as
    // hi.hi + hi.lo + lo.hi with float32 accumulation (see gemm_rows_f16x3.hip): float32-grade accuracy at a fifth of
    // the matrix-pipe time of the fp32 MFMA, which matters here because the pipe is shared with the recurrence
    constexpr int NT16 = 3 * N / 16;                     // tiles of vI rows
    constexpr int NTW = (NT16 + 3) / 4;                  // tiles per proj wave (tile pw + 4*i; the last may be absent)
    constexpr int KBLK = (I + 31) / 32;
    constexpr int GS = 4;                                // time steps per projection group
    // ---------------- LDS ----------------
    constexpr int KB = 8;                                // steps per x block / per h_out block
    constexpr int R = 2 * GS;                            // vI ring: the projection works one group of steps ahead
    constexpr int XIMG = 4 * I;                          // floats of one step's x image: [k/4][chunk][k%4]
    constexpr int XPIECES = KB * I;                      // 16-byte pieces per x block
    constexpr int NDMA = (XPIECES / 64 + 3) / 4;
    // state history: h(s) lives in slot s % HSLOTS as the packed A-operand image [neuron][chunk]; step s reads slot
    // s-1 and writes slot s, and the proj waves copy finished 8-step blocks to h_out (no separate output staging)
    constexpr int HSLOTS = 3 * KB, HIMG = 4 * N;
    __shared__ __attribute__((aligned(16))) float xbuf[2 * KB * XIMG];
    __shared__ __attribute__((aligned(16))) float vbuf[R * 3 * N * 4];      // vI[slot][row][chunk]
    __shared__ __attribute__((aligned(16))) float hring[HSLOTS * HIMG];
    __shared__ __attribute__((aligned(16))) float rhbuf[N * 4];
    __shared__ float bias_lds[3 * N];
    __shared__ __attribute__((aligned(64))) int flags[16];
    __shared__ __attribute__((aligned(64))) int xflags[16];

    const int variant = DIAG ? (((diag & 1) ? 4 : 0) | ((diag & 2) ? 8 : 0)) : 0;   // 4: s_memtime stamps, 8: skip projection MFMAs
    const unsigned long long clk0 = clock64(), wall0 = wall_clock64();
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    const int blk = lane >> 2, ci = lane & 3;
    const bool is_rec = wave < 4;
    const int cls = (lane & 15) >> 2;                    // which counter group this lane watches when polling

    for (int i = tid; i < HIMG; i += 512) hring[(HSLOTS - 1) * HIMG + i] = 0.0f;      // h(-1) = 0
    for (int i = tid; i < 3 * N; i += 512) bias_lds[i] = bias ? bias[i] : 0.0f;
    if (tid < 16) { flags[tid] = 0; xflags[tid] = 0; }
    __syncthreads();                                     // the only hardware barrier

    if (is_rec) {
        // =================================================================================================
        // recurrent waves
        // =================================================================================================
        constexpr bool OWN_FIRST = (SA == 1) && (NW % 4 == 0);
        constexpr int NVO = (NW + 15) / 16;                  // operand registers that hold own-neuron state
        const int rot = OWN_FIRST ? NW * wave : 0;
        const int la = lane % LPA, ga = lane / LPA;
        const bool validA = la < 2 * NW;
        const bool isR = la >= NW;
        const int neuronA = wave * NW + (validA ? (la % NW) : 0);
        const int rowA = isR ? N + neuronA : neuronA;
        const int lb = lane % LPB, gb = lane / LPB;
        const bool validB = lb < NW;
        const int neuronB = wave * NW + (validB ? lb : 0);
        const bool zlane = lane < NW;
        const bool rlane = lane >= NW && lane < 2 * NW;
        float wA[MA], wB[MB];
        {
            // OWN_FIRST (one K-slice): this wave's phase-A chain runs over k in the rotated order rot, rot+1, ... so that
            // its first NW products use the wave's own neurons, whose state needs no handshake
            const float *pa = sW + (size_t)rowA * N + ga * MA;
#pragma unroll
            for (int m = 0; m < MA; m++) {
                const int k = OWN_FIRST ? (m + rot) % N : m;
                wA[m] = validA ? pa[k] : 0.0f;
            }
            const float *pb = sW2 + (size_t)neuronB * N + gb * MB;
#pragma unroll
            for (int m = 0; m < MB; m++) wB[m] = validB ? pb[m] : 0.0f;
        }
        const int addrA0 = 4 * ((blk / GA) * MA + (blk % GA)) + ci;
        int addrA[NV];                                       // operand register v, this lane's block: k = 16*v + blk (+ rot)
#pragma unroll
        for (int v = 0; v < NV; v++) addrA[v] = OWN_FIRST ? 4 * ((16 * v + blk + rot) % N) + ci : addrA0 + 4 * v * GA;
        const int addrB0 = 4 * ((blk / GB) * MB + (blk % GB)) + ci;
        const float mask_zr = (validA && ga == 0) ? 1.0f : 0.0f;
        const float mask_c = zlane ? 1.0f : 0.0f;
        __builtin_amdgcn_s_setprio(3);      // the serial chain goes first; projection MFMAs fill its gaps
        // what this lane's watched counter must reach, as step + offset (branch-free in the loop); unwatched counters
        // get a huge negative offset.  Phase A: fB >= s.  Phase B: fA >= s+1.  Prefetch of vI(s+1): vready >= s+2 and
        // the state slot of step s+1 copied out (flushed >= (s+1)/KB - 2).
        constexpr int NOWATCH = INT_MIN / 2;
        const int offA = cls == 1 ? 0 : NOWATCH;
        const int offB = cls == 0 ? 1 : NOWATCH;
        const bool watch_flush = cls == 3;
        const int offV = cls == 2 ? 2 : NOWATCH;

        unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        if (DIAG && (variant & 4)) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }

        f32x4 a0, c0;                                        // vI(s) rows of this lane's outputs, read one step ahead
        for (;;) {                                           // vI(0)
            const int f = poll_issue(flags, lane);
            a0 = *reinterpret_cast<const f32x4 *>(&vbuf[4 * rowA]);
            c0 = *reinterpret_cast<const f32x4 *>(&vbuf[4 * (2 * N + neuronB)]);
            const bool ok = poll_result(f, cls == 2 ? 1 : NOWATCH);
            keep(a0); keep(c0);
            if (ok) break;
        }

        for (int s = 0; s < T; s++) {
            const float *hprev = hring + ((s + HSLOTS - 1) % HSLOTS) * HIMG;
            float *hcur = hring + (s % HSLOTS) * HIMG;
            const int needA = s + offA, needB = s + offB;
            const int needV = watch_flush ? (s + 1) / KB - 2 : s + offV;
            STAMP(0)

            // ---------------- phase A: z | r ----------------
            // this wave's own neurons need no handshake (LDS keeps a wave's operations in order)
            const f32x4 hown = *reinterpret_cast<const f32x4 *>(&hprev[4 * neuronA]);
            float hp[NV];
            f32x4 accA[4] = {a0 * mask_zr, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if constexpr (OWN_FIRST) {
                // own quarter first: its operands are this wave's own (ordered) writes; the other waves' state is
                // requested at the same time and checked after the first NW MFMAs, which hide the round trip
                float ho[NVO];
#pragma unroll
                for (int v = 0; v < NVO; v++) ho[v] = hprev[addrA[v]];
                // the request for the other waves' state goes out as late as its round trip allows (~16 MFMAs): the
                // later the counters are sampled, the likelier every wave has published
                constexpr int LATE = 0;           // (sampling the counters later, after NW-16 MFMAs, measured no better)
                __builtin_amdgcn_sched_barrier(0);
                mfma_chain_range<CBA, GA, 0>(ho, wA, accA, std::make_integer_sequence<int, LATE>{});
                __builtin_amdgcn_sched_barrier(0);
                int f = poll_issue(flags, lane);
#pragma unroll
                for (int v = NW / 16; v < NV; v++) hp[v] = hprev[addrA[v]];
                __builtin_amdgcn_sched_barrier(0);
                mfma_chain_range<CBA, GA, LATE>(ho, wA, accA, std::make_integer_sequence<int, NW - LATE>{});
                __builtin_amdgcn_sched_barrier(0);
                bool ok = poll_result(f, needA);
#pragma unroll
                for (int v = NW / 16; v < NV; v++) keep(hp[v]);
                while (!ok) {
                    f = poll_issue(flags, lane);
#pragma unroll
                    for (int v = NW / 16; v < NV; v++) hp[v] = hprev[addrA[v]];
                    ok = poll_result(f, needA);
#pragma unroll
                    for (int v = NW / 16; v < NV; v++) keep(hp[v]);
                }
                STAMP(1)
                mfma_chain_range<CBA, GA, NW>(hp, wA, accA, std::make_integer_sequence<int, MA - NW>{});
            } else {
                for (;;) {
                    const int f = poll_issue(flags, lane);
#pragma unroll
                    for (int v = 0; v < NV; v++) hp[v] = hprev[addrA[v]];
                    const bool ok = poll_result(f, needA);
#pragma unroll
                    for (int v = 0; v < NV; v++) keep(hp[v]);
                    if (ok) break;
                }
                STAMP(1)
                mfma_chain<CBA, GA>(hp, wA, accA, std::make_integer_sequence<int, MA>{});
            }
            f32x4 g = sum_slices<SA>((accA[0] + accA[1]) + (accA[2] + accA[3]));
            STAMP(2)
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] = act_sel<GACT>(gate_act, g[i]);
            if constexpr (SAVE) {
                if (validA && ga == 0) {                     // one lane per gate row; its four values are the four chunks
                    const size_t trow = (size_t)(reverse ? T - 1 - s : s) * B + b0;
#pragma unroll
                    for (int c4 = 0; c4 < 4; c4++)
                        if (b0 + c4 < B) zr_out[(trow + c4) * (2 * N) + rowA] = g[c4];
                }
            }
            if (rlane) *reinterpret_cast<f32x4 *>(&rhbuf[4 * neuronA]) = g * hown;
            publish(flags, wave, s + 1, lane);
            STAMP(3)

            // ---------------- phase B: candidate ----------------
            float rp[NV];
            for (;;) {
                const int f = poll_issue(flags, lane);
#pragma unroll
                for (int v = 0; v < NV; v++) rp[v] = rhbuf[addrB0 + 4 * v * GB];
                const bool ok = poll_result(f, needB);
#pragma unroll
                for (int v = 0; v < NV; v++) keep(rp[v]);
                if (ok) break;
            }
            STAMP(4)
            // vI(s+1), one step ahead (its latency disappears behind the candidate chain)
            const float *vnext = vbuf + ((s + 1) % R) * (3 * N * 4);
            const bool more = s + 1 < T;
            int fv = 0;
            f32x4 a0n = a0, c0n = c0;
            if (more) {
                fv = poll_issue(flags, lane);
                a0n = *reinterpret_cast<const f32x4 *>(&vnext[4 * rowA]);
                c0n = *reinterpret_cast<const f32x4 *>(&vnext[4 * (2 * N + neuronB)]);
            }
            f32x4 accB[4] = {c0 * mask_c, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain<CBB, GB>(rp, wB, accB, std::make_integer_sequence<int, MB>{});
            f32x4 cc = sum_slices<SB>((accB[0] + accB[1]) + (accB[2] + accB[3]));
            if (more) {                                      // long since answered; checked before the writes below queue up
                bool ok = poll_result(fv, needV);
                keep(a0n); keep(c0n);
                while (!ok) {
                    const int f2 = poll_issue(flags, lane);
                    a0n = *reinterpret_cast<const f32x4 *>(&vnext[4 * rowA]);
                    c0n = *reinterpret_cast<const f32x4 *>(&vnext[4 * (2 * N + neuronB)]);
                    ok = poll_result(f2, needV);
                    keep(a0n); keep(c0n);
                }
            }
            STAMP(5)
            if (zlane) {
                f32x4 hn;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float hbar = act_sel<ACT>(act, cc[i]);
                    hn[i] = g[i] * hown[i] + (1.0f - g[i]) * hbar;      // layers.py:1020
                }
                *reinterpret_cast<f32x4 *>(&hcur[4 * neuronB]) = hn;
            }
            publish(flags, 4 + wave, s + 1, lane);
            a0 = a0n;
            c0 = c0n;
            STAMP(6)
        }
        if (DIAG && (variant & 4) && blockIdx.x == 0 && tid == 0)
            for (int i = 0; i < 9; i++) slk_dbg_stamp[i] = stamp_acc[i];
    } else {
        // =================================================================================================
        // projection waves
        // =================================================================================================
        const int pw = wave - 4;
        const int col = lane & 15, kq = lane >> 4;
        constexpr bool LAST_MAYBE = (NT16 % 4) != 0;        // the last tile slot exists only for some waves
        const bool last_ok = pw + 4 * (NTW - 1) < NT16;
        // B operands: lane holds vI row 16*t + col, k = 32*kb + 8*kq + 0..7, as fp16 hi and lo parts
        // (every row scaled by a power of two to a maximum in [1, 2) before the split, as in gru_fused16.hip: any finite float32
        // weight is then inside fp16 range; the lane that holds a row is the lane that stores it and undoes the scale)
        half8 whi[NTW][KBLK], wlo[NTW][KBLK];
        float inv_w[NTW];
#pragma unroll
        for (int i = 0; i < NTW; i++) {
            const bool ok = (i < NTW - 1) || !LAST_MAYBE || last_ok;
            const int row = ok ? 16 * (pw + 4 * i) + col : 0;
            float u[KBLK][8];
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kq;
                const bool kok = ok && (I % 32 == 0 || k0 < I);
                const float *src = iW + (size_t)row * I + (kok ? k0 : 0);
                const float4 u0 = *reinterpret_cast<const float4 *>(src), u1 = *reinterpret_cast<const float4 *>(src + 4);
                const float t[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    u[kb][j] = kok ? t[j] : 0.0f;
                    m = fmaxf(m, fabsf(u[kb][j]));
                }
            }
            const float ws = pow2_scale(kgroup_max(m), inv_w[i]);
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float v = u[kb][j] * ws;
                    const _Float16 h = (_Float16)v;
                    whi[i][kb][j] = h;
                    wlo[i][kb][j] = (_Float16)(v - (float)h);
                }
            }
        }
        // A operands: lane supplies row m = lane & 15 = (step in group, chunk) and k = 32*kb + 8*kq + 0..7 from the
        // step's x image, where element x[chunk][k] sits at 16*(k>>2) + 4*chunk + (k&3)
        const int a_step = col >> 2, a_chunk = col & 3;
        auto dma_block = [&](int s0, int slot) {
#pragma unroll
            for (int j = 0; j < NDMA; j++) {
                const int piece0 = (j * 4 + pw) * 64;
                if (piece0 < XPIECES) {
                    const int p = piece0 + lane;
                    const int kk = p / I, pp = p % I, q = pp >> 2, c = pp & 3;
                    const int bc = min(b0 + c, B - 1);
                    // ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step, and the
                    // steps past the end re-read the last valid row (their results are never stored)
                    const int Tc = lens ? min(max(lens[bc], 1), T) : T;
                    const int ss = min(s0 + kk, Tc - 1);
                    const int tt = reverse ? Tc - 1 - ss : ss;
                    const float *src = x + ((size_t)tt * B + bc) * ldx + 4 * q;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)&xbuf[slot * (KB * XIMG) + piece0 * 4],
                                                     16, 0, 0);
                }
            }
        };
        // h_out: part j of a finished 8-step block = 256 float4 (step kk, chunk c, neurons 4*f4..4*f4+3), gathered from
        // the [neuron][chunk] images with four ds_read_b32 (4-way bank conflicts, off the critical path) and stored as
        // 4 x 256-byte runs per wave
        const bool vec_store = (ldh % 4 == 0) && ((reinterpret_cast<uintptr_t>(h_out) & 15) == 0);
        constexpr int OF4 = KB * N;                         // float4s per block
        constexpr int NFL = (OF4 + 255) / 256;              // parts per block
        auto flush_part = [&](int kb, int j) {
            const int idx = (tid - 256) + 256 * j;
            const int c = idx & 3, rest = idx >> 2, f4 = rest % (N / 4), kk = rest / (N / 4);
            const int ss = kb * KB + kk;
            const int Tc = (lens && b0 + c < B) ? min(max(lens[b0 + c], 1), T) : T;
            if (idx < OF4 && ss < Tc && b0 + c < B) {
                const float *src = hring + (ss % HSLOTS) * HIMG + 16 * f4 + c;
                const float v0 = src[0], v1 = src[4], v2 = src[8], v3 = src[12];
                const int tt = reverse ? Tc - 1 - ss : ss;
                float *dst = h_out + ((size_t)tt * B + b0 + c) * ldh + 4 * f4;
                if (vec_store) *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                else { dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; }
            }
        };
        auto wait_flags = [&](int group, int value) {       // every counter of `group` (0 fA, 1 fB) >= value
            const int need = cls == group ? value : INT_MIN;
            while (!reached(flags, lane, need)) __builtin_amdgcn_s_sleep(1);
        };
        auto wait_xflags = [&](int group, int value) {      // group 0 xready, 1 xdone (lanes watch xflags[l & 15], 8..15 stay 0)
            const int need = cls == group ? value : INT_MIN;
            while (!reached(xflags, lane, need)) __builtin_amdgcn_s_sleep(1);
        };

        dma_block(0, 0);
        const int NG = (T + GS - 1) / GS;
        for (int q = 0; q < NG; q++) {
            if ((q & 1) == 0) {                                         // KB = 2 groups: a new x block starts here
                const int xb = q / 2;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my share of block xb (issued a block ago) landed
                publish(xflags, pw, xb + 1, lane);
                publish(xflags, 4 + pw, xb, lane);                      // and I am done reading block xb-1
                if ((xb + 1) * KB < T) {
                    wait_xflags(1, xb);                                 // slot (xb+1)&1 held block xb-1: everyone past it
                    dma_block((xb + 1) * KB, (xb + 1) & 1);
                }
                wait_xflags(0, xb + 1);
            } else if (q >= 3) {
                // copy a finished block of states out (the recurrence does not need this wave for a while: vI is
                // published through step 4q-1 and block fkb ends at step 4q-5)
                const int fkb = (q - 3) / 2;
                wait_flags(1, (fkb + 1) * KB);
                for (int j = 0; j < NFL; j++) flush_part(fkb, j);
                publish(flags, 12 + pw, fkb + 1, lane);
            }
            // ---- per K block: the group's A operands (x split into halves on the fly), then every tile's three MFMAs ----
            f32x4 acc[NTW];
#pragma unroll
            for (int i = 0; i < NTW; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float *img = xbuf + ((q >> 1) & 1) * (KB * XIMG) + (GS * (q & 1) + a_step) * XIMG + 4 * a_chunk;
            // each row of x is scaled by a power of two so that its largest |x| lies in [1, 2) (exact, undone on the
            // accumulators): the first layer's input is an unbounded elu convolution output
            float amax = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kq;
                const bool kok = (I % 32 == 0) || k0 < I;
                const float *src = img + 4 * (kok ? k0 : 0);
                const f32x4 u0 = *reinterpret_cast<const f32x4 *>(src), u1 = *reinterpret_cast<const f32x4 *>(src + 16);
#pragma unroll
                for (int j = 0; j < 4; j++) amax = fmaxf(amax, kok ? fmaxf(fabsf(u0[j]), fabsf(u1[j])) : 0.0f);
            }
            float xinv;
            const float xs = pow2_scale(kgroup_max(amax), xinv);
            // the accumulator rows of this lane are (step kq, chunk 0..3): their inverse scales sit in lanes 4*kq + (0..3)
            f32x4 inv;
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) inv[r4] = __shfl(xinv, 4 * kq + r4);
#pragma unroll
            for (int kb = 0; kb < KBLK; kb++) {
                const int k0 = 32 * kb + 8 * kq;
                const bool kok = (I % 32 == 0) || k0 < I;
                const float *src = img + 4 * (kok ? k0 : 0);           // 16 * (k0 / 4)
                const f32x4 u0 = *reinterpret_cast<const f32x4 *>(src), u1 = *reinterpret_cast<const f32x4 *>(src + 16);
                half8 ahi, alo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float v = kok ? (j < 4 ? u0[j & 3] : u1[j & 3]) * xs : 0.0f;
                    const _Float16 h = (_Float16)v;
                    ahi[j] = h;
                    alo[j] = (_Float16)(v - (float)h);
                }
                if (!(DIAG && (variant & 8))) {
#pragma unroll
                    for (int i = 0; i < NTW; i++) {
                        if (i < NTW - 1 || !LAST_MAYBE || last_ok) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, wlo[i][kb], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, whi[i][kb], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, whi[i][kb], acc[i], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NTW; i++) {
                const float tb = bias_lds[(i < NTW - 1 || !LAST_MAYBE || last_ok) ? 16 * (pw + 4 * i) + col : 0];
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) acc[i][r4] = fmaf(acc[i][r4] * inv[r4], inv_w[i], tb);
            }
            // the group's ring slots were last read by phase A of steps GS*q - R ... GS*q + GS-1 - R
            if (GS * q + GS > R) wait_flags(0, GS * q + GS - R);
            // D: lane holds the four chunks of (step GS*q + kq, vI row 16*t + col) = one 16-byte entry of vbuf
            const int st = GS * q + kq;
            if (st < T) {
                float *vdst = vbuf + (st % R) * (3 * N * 4) + 4 * col;
#pragma unroll
                for (int i = 0; i < NTW; i++)
                    if (i < NTW - 1 || !LAST_MAYBE || last_ok) *reinterpret_cast<f32x4 *>(&vdst[64 * (pw + 4 * i)]) = acc[i];
            }
            publish(flags, 8 + pw, GS * q + GS, lane);
        }
        // blocks of states the loop did not copy out
        wait_flags(1, T);
        const int kbl = (T - 1) / KB;
        for (int kb = 0; kb <= kbl; kb++)
            if (2 * kb + 3 >= NG)
                for (int j = 0; j < NFL; j++) flush_part(kb, j);
    }
    if (DIAG && blockIdx.x == 0 && tid == 0) {
        slk_dbg_clock[0] = clock64() - clk0;
        slk_dbg_clock[1] = wall_clock64() - wall0;
    }
}

// One workgroup per CU: the kernel is a latency-bound serial chain, and two of these workgroups on one CU (possible for
// the small shapes, e.g. when the two directions of a birnn run on separate streams) slow each other down while other
// CUs idle.  Asking for enough dynamic LDS that two cannot share a CU makes the dispatcher spread them.
template <typename K>
static size_t exclusive_cu_lds(K kernel)
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

template <int I, int N>
static int launch_fused(const float *x, long ldx, const float *iW, const float *bias, const float *sW, const float *sW2,
                        float *y, long ldy, int T, int B, int reverse, const int *lens, hipStream_t s)
{
    const int diag = (reverse >> 1) & 3;
    if constexpr (I == 96 && N == 96) {
        if (diag) {
            hipLaunchKernelGGL((gru_fused_kernel<I, N, SLK_ACT_TANH, SLK_ACT_SIGMOID, true>), dim3((B + 3) / 4), dim3(512), 0, s,
                               x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse & 1, SLK_ACT_TANH, SLK_ACT_SIGMOID, diag, lens,
                               (float *)nullptr);
            return slk_launch_status();
        }
    }
    const size_t dyn_lds = SLK_PER_DEVICE(size_t, exclusive_cu_lds(gru_fused_kernel<I, N, SLK_ACT_TANH, SLK_ACT_SIGMOID, false>));
    hipLaunchKernelGGL((gru_fused_kernel<I, N, SLK_ACT_TANH, SLK_ACT_SIGMOID, false>), dim3((B + 3) / 4), dim3(512), dyn_lds, s,
                       x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse & 1, SLK_ACT_TANH, SLK_ACT_SIGMOID, 0, lens,
                       (float *)nullptr);
    return slk_launch_status();
}

template <int I, int N>
static int launch_fused_train(const float *x, long ldx, const float *iW, const float *bias, const float *sW, const float *sW2,
                              float *y, long ldy, float *zr_out, int T, int B, int reverse, hipStream_t s)
{
    const size_t dyn_lds = SLK_PER_DEVICE(size_t, exclusive_cu_lds(gru_fused_kernel<I, N, SLK_ACT_TANH, SLK_ACT_SIGMOID, false, true>));
    hipLaunchKernelGGL((gru_fused_kernel<I, N, SLK_ACT_TANH, SLK_ACT_SIGMOID, false, true>), dim3((B + 3) / 4), dim3(512),
                       dyn_lds, s, x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse & 1, SLK_ACT_TANH, SLK_ACT_SIGMOID, 0,
                       (const int *)nullptr, zr_out);
    return slk_launch_status();
}

// (bits 1-2 of `reverse` request a diagnostic launch (96 -> 96 only) that fills the s_memtime stamps read by slk_debug_read_stamps;
//  undocumented in the public header on purpose -- tools/bench_kernels.py uses it.)
// Returns SLK_ERR_UNSUPPORTED when no fused instantiation covers the request (the caller then uses
// projection GEMM + gru_mfma_kernel).
static int gru_fused_entry(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias,
                           float *y, long ldy, int T, int B, int insize, int n, int reverse, int act, int gate_act,
                           const int32_t *lens, slk_stream_t stream)
{
    if (!x || !iW || !sW || !sW2 || !y || T < 1 || B < 1 || insize < 1 || n < 1 || ldx < insize || ldy < n)
        return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;   // 16-byte DMA pieces
    hipStream_t s = slk_stream(stream);
#define FUSED(II, NN) \
    if (insize == II && n == NN) return launch_fused<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse, lens, s);
    // only shapes whose two roles fit 256 VGPRs without spilling are instantiated (e.g. 128->112 / 144->112 spill
    // > 1 KB per lane and run far slower than the two-kernel path)
    // (128 -> 96 left this list when the projection gained its row scaling: four K blocks no longer fit 256 registers without
    // spills; that shape takes the projection GEMM + recurrence kernel in this arithmetic, and gru_bar16 by default)
    FUSED(96, 96) FUSED(64, 64) FUSED(32, 96) FUSED(16, 16) FUSED(48, 32) FUSED(64, 96) FUSED(16, 64)
#undef FUSED
    return SLK_ERR_UNSUPPORTED;
}

extern "C" int slk_gru_fused_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2,
                                 const float *bias, float *y, long ldy, int T, int B, int insize, int n, int reverse,
                                 int act, int gate_act, slk_stream_t stream)
{
    return gru_fused_entry(x, ldx, iW, sW, sW2, bias, y, ldy, T, B, insize, n, reverse, act, gate_act, nullptr, stream);
}

// Ragged batch (reads of different lengths padded to T): lens[b] in [1, T] = valid steps of chunk b.  Rows t >= lens[b] of
// y are left untouched; with reverse = 1 the scan of chunk b starts at ITS last step (Reverse(Gru) on the unpadded read).
extern "C" int slk_gru_fused_ragged_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2,
                                        const float *bias, float *y, long ldy, int T, int B, int insize, int n, int reverse,
                                        int act, int gate_act, const int32_t *lens, slk_stream_t stream)
{
    if (!lens) return SLK_ERR_INVALID_ARG;
    return gru_fused_entry(x, ldx, iW, sW, sW2, bias, y, ldy, T, B, insize, n, reverse & 1, act, gate_act, lens, stream);
}

// Forward pass of a training step: slk_gru_fused_f32 that also leaves the activated gates of every step in
// zr_out[(t*B + b)][2n] = [z | r] (what the reverse scan slk_gru_backward_f32 consumes).  Same shapes as slk_gru_fused_f32.
extern "C" int slk_gru_fused_train_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2,
                                       const float *bias, float *y, long ldy, float *zr_out, int T, int B, int insize, int n,
                                       int reverse, int act, int gate_act, slk_stream_t stream)
{
    if (!x || !iW || !sW || !sW2 || !y || !zr_out || T < 1 || B < 1 || insize < 1 || n < 1 || ldx < insize || ldy < n)
        return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;
    hipStream_t s = slk_stream(stream);
#define FUSED(II, NN) \
    if (insize == II && n == NN) return launch_fused_train<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, zr_out, T, B, reverse, s);
    // (128 -> 96 left this list when the projection gained its row scaling: four K blocks no longer fit 256 registers without
    // spills; that shape takes the projection GEMM + recurrence kernel in this arithmetic, and gru_bar16 by default)
    FUSED(96, 96) FUSED(64, 64) FUSED(32, 96) FUSED(16, 16) FUSED(48, 32) FUSED(64, 96) FUSED(16, 64)
#undef FUSED
    return SLK_ERR_UNSUPPORTED;
}
