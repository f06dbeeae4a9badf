// gru_fused16.hip -- a whole Gru layer (sloika/layers.py:1010-1021) in ONE persistent kernel, with BOTH halves -- the
// time-parallel input projection and the recurrence -- on the fp16 matrix pipe at float32-grade accuracy.
//
// gru_fused.hip keeps the recurrent products in exact fp32 (v_mfma_f32_4x4x1, 8 cycles per 64 outputs x 4 chunks x 1 k):
// 144 MFMAs = 1150 pipe cycles per step on a chain that is strictly serial, plus K-slice sums and 4-wide gate maths on
// 48 of 64 lanes.  Here every float32 operand of the recurrence is split  v = hi + lo  (two fp16 halves, 22 significand
// bits) and a product is  w_hi.h_lo + w_lo.h_hi + w_hi.h_hi  in float32 accumulators (v_mfma_f32_16x16x32_f16, 16 cycles
// per 16 neurons x 16 columns x 32 k): a tile of 16 neurons over all of K = 96 costs 9 MFMAs instead of 96 x 4x4x1.
//
// Layout that makes the 16x16 tile dense although a workgroup only has 4 chunks (B = 1024 chunks over 256 CUs):
//   * A = weights (row i = neuron 16*tile + i), B = state, and the state of chunk c is supplied in ALL FOUR column groups
//     (column 4q + c, q = 0..3).  Lane (q, c, g = lane>>4) then receives D[4g + r][4q + c], r = 0..3 -- the same four neurons
//     four times over -- and keeps r = q: every lane owns ONE (neuron 16*tile + 4g + q, chunk c) pair, 64 lanes = 16 neurons x
//     4 chunks, so gate arithmetic is one value per lane and tile, on all 64 lanes.
//   * chain wave w owns neurons 32w .. 32w+31 (tiles 2w, 2w+1) for z, r AND the candidate: h, z, r of a (neuron, chunk) pair
//     live in one lane's registers; only the MFMA operands travel.
//   * the order of k inside a 32-wide K block is free (A and B only have to agree): k-block w is exactly what wave w
//     produces, laid out so that a lane's two values (tiles 2w, 2w+1) are adjacent halves -> ONE ds_write_b32 for the hi
//     image and one for the lo image per exchange; readers fetch their B operand as one ds_read_b128 per K block and image.
//
// Waves: 0 .. N/32-1 chain (raised priority), 4-7 projection / x DMA / h_out copies exactly as in gru_fused.hip, coupled by
// LDS progress counters (lds_flags.h), no s_barrier after start-up.
// Accuracy: hi + lo carries 22 bits, the dropped lo.lo term is < 2^-22 relative.  |h| <= 1 and |r.h| <= 1 are inside fp16
// range by construction; every row of x (unbounded: the first layer reads an elu convolution) and every row of the three
// weight matrices is scaled by a power of two to a maximum in [1, 2) before its split and the accumulators are scaled
// back (exact), so any finite float32 input or weight is handled at float32-grade accuracy relative to its row maximum.
// Callers that need plain fp32 arithmetic use gru_fused.hip / the two-kernel path.
#include <limits.h>

#include "f16split.h"
#include "lds_flags.h"

// Diagnostic instantiation: shader-clock cycles workgroup 0's chain wave 0 spends between the marks of one step, summed over
// the scan (tools/bench_kernels.py --what gruf16 reads them).  The production instantiation carries none of this.
__device__ unsigned long long slk_dbg_stamp16[16];
__device__ unsigned long long slk_dbg_pstamp16[8][8];          // projection waves of workgroup 0: cycles per section
#ifdef SLK_DIAG                          /* tools/build_diag_lib.sh */
extern "C" SLK_API int slk_debug_read_pstamps16(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_pstamp16), sizeof(unsigned long long) * 64) == hipSuccess ? SLK_OK
                                                                                                                   : SLK_ERR_LAUNCH;
}
#endif
#define PSTAMP(i)                                                                     \
    if constexpr (DIAG) {                                                             \
        unsigned long long tnow;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        pacc[i] += tnow - ptprev;                                                     \
        ptprev = tnow;                                                                \
    }
#ifdef SLK_DIAG                          /* tools/build_diag_lib.sh */
extern "C" SLK_API int slk_debug_read_stamps16(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(slk_dbg_stamp16), sizeof(unsigned long long) * 16) == hipSuccess ? SLK_OK
                                                                                                                  : SLK_ERR_LAUNCH;
}
#endif
#define STAMP16(i)                                                                    \
    if constexpr (DIAG) {                                                             \
        unsigned long long tnow;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        stamp_acc[i] += tnow - tprev;                                                 \
        tprev = tnow;                                                                 \
    }

// SAVE (training, sloika_amd/train.py): the activated gates z | r of every step are also written to zr_out[(t*B + b)][2N].
// ABL (diagnostic builds only): 1 = no MFMAs in the chain waves, 2 = chain polls never wait, 4 = cheap activations,
// 8 = projection waves idle (vI ring holds whatever was there) -- timing experiments, results are garbage.
template <int I, int N, bool SAVE, bool DIAG = false, int ABL = 0>
__global__ void __launch_bounds__(512, 2) gru_fused16_kernel(const float *__restrict__ x, long ldx,
                                                             const float *__restrict__ iW, const float *__restrict__ bias,
                                                             const float *__restrict__ sW, const float *__restrict__ sW2,
                                                             float *__restrict__ h_out, long ldh, int T, int B, int reverse,
                                                             const int *__restrict__ lens, float *__restrict__ zr_out)
{
    static_assert(I % 16 == 0 && N % 32 == 0 && N <= 96, "unsupported size for the fp16-split fused GRU kernel");
    constexpr int NCW = N / 32;                          // chain waves = 32-wide K blocks of the recurrent products
    constexpr int KBS = N / 32;
    // ---------------- projection role constants (as gru_fused.hip) ----------------
    constexpr int NT16 = 3 * N / 16;                     // tiles of vI rows
    constexpr int KBLK = (I + 31) / 32;
    constexpr int GS = 4;                                // time steps per projection group
    // ---------------- LDS ----------------
    constexpr int KB = 8;                                // steps per x block / per h_out block
    constexpr int R = 2 * GS;                            // vI ring: the projection works one group of steps ahead
    constexpr int XIMG = 4 * I;                          // floats of one step's x image: [k/4][chunk][k%4]
    constexpr int XPIECES = KB * I;                      // 16-byte pieces per x block
    constexpr int XSLOTS = 4;                            // x ring: blocks are requested XSLOTS-1 blocks (24 steps) ahead
    constexpr int NPER = (XPIECES / 64 + 1) / 2;         // 1 KiB DMA requests per block and DMA wave (service waves 0, 1)
    constexpr int HSLOTS = 3 * KB, HIMG = 4 * N;         // float32 state history [slot][neuron][chunk] = h_out staging
    __shared__ __attribute__((aligned(16))) float xbuf[XSLOTS * KB * XIMG];
    __shared__ __attribute__((aligned(16))) float vbuf[R * 3 * N * 4];      // vI[slot][row][chunk]
    __shared__ __attribute__((aligned(16))) float hring[HSLOTS * HIMG];
    // MFMA B-operand images of h and r*h: [k block][k group 4][chunk 4][8 halves], hi and lo parts
    __shared__ __attribute__((aligned(16))) unsigned h_hi[2 * N], h_lo[2 * N], rh_hi[2 * N], rh_lo[2 * N];
    __shared__ float bias_lds[3 * N];
    // progress counters, in groups of four (lane l polls counter l & 31; its group is `cls`):
    //   flags : 0 fA (r*h of step s published -> s+1)   1 fB (h of step s -> s+1)   2 vready of the service waves 4-7
    //           3 flushed (h_out blocks copied)          4 vready of the extra projection waves NCW..3
    //   xflags: 0 xready   1 xdone of the service waves   2 xdone of the extra projection waves
    // counters of waves that do not exist are preset to INT_MAX and never hold anyone up
    __shared__ __attribute__((aligned(128))) int flags[32];
    __shared__ __attribute__((aligned(128))) int xflags[32];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    const int cls = (lane & 31) >> 2;                    // which counter group this lane watches when polling
    // ---------------- who computes which tiles of vI ----------------
    // Chain waves 0..NCW-1 sit on NCW of the four SIMDs (a workgroup's waves go round the SIMDs, so wave i and wave i+4
    // share one).  The projection's MFMAs go mostly to the waves on the REMAINING SIMDs ("free": waves NCW..3 and
    // 4+NCW..7); the waves that share a SIMD with a chain wave take what is left.  Measured with the projection spread
    // evenly over waves 4-7, the chain spent a sixth of every step waiting for vI.
    constexpr int NFREE = 2 * (4 - NCW);
    constexpr int TILE_CAP = (N == 96) ? 4 : (N == 64 ? 3 : 1);      // register budget of one wave
    constexpr int FREE_EACH = (NT16 + NFREE - 1) / NFREE < TILE_CAP ? (NT16 + NFREE - 1) / NFREE : TILE_CAP;
    constexpr int REM = NT16 - NFREE * FREE_EACH > 0 ? NT16 - NFREE * FREE_EACH : 0;
    constexpr int SH_BASE = REM / NCW, SH_EXTRA = REM % NCW;
    constexpr int MAXT = FREE_EACH > SH_BASE + (SH_EXTRA ? 1 : 0) ? FREE_EACH : SH_BASE + (SH_EXTRA ? 1 : 0);
    static_assert(NFREE * FREE_EACH + REM >= NT16, "tile assignment");

    for (int i = tid; i < 2 * N; i += 512) { h_hi[i] = 0u; h_lo[i] = 0u; }             // h(-1) = 0
    for (int i = tid; i < 3 * N; i += 512) bias_lds[i] = bias ? bias[i] : 0.0f;
    if (tid < 32) {
        const int idx = tid & 3, grp = tid >> 2;
        const bool extra_missing = idx >= 4 - NCW;          // extra projection wave idx = wave NCW + idx
        // service waves 0, 1 run the x DMA (xready), 2, 3 copy states out (flushed)
        flags[tid] = ((grp < 2 && idx >= NCW) || (grp == 4 && extra_missing) || (grp == 3 && idx < 2)) ? INT_MAX : 0;
        xflags[tid] = ((grp == 2 && extra_missing) || (grp == 0 && idx >= 2)) ? INT_MAX : 0;
    }
    __syncthreads();                                     // the only hardware barrier

    if (wave < NCW) {
        // =================================================================================================
        // chain waves
        // =================================================================================================
        const int w = wave;
        const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
        // A operands: lane supplies row i = lane & 15 of a tile and, for k group g, elements j = 0..7 of a K block, where
        // element (g, j) of block kb is neuron 32*kb + 16*(j&1) + 4*g + (j>>1) -- the order the owners' packed writes create.
        // K blocks are visited in the rotated order w, w+1, ...: the wave's own block first (no handshake needed).
        // Every weight row is scaled by a power of two so that its largest magnitude lies in [1, 2) before the split (any
        // finite float32 weight is then inside fp16 range with 22 bits relative to the row maximum); the inverse scale is
        // applied to the accumulator of the lane that owns the row's neuron.
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
            float iz, ir, ic;
            const float sz = pow2_scale(kgroup_max(mz), iz), sr = pow2_scale(kgroup_max(mr), ir), sc = pow2_scale(kgroup_max(mc), ic);
            // the lane that consumes row 4g + q of this tile (lane index = row index in k group 0 holds its inverse scale)
            inv_z[p] = __shfl(iz, 4 * g + q); inv_r[p] = __shfl(ir, 4 * g + q); inv_c[p] = __shfl(ic, 4 * g + q);
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
        // B operand of K block kb: 16-byte piece (kb*4 + g)*4 + c of an image (the same piece for every q)
        int boff[KBS];
#pragma unroll
        for (int i = 0; i < KBS; i++) boff[i] = ((((w + i) % KBS) * 4 + g) * 4 + c) * 4;        // in dwords
        const int wd = ((w * 4 + g) * 4 + c) * 4 + q;                                           // my packed pair, in dwords
        const int n0 = 32 * w + 4 * g + q;                                                      // my neuron of tile 2w (+16: 2w+1)
        auto ldB = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

        __builtin_amdgcn_s_setprio(3);      // the serial chain goes first; projection MFMAs fill its gaps
        constexpr int NOWATCH = INT_MIN / 2;
        const int offH = cls == 1 ? 0 : NOWATCH;             // h(s-1) complete:   fB >= s
        const int offRH = cls == 0 ? 1 : NOWATCH;            // r*h of step s:     fA >= s+1
        const bool watch_flush = cls == 3;
        const int offV = (cls == 2 || cls == 4) ? 2 : NOWATCH;   // vI(s+1) written: every projection wave's vready >= s+2

        unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
        if constexpr (DIAG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory"); }
        float hold[2] = {0.0f, 0.0f};
        float vz[2], vr[2], vc[2];                           // vI rows of my two neurons, read one step ahead
        for (;;) {                                           // vI(0)
            const int f = poll_issue<32>(flags, lane);
#pragma unroll
            for (int p = 0; p < 2; p++) {
                vz[p] = vbuf[4 * (n0 + 16 * p) + c];
                vr[p] = vbuf[4 * (N + n0 + 16 * p) + c];
                vc[p] = vbuf[4 * (2 * N + n0 + 16 * p) + c];
            }
            const bool ok = poll_result(f, (cls == 2 || cls == 4) ? 1 : NOWATCH) || (ABL & 2);
#pragma unroll
            for (int p = 0; p < 2; p++) { keep(vz[p]); keep(vr[p]); keep(vc[p]); }
            if (ok) break;
        }

        for (int s = 0; s < T; s++) {
            const int needH = s + offH, needRH = s + offRH;
            const int needV = watch_flush ? (s + 1) / KB - 2 : s + offV;
            half8 bh[KBS], bl[KBS];
            f32x4 accR[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accZ[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

            STAMP16(0)
            // Schedule of one step (MFMA = 16 pipe cycles, of which the issuing wave is busy for 8: the other 8 take VALU):
            //   r  own block (6 MFMAs) while the other waves' h(s-1) arrives, then r others (12)      -- the critical path
            //   z  all blocks (18)      || sigmoid(r), r*h, split, write, publish; own r*h read back
            //   c  own block (6) while the others' r*h arrives, c others (12)   || sigmoid(z), vI(s+1) reads
            //   tanh, blend, split, write, publish
            // ---------------- r: products with h(s-1) ----------------
            bh[0] = ldB(h_hi, boff[0]);
            bl[0] = ldB(h_lo, boff[0]);
            int f = poll_issue<32>(flags, lane);
#pragma unroll
            for (int i = 1; i < KBS; i++) { bh[i] = ldB(h_hi, boff[i]); bl[i] = ldB(h_lo, boff[i]); }
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (KBS - 1) + 1) : "memory");          // own block has landed
            keep(bh[0]); keep(bl[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; p++) accR[p] = mfma3<ABL>(wr_hi[p][0], wr_lo[p][0], bh[0], bl[0], accR[p]);
            if constexpr (KBS > 1) {
                bool ok = poll_result(f, needH) || (ABL & 2);
#pragma unroll
                for (int i = 1; i < KBS; i++) { keep(bh[i]); keep(bl[i]); }
                while (!ok) {
                    if constexpr (DIAG) stamp_acc[8]++;
                    f = poll_issue<32>(flags, lane);
#pragma unroll
                    for (int i = 1; i < KBS; i++) { bh[i] = ldB(h_hi, boff[i]); bl[i] = ldB(h_lo, boff[i]); }
                    ok = poll_result(f, needH);
#pragma unroll
                    for (int i = 1; i < KBS; i++) { keep(bh[i]); keep(bl[i]); }
                }
                STAMP16(1)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 1; i < KBS; i++)
#pragma unroll
                    for (int p = 0; p < 2; p++) accR[p] = mfma3<ABL>(wr_hi[p][i], wr_lo[p][i], bh[i], bl[i], accR[p]);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f)::"memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- z MFMAs, first part, with the r epilogue in their issue gaps ----------------
#pragma unroll
            for (int p = 0; p < 2; p++) accZ[p] = mfma3<ABL>(wz_hi[p][0], wz_lo[p][0], bh[0], bl[0], accZ[p]);
            float rr[2];
#pragma unroll
            for (int p = 0; p < 2; p++) rr[p] = (ABL & 4) ? fmaf(sel4(accR[p], q), inv_r[p], vr[p]) * 0.01f : slk_sigmoid(fmaf(sel4(accR[p], q), inv_r[p], vr[p]));
            {
                unsigned hi, lo;
                split2(rr[0] * hold[0], rr[1] * hold[1], hi, lo);
                rh_hi[wd] = hi;
                rh_lo[wd] = lo;
            }
            // one MFMA, then up to four VALU instructions, for as long as both last
#pragma unroll
            for (int i = 0; i < 6; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
            publish(flags, w, s + 1, lane);
            STAMP16(2)
            if constexpr (SAVE) {
                const size_t trow = (size_t)(reverse ? T - 1 - s : s) * B + b0 + c;
                if (b0 + c < B) {
                    zr_out[trow * (2 * N) + N + n0] = rr[0];
                    zr_out[trow * (2 * N) + N + n0 + 16] = rr[1];
                }
            }

            // ---------------- candidate: products with r*h ----------------
            f32x4 accC[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            half8 ch[KBS], cl[KBS];
            ch[0] = ldB(rh_hi, boff[0]);
            cl[0] = ldB(rh_lo, boff[0]);
            f = poll_issue<32>(flags, lane);
#pragma unroll
            for (int i = 1; i < KBS; i++) { ch[i] = ldB(rh_hi, boff[i]); cl[i] = ldB(rh_lo, boff[i]); }
            // the rest of the z MFMAs execute while the other waves' r*h is on its way
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < KBS; i++)
#pragma unroll
                for (int p = 0; p < 2; p++) accZ[p] = mfma3<ABL>(wz_hi[p][i], wz_lo[p][i], bh[i], bl[i], accZ[p]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (KBS - 1) + 1) : "memory");
            keep(ch[0]); keep(cl[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; p++) accC[p] = mfma3<ABL>(wc_hi[p][0], wc_lo[p][0], ch[0], cl[0], accC[p]);
            if constexpr (KBS > 1) {
                bool ok = poll_result(f, needRH) || (ABL & 2);
#pragma unroll
                for (int i = 1; i < KBS; i++) { keep(ch[i]); keep(cl[i]); }
                while (!ok) {
                    if constexpr (DIAG) stamp_acc[9]++;
                    f = poll_issue<32>(flags, lane);
#pragma unroll
                    for (int i = 1; i < KBS; i++) { ch[i] = ldB(rh_hi, boff[i]); cl[i] = ldB(rh_lo, boff[i]); }
                    ok = poll_result(f, needRH);
#pragma unroll
                    for (int i = 1; i < KBS; i++) { keep(ch[i]); keep(cl[i]); }
                }
                STAMP16(3)
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f)::"memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            // c others, with the update gate and the requests for vI(s+1) in their issue gaps
            // vI(s+1), one step ahead (its latency disappears behind the candidate chain)
            const float *vnext = vbuf + ((s + 1) % R) * (3 * N * 4);
            const bool more = s + 1 < T;
            int fv = 0;
            float vzn[2] = {vz[0], vz[1]}, vrn[2] = {vr[0], vr[1]}, vcn[2] = {vc[0], vc[1]};
            auto read_vnext = [&] {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    vzn[p] = vnext[4 * (n0 + 16 * p) + c];
                    vrn[p] = vnext[4 * (N + n0 + 16 * p) + c];
                    vcn[p] = vnext[4 * (2 * N + n0 + 16 * p) + c];
                }
            };
            if (more) {
                fv = poll_issue<32>(flags, lane);
                read_vnext();
            }
#pragma unroll
            for (int i = 1; i < KBS; i++)
#pragma unroll
                for (int p = 0; p < 2; p++) accC[p] = mfma3<ABL>(wc_hi[p][i], wc_lo[p][i], ch[i], cl[i], accC[p]);
            float zz[2];
#pragma unroll
            for (int p = 0; p < 2; p++) zz[p] = (ABL & 4) ? fmaf(sel4(accZ[p], q), inv_z[p], vz[p]) * 0.01f : slk_sigmoid(fmaf(sel4(accZ[p], q), inv_z[p], vz[p]));
            float omz[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                omz[p] = 1.0f - zz[p];
                asm volatile("" : "+v"(zz[p]), "+v"(omz[p]));                 // pinned here: not sunk to the blend below
            }
#pragma unroll
            for (int i = 0; i < 6 * (KBS - 1); i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SAVE) {
                const size_t trow = (size_t)(reverse ? T - 1 - s : s) * B + b0 + c;
                if (b0 + c < B) {
                    zr_out[trow * (2 * N) + n0] = zz[0];
                    zr_out[trow * (2 * N) + n0 + 16] = zz[1];
                }
            }
            STAMP16(4)
            float hn[2];
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const float hbar = (ABL & 4) ? fmaf(sel4(accC[p], q), inv_c[p], vc[p]) * 0.01f : slk_tanh(fmaf(sel4(accC[p], q), inv_c[p], vc[p]));
                hn[p] = zz[p] * hold[p] + omz[p] * hbar;                      // layers.py:1020
            }
            STAMP16(7)
            if (more) {                                      // long since answered; checked before the writes below queue up
                bool ok = poll_result(fv, needV) || (ABL & 2);
#pragma unroll
                for (int p = 0; p < 2; p++) { keep(vzn[p]); keep(vrn[p]); keep(vcn[p]); }
                while (!ok) {
                    if constexpr (DIAG) stamp_acc[10]++;
                    const int f2 = poll_issue<32>(flags, lane);
                    read_vnext();
                    ok = poll_result(f2, needV);
#pragma unroll
                    for (int p = 0; p < 2; p++) { keep(vzn[p]); keep(vrn[p]); keep(vcn[p]); }
                }
            }
            STAMP16(6)
            {
                unsigned hi, lo;
                split2(hn[0], hn[1], hi, lo);
                h_hi[wd] = hi;
                h_lo[wd] = lo;
                float *hcur = hring + (s % HSLOTS) * HIMG;
                hcur[4 * n0 + c] = hn[0];
                hcur[4 * (n0 + 16) + c] = hn[1];
            }
            publish(flags, 4 + w, s + 1, lane);
#pragma unroll
            for (int p = 0; p < 2; p++) { hold[p] = hn[p]; vz[p] = vzn[p]; vr[p] = vrn[p]; vc[p] = vcn[p]; }
            STAMP16(5)
        }
        if constexpr (DIAG) {
            if (blockIdx.x == 0 && tid == 0)
                for (int i = 0; i < 12; i++) slk_dbg_stamp16[i] = stamp_acc[i];
        }
    } else {
        // =================================================================================================
        // projection waves (vI = x.iW^T + b four steps at a time); the service waves 4-7 also run the x DMA and copy
        // finished blocks of states to h_out
        // =================================================================================================
        const bool service = wave >= 4;
        const int pw = wave - 4;                            // service index 0..3
        const int ew = wave - NCW;                          // extra index 0..3-NCW (waves NCW..3)
        const bool on_free_simd = (wave & 3) >= NCW;
        const int free_idx = wave < 4 ? wave - NCW : (4 - NCW) + (wave - 4 - NCW);
        const int ntile = on_free_simd ? FREE_EACH : SH_BASE + ((wave - 4) < SH_EXTRA ? 1 : 0);
        const int tile0 = on_free_simd ? free_idx * FREE_EACH
                                       : NFREE * FREE_EACH + (wave - 4) * SH_BASE + min(wave - 4, SH_EXTRA);
        const int my_tiles = max(0, min(ntile, NT16 - tile0));
        const int col = lane & 15, kq = lane >> 4;
        // B operands: lane holds vI row 16*t + col, k = 32*kb + 8*kq + 0..7, as fp16 hi and lo parts
        // (rows scaled by a power of two like the recurrent weights; the lane that holds a row is the lane that stores it)
        half8 whi[MAXT][KBLK], wlo[MAXT][KBLK];
        float inv_w[MAXT];
#pragma unroll
        for (int i = 0; i < MAXT; i++) {
            const bool ok = i < my_tiles;
            const int row = ok ? 16 * (tile0 + i) + col : 0;
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
        auto dma_block = [&](int s0, int slot) {             // service waves 0, 1: NPER requests each
#pragma unroll
            for (int j = 0; j < NPER; j++) {
                const int piece0 = (j * 2 + pw) * 64;
                if (piece0 < XPIECES) {
                    const int p = piece0 + lane;
                    const int kk = p / I, pp = p % I, qq = pp >> 2, cc = pp & 3;
                    const int bc = min(b0 + cc, B - 1);
                    // ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step, and the
                    // steps past the end re-read the last valid row (their results are never stored)
                    const int Tc = lens ? min(max(lens[bc], 1), T) : T;
                    const int ss = min(s0 + kk, Tc - 1);
                    const int tt = reverse ? Tc - 1 - ss : ss;
                    const float *src = x + ((size_t)tt * B + bc) * ldx + 4 * qq;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                     (__attribute__((address_space(3))) void *)&xbuf[slot * (KB * XIMG) + piece0 * 4],
                                                     16, 0, 0);
                } else {
                    // keep the number of requests per block fixed (the waits below count them): repeat the first piece
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + 4 * (lane % (I / 4))),
                                                     (__attribute__((address_space(3))) void *)&xbuf[slot * (KB * XIMG) + XPIECES * 4 - 256],
                                                     16, 0, 0);
                }
            }
        };
        const bool vec_store = (ldh % 4 == 0) && ((reinterpret_cast<uintptr_t>(h_out) & 15) == 0);
        constexpr int OF4 = KB * N;                         // float4s per block
        constexpr int NFL = (OF4 + 127) / 128;              // parts per block (service waves 2, 3: 128 threads)
        auto flush_part = [&](int kb, int j) {
            const int idx = (tid - 384) + 128 * j;
            const int cc = idx & 3, rest = idx >> 2, f4 = rest % (N / 4), kk = rest / (N / 4);
            const int ss = kb * KB + kk;
            const int Tc = (lens && b0 + cc < B) ? min(max(lens[b0 + cc], 1), T) : T;
            if (idx < OF4 && ss < Tc && b0 + cc < B) {
                const float *src = hring + (ss % HSLOTS) * HIMG + 16 * f4 + cc;
                const float v0 = src[0], v1 = src[4], v2 = src[8], v3 = src[12];
                const int tt = reverse ? Tc - 1 - ss : ss;
                float *dst = h_out + ((size_t)tt * B + b0 + cc) * ldh + 4 * f4;
                if (vec_store) *reinterpret_cast<float4 *>(dst) = make_float4(v0, v1, v2, v3);
                else { dst[0] = v0; dst[1] = v1; dst[2] = v2; dst[3] = v3; }
            }
        };
        auto wait_flags = [&](int group, int value) {       // every counter of `group` (0 fA, 1 fB) >= value
            const int need = cls == group ? value : INT_MIN;
            while (!reached<32>(flags, lane, need)) __builtin_amdgcn_s_sleep(1);
        };
        auto wait_xready = [&](int value) {
            const int need = cls == 0 ? value : INT_MIN;
            while (!reached<32>(xflags, lane, need)) __builtin_amdgcn_s_sleep(1);
        };
        auto wait_xdone = [&](int value) {                  // service and extra waves past block value-1
            const int need = (cls == 1 || cls == 2) ? value : INT_MIN;
            while (!reached<32>(xflags, lane, need)) __builtin_amdgcn_s_sleep(1);
        };

        const bool dma_wave = service && pw < 2, flush_wave = service && pw >= 2;
        unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ptprev = 0;
        if constexpr (DIAG) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ptprev)::"memory"); }
        const int NBLK = (T + KB - 1) / KB;
        if (dma_wave)
            for (int xb = 0; xb < XSLOTS - 1 && xb < NBLK; xb++) dma_block(xb * KB, xb);
        const int NG = (T + GS - 1) / GS;
        for (int qg = 0; qg < NG; qg++) {
            PSTAMP(0)
            if ((qg & 1) == 0) {                                        // KB = 2 groups: a new x block starts here
                const int xb = qg / 2;
                if (dma_wave) {
                    // my share of block xb has landed once at most the requests of the younger blocks are outstanding
                    // (these waves issue no other vector memory operation; requests complete in order)
                    const int younger = min(XSLOTS - 2, NBLK - 1 - xb);
                    if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPER) : "memory");
                    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPER) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    publish(xflags, pw, xb + 1, lane);
                }
                if (service) publish(xflags, 4 + pw, xb, lane);         // I am done reading block xb-1
                else publish(xflags, 8 + ew, xb, lane);
                if (dma_wave && xb + XSLOTS - 1 < NBLK) {
                    wait_xdone(xb);                                     // the slot held block xb-1: everyone past it
                    dma_block((xb + XSLOTS - 1) * KB, (xb + XSLOTS - 1) % XSLOTS);
                }
                wait_xready(xb + 1);
            } else if (flush_wave && qg >= 3) {
                // copy a finished block of states out (vI is published through step 4qg-1, block fkb ends at step 4qg-5)
                const int fkb = (qg - 3) / 2;
                wait_flags(1, (fkb + 1) * KB);
                for (int j = 0; j < NFL; j++) flush_part(fkb, j);
                publish(flags, 12 + pw, fkb + 1, lane);
            }
            PSTAMP(1)
            if (my_tiles > 0 && !(ABL & 8)) {
                // ---- the group's A operands (x split into halves on the fly; each row scaled by a power of two so that its
                //      largest |x| lies in [1, 2) -- exact, undone on the accumulators), then every tile's three MFMAs ----
                const float *img = xbuf + ((qg >> 1) % XSLOTS) * (KB * XIMG) + (GS * (qg & 1) + a_step) * XIMG + 4 * a_chunk;
                // (two passes over the image instead of holding it: the weights leave few registers)
                float amax = 0.0f;
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) {
                    const int k0 = 32 * kb + 8 * kq;
                    const bool kok = (I % 32 == 0) || k0 < I;
                    const float *src = img + 4 * (kok ? k0 : 0);           // 16 * (k0 / 4)
                    const f32x4 u0 = *reinterpret_cast<const f32x4 *>(src), u1 = *reinterpret_cast<const f32x4 *>(src + 16);
#pragma unroll
                    for (int j = 0; j < 4; j++) amax = fmaxf(amax, kok ? fmaxf(fabsf(u0[j]), fabsf(u1[j])) : 0.0f);
                }
                amax = kgroup_max(amax);
                float xinv;
                const float xs = pow2_scale(amax, xinv);
                // the accumulator rows of this lane are (step kq, chunk 0..3): their inverse scales sit in lanes 4*kq + (0..3)
                f32x4 inv;
#pragma unroll
                for (int r4 = 0; r4 < 4; r4++) inv[r4] = __shfl(xinv, 4 * kq + r4);
                f32x4 acc[MAXT];
#pragma unroll
                for (int i = 0; i < MAXT; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < KBLK; kb++) {
                    const int k0 = 32 * kb + 8 * kq;
                    const bool kok = (I % 32 == 0) || k0 < I;
                    const float *src = img + 4 * (kok ? k0 : 0);
                    const f32x4 u0 = *reinterpret_cast<const f32x4 *>(src), u1 = *reinterpret_cast<const f32x4 *>(src + 16);
                    half8 ahi, alo;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float v = kok ? (j < 4 ? u0[j & 3] : u1[j & 3]) * xs : 0.0f;
                        const _Float16 h = (_Float16)v;
                        ahi[j] = h;
                        alo[j] = (_Float16)(v - (float)h);
                    }
#pragma unroll
                    for (int i = 0; i < MAXT; i++) {
                        if (i < my_tiles) {
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, wlo[i][kb], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, whi[i][kb], acc[i], 0, 0, 0);
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, whi[i][kb], acc[i], 0, 0, 0);
                        }
                    }
                }
                PSTAMP(2)
                // the group's ring slots were last read during steps GS*qg - R ... GS*qg + GS-1 - R
                if (GS * qg + GS > R) wait_flags(0, GS * qg + GS - R);
                PSTAMP(3)
                // D: lane holds the four chunks of (step GS*qg + kq, vI row 16*t + col) = one 16-byte entry of vbuf
                const int st = GS * qg + kq;
                if (st < T) {
                    float *vdst = vbuf + (st % R) * (3 * N * 4) + 4 * col;
#pragma unroll
                    for (int i = 0; i < MAXT; i++)
                        if (i < my_tiles) {
                            const float tb = bias_lds[16 * (tile0 + i) + col];
                            f32x4 o;
#pragma unroll
                            for (int r4 = 0; r4 < 4; r4++) o[r4] = fmaf(acc[i][r4] * inv[r4], inv_w[i], tb);
                            *reinterpret_cast<f32x4 *>(&vdst[64 * (tile0 + i)]) = o;
                        }
                }
            }
            publish(flags, service ? 8 + pw : 16 + ew, GS * qg + GS, lane);
            PSTAMP(4)
        }
        if constexpr (DIAG) {
            if (blockIdx.x == 0 && lane == 0)
                for (int i = 0; i < 8; i++) slk_dbg_pstamp16[wave][i] = pacc[i];
        }
        if (flush_wave) {
            // blocks of states the loop did not copy out
            wait_flags(1, T);
            const int kbl = (T - 1) / KB;
            for (int kb = 0; kb <= kbl; kb++)
                if (2 * kb + 3 >= NG)
                    for (int j = 0; j < NFL; j++) flush_part(kb, j);
        }
    }
}

// One workgroup per CU (see gru_fused.hip): ask for enough dynamic LDS that two cannot share a CU.
template <typename K>
static size_t exclusive_cu_lds16(K kernel)
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
static int launch_fused16(const float *x, long ldx, const float *iW, const float *bias, const float *sW, const float *sW2,
                          float *y, long ldy, int T, int B, int reverse, const int *lens, float *zr_out, hipStream_t s)
{
    if constexpr (I == 96 && N == 96) {
        const int dv = reverse >> 1;                    // diagnostic launches (undocumented bits, tools/bench_kernels.py)
#define DIAG_LAUNCH(CODE, STAMPS, ABLV)                                                                                   \
        if (dv == CODE) {                                                                                                 \
            const size_t dyn = SLK_PER_DEVICE(size_t, exclusive_cu_lds16(gru_fused16_kernel<I, N, false, STAMPS, ABLV>));                  \
            hipLaunchKernelGGL((gru_fused16_kernel<I, N, false, STAMPS, ABLV>), dim3((B + 3) / 4), dim3(512), dyn, s, x, ldx, \
                               iW, bias, sW, sW2, y, ldy, T, B, reverse & 1, lens, zr_out);                                \
            return slk_launch_status();                                                                                    \
        }
        DIAG_LAUNCH(1, true, 0) DIAG_LAUNCH(2, false, 1) DIAG_LAUNCH(3, false, 2) DIAG_LAUNCH(4, false, 4) DIAG_LAUNCH(5, false, 8)
        DIAG_LAUNCH(6, false, 3) DIAG_LAUNCH(7, false, 7) DIAG_LAUNCH(8, false, 15) DIAG_LAUNCH(9, false, 10) DIAG_LAUNCH(10, true, 3)
#undef DIAG_LAUNCH
    }
    if (zr_out) {
        const size_t dyn = SLK_PER_DEVICE(size_t, exclusive_cu_lds16(gru_fused16_kernel<I, N, true>));
        hipLaunchKernelGGL((gru_fused16_kernel<I, N, true>), dim3((B + 3) / 4), dim3(512), dyn, s, x, ldx, iW, bias, sW, sW2, y,
                           ldy, T, B, reverse & 1, lens, zr_out);
    } else {
        const size_t dyn = SLK_PER_DEVICE(size_t, exclusive_cu_lds16(gru_fused16_kernel<I, N, false>));
        hipLaunchKernelGGL((gru_fused16_kernel<I, N, false>), dim3((B + 3) / 4), dim3(512), dyn, s, x, ldx, iW, bias, sW, sW2, y,
                           ldy, T, B, reverse & 1, lens, zr_out);
    }
    return slk_launch_status();
}

// Whole Gru layer with projection AND recurrence as 3-term fp16 splits (float32 accumulation).  lens: NULL or the ragged
// lengths (see slk_gru_fused_ragged_f32); zr_out: NULL or [T*B][2n] for the activated gates (training forward pass).
// SLK_ERR_UNSUPPORTED when no instantiation covers the request: the caller falls back to slk_gru_fused_f32 (fp32 recurrence).
extern "C" int slk_gru_fused16_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2,
                                   const float *bias, float *y, long ldy, int T, int B, int insize, int n, int reverse,
                                   int act, int gate_act, const int32_t *lens, float *zr_out, slk_stream_t stream)
{
    if (!x || !iW || !sW || !sW2 || !y || T < 1 || B < 1 || insize < 1 || n < 1 || ldx < insize || ldy < n)
        return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if ((ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;   // 16-byte DMA pieces
    hipStream_t s = slk_stream(stream);
#define FUSED16(II, NN) \
    if (insize == II && n == NN) return launch_fused16<II, NN>(x, ldx, iW, bias, sW, sW2, y, ldy, T, B, reverse, lens, zr_out, s);
    FUSED16(96, 96) FUSED16(64, 64) FUSED16(32, 96) FUSED16(128, 96) FUSED16(64, 96) FUSED16(48, 32) FUSED16(16, 64)
#undef FUSED16
    return SLK_ERR_UNSUPPORTED;
}
