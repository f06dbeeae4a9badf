// gru_backward_mfma.hip -- the reverse scan of a Gru layer (see csrc/train.hip for the maths and the other two kernels)
// on the fp32 matrix pipe: four chunks per workgroup as the four rows of v_mfma_f32_4x4x1_f32, the mirror image of
// gru_mfma_kernel (recurrent.hip).  Per step and chunk the scan does 3 N^2 multiply-adds in two dependent matrix-vector
// products (drh = dac . sW2, N -> N; carry += [daz dar] . sW, 2N -> N); the VALU kernels of train.hip are bound by FMA issue
// plus a latency skeleton of barriers and LDS round trips, here the products cost a quarter of the issue slots.
//
//   * 4 waves, wave w owns neurons w*N/4 .. ; both products are in "one output column per lane, K split in S slices" form
//     (the candidate phase of gru_mfma_kernel), with the TRANSPOSED weights in registers;
//   * the operands of a step ([dy | z | r | h_t | h_prev], 5N floats per chunk) are brought in by LDS-DMA D steps ahead by a
//     fifth, loader wave: the compute waves store da / rh every step and a wave with stores in flight can only wait for a
//     load with vmcnt(0), while the loader's counted vmcnt says exactly "the step needed next has landed";
//   * two LDS-only barriers per step; the exchanged vectors are double-buffered by step parity.
#include "common.h"
#include "mfma4.h"

__device__ __forceinline__ float gru_candidate_mfma(float h_t, float z, float h)
{
    const float omz = 1.0f - z;                     // see gru_candidate in train.hip
    return omz > 0.0f ? slk_clip((h_t - z * h) * slk_rcp(omz), -1.0f, 1.0f) : 0.0f;
}

#ifdef GBM_DIAG
__device__ unsigned long long gbm_stamps[8];
#define GBM_STAMP(k)                                                                       \
    {                                                                                      \
        unsigned long long tn_;                                                            \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn_)::"memory");        \
        stamp_acc[k] += tn_ - tprev;                                                       \
        tprev = tn_;                                                                       \
    }
extern "C" int slk_gbm_read_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gbm_stamps), sizeof(gbm_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define GBM_STAMP(k)
#endif

template <int N>
__global__ void __launch_bounds__(320) gru_backward_mfma_kernel(const float *__restrict__ dy, long lddy,
                                                                const float *__restrict__ hprev, long ldhp,
                                                                const float *__restrict__ zr,
                                                                const float *__restrict__ hout, long ldh,
                                                                const float *__restrict__ sW, const float *__restrict__ sW2,
                                                                float *__restrict__ da, float *__restrict__ rh, int T, int B,
                                                                int reverse)
{
    constexpr int NW = N / 4;                                        // neurons per wave
    constexpr int S = (NW <= 16) ? 4 : ((NW <= 32) ? 2 : 1);         // K-slices of both products
    constexpr int LP = 64 / S;                                       // lanes per slice
    constexpr int M1 = N / S, M2 = 2 * N / S;                        // MFMAs per product
    constexpr int G = 16 / S, CB = 4 - ilog2(S);
    constexpr int NV1 = N / 16, NV2 = 2 * N / 16;                    // packed operand registers
    static_assert(N % 16 == 0 && NW % 4 == 0 && NW <= 32, "unsupported size for the MFMA reverse scan");
    constexpr int D = 4;                                             // operand ring: steps in flight ahead of their use
    static_assert(20 * (D - 1) <= 63, "vmcnt is a 6-bit counter");

    __shared__ __attribute__((aligned(16))) float obuf[D][4][5 * N];   // [step % D][chunk][dy | z | r | h_t | h_prev]
    __shared__ __attribute__((aligned(16))) float dacbuf[2][N * 4];     // [parity] dac[k][chunk]
    __shared__ __attribute__((aligned(16))) float dzrbuf[2][2 * N * 4]; // [parity] [daz | dar][k][chunk]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool loader = wave == 4;
    const int b0 = blockIdx.x * 4;
    const int blk = lane >> 2, ci = lane & 3;
    const int lb = lane % LP, gb = lane / LP;
    const bool validB = !loader && lb < NW;
    const int neuron = (loader ? 0 : wave * NW) + (validB ? lb : 0);

    // transposed weights: product 1 column `neuron` of sW2, product 2 column `neuron` of sW, rows of this lane's K-slice
    float w1[M1], w2[M2];
#pragma unroll
    for (int m = 0; m < M1; m++) w1[m] = validB ? sW2[(size_t)(gb * M1 + m) * N + neuron] : 0.0f;
#pragma unroll
    for (int m = 0; m < M2; m++) w2[m] = validB ? sW[(size_t)(gb * M2 + m) * N + neuron] : 0.0f;
    const int addr1 = 4 * ((blk / G) * M1 + (blk % G)) + ci;
    const int addr2 = 4 * ((blk / G) * M2 + (blk % G)) + ci;

    auto row = [&](int s, int c) { return (size_t)(reverse ? T - 1 - s : s) * B + min(b0 + c, B - 1); };
    // loader wave: the 20 operand rows of one scan step per call, walking down from step T-1.  Row pointers are kept per
    // chunk and stepped by the time stride -- recomputing 20 64-bit addresses per step made the loader the last wave at
    // the barrier.
    const long tstep_rows = reverse ? (long)B : -(long)B;             // rows from scan step s to s-1
    const float *src[4][5];
    if (loader) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const size_t m = row(T - 1, c);
            src[c][0] = dy + m * lddy;                                // wave-uniform: lives in scalar registers, stepped by
            src[c][1] = zr + m * (2 * N);                             // scalar adds; the lane's 16 bytes are a 32-bit offset
            src[c][2] = zr + m * (2 * N) + N;
            src[c][3] = hout + m * ldh;
            src[c][4] = hprev + m * ldhp;
        }
    }
    const long step5[5] = {tstep_rows * lddy, tstep_rows * (2 * N), tstep_rows * (2 * N), tstep_rows * ldh, tstep_rows * ldhp};
    auto issue = [&](int sp) {                                       // must be called for sp = T-1, T-2, ... in turn
        float *dst = &obuf[sp % D][0][0];
        const unsigned voff = 16u * (unsigned)lane;
        if (lane < N / 4) {
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int a = 0; a < 5; a++)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(src[c][a]) + voff),
                        (__attribute__((address_space(3))) void *)(dst + (c * 5 + a) * N), 16, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int a = 0; a < 5; a++) src[c][a] += step5[a];
    };

    if (loader) {
        for (int sp = T - 1; sp >= 0 && sp > T - 1 - D; sp--) issue(sp);
    } else {
        __builtin_amdgcn_s_setprio(3);               // the loader shares a SIMD with compute wave 0: the serial chain goes first
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // element-wise part: every K-slice holds the complete sums after sum_slices, so slice gb finishes chunks gb*CPL ..
    // (CPL = 4 / S of the four) instead of slice 0 doing all four one after the other -- it is on the critical path
    constexpr int CPL = 4 / S;
    const bool owner = validB;
    const int c0 = gb * CPL;
    float *dap[CPL], *rhp[CPL];                                       // this lane's output rows, stepped by the time stride
    bool live[CPL];
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        const size_t m = row(T - 1, c0 + i);
        dap[i] = da + m * (3 * N) + neuron;
        rhp[i] = rh + m * N + neuron;
        live[i] = owner && b0 + c0 + i < B;
    }
    const long tstep = tstep_rows;

    float carry[CPL];
#pragma unroll
    for (int i = 0; i < CPL; i++) carry[i] = 0.0f;
    // what a step needs that does not depend on the carry -- gates, state, the two gradient factors
    //   fa = (1-z)(1-c^2)   (dac = g fa)        fz = (h-c) z (1-z)   (daz = g fz)
    // -- is prepared while the previous step's second product runs on the matrix pipe; only g = dy + carry and two
    // multiplications remain on the serial chain
    float p_dy[CPL], z[CPL], r[CPL], h[CPL], fa[CPL], fz[CPL];
    auto prepare = [&](int sp) {
        const float *op = &obuf[sp % D][0][0];
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const float *o = op + (c0 + i) * (5 * N) + neuron;
            p_dy[i] = o[0]; z[i] = o[N]; r[i] = o[2 * N]; h[i] = o[4 * N];
            const float cc = gru_candidate_mfma(o[3 * N], z[i], h[i]);
            fa[i] = (1.0f - z[i]) * (1.0f - cc * cc);
            fz[i] = (h[i] - cc) * z[i] * (1.0f - z[i]);
        }
    };
    prepare(T - 1);
#ifdef GBM_DIAG
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
    for (int it = 0; it < T; it++) {
        const int s = T - 1 - it;
        GBM_STAMP(0)
        float *dacb = dacbuf[it & 1], *dzrb = dzrbuf[it & 1];
        float g[CPL];
        if (owner) {
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                g[i] = p_dy[i] + carry[i];
                const float dac = g[i] * fa[i], daz = g[i] * fz[i];
                dacb[4 * neuron + c0 + i] = dac;
                dzrb[4 * neuron + c0 + i] = daz;
                if (live[i]) {
                    dap[i][0] = daz;
                    dap[i][2 * N] = dac;
                }
            }
        }
        GBM_STAMP(1)
        lds_barrier();                                               // 1: dac visible; ring slot s % D is free again
        GBM_STAMP(2)
        // ---------------- product 1: drh = dac . sW2 ----------------
        float keep[CPL], rh_now[CPL];
#pragma unroll
        for (int i = 0; i < CPL; i++) keep[i] = 0.0f;
        if (loader) {
            // step s-1 is read after barrier 2; the D-1 younger steps (20 loads each) may stay in flight -- unless fewer than
            // that were issued (the last D steps), where everything is awaited
            if (s - D >= 0) {
                issue(s - D);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(20 * (D - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            float vp[NV1];
#pragma unroll
            for (int v = 0; v < NV1; v++) vp[v] = dacb[addr1 + 4 * v * G];
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain<CB, G>(vp, w1, acc, std::make_integer_sequence<int, M1>{});
            const f32x4 drh = sum_slices<S>((acc[0] + acc[1]) + (acc[2] + acc[3]));
            if (owner) {
#pragma unroll
                for (int i = 0; i < CPL; i++) {
                    const float d = drh[c0 + i];
                    rh_now[i] = r[i] * h[i];
                    const float dar = d * rh_now[i] * (1.0f - r[i]);
                    keep[i] = g[i] * z[i] + d * r[i];
                    dzrb[4 * (N + neuron) + c0 + i] = dar;
                    if (live[i]) {
                        dap[i][N] = dar;
                        rhp[i][0] = rh_now[i];
                    }
                }
            }
        }
        GBM_STAMP(3)
        lds_barrier();                                               // 2: [daz dar] visible; ring slot (s-1) % D has landed
        GBM_STAMP(4)
        // ---------------- product 2: carry = keep + [daz dar] . sW ----------------
        if (!loader) {
            float vp[NV2];
#pragma unroll
            for (int v = 0; v < NV2; v++) vp[v] = dzrb[addr2 + 4 * v * G];
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain<CB, G>(vp, w2, acc, std::make_integer_sequence<int, M2>{});
            prepare(s > 0 ? s - 1 : 0);                             // every lane, no branch: one basic block with the MFMAs
            // an in-order wave only overlaps the two pipes if the instructions alternate in program order: ask for
            // 1 MFMA : 1 LDS/VALU/transcendental pattern over the preparation (~70 instructions)
#pragma unroll
            for (int k = 0; k < 5 * CPL; k++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int k = 0; k < 30 * CPL; k++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 1, 0);
            }
            const f32x4 tot = sum_slices<S>((acc[0] + acc[1]) + (acc[2] + acc[3]));
#pragma unroll
            for (int i = 0; i < CPL; i++) carry[i] = keep[i] + tot[c0 + i];
        }
        GBM_STAMP(5)
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            dap[i] += tstep * (3 * N);
            rhp[i] += tstep * N;
        }
        // no third barrier: the next step writes the other parity's buffers, and these are rewritten two steps on, after every
        // wave has passed two more barriers
    }
#ifdef GBM_DIAG
    if (blockIdx.x == 0 && (tid == 0 || tid == 256))
        for (int k = 0; k < 4; k++) gbm_stamps[(tid == 0 ? 0 : 4) + k] = k == 0 ? stamp_acc[0] + stamp_acc[1] : stamp_acc[k + 1] + (k == 3 ? stamp_acc[5] : 0);
#endif
}

template <int N>
static int launch_gru_backward_mfma(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                                    long ldh, const float *sW, const float *sW2, float *da, float *rh, int T, int B,
                                    int reverse, hipStream_t s)
{
    hipLaunchKernelGGL((gru_backward_mfma_kernel<N>), dim3((B + 3) / 4), dim3(320), 0, s, dy, lddy, hprev, ldhp, zr, h, ldh, sW,
                       sW2, da, rh, T, B, reverse);
    return slk_launch_status();
}

// Returns SLK_ERR_UNSUPPORTED when the VALU kernels of train.hip have to be used (size, alignment).
int slk_gru_backward_mfma_dispatch(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                                   long ldh, const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n,
                                   int reverse, hipStream_t s)
{
    const bool aligned = lddy % 4 == 0 && ldhp % 4 == 0 && ldh % 4 == 0 &&
                         ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(hprev) |
                           reinterpret_cast<uintptr_t>(zr) | reinterpret_cast<uintptr_t>(h)) & 15) == 0;
    if (!aligned) return SLK_ERR_UNSUPPORTED;
    switch (n) {
#define CASE(NN) case NN: return launch_gru_backward_mfma<NN>(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, reverse, s)
        CASE(16); CASE(32); CASE(48); CASE(64); CASE(96); CASE(112);      // 128 spills: the VALU kernels take it
#undef CASE
    default: return SLK_ERR_UNSUPPORTED;
    }
}
