// lstm_scan16.hip -- the scan of Lstm.step (sloika/layers.py:677-691, peepholes, interleaved gate rows) on the execution plan of
// gru_scan16.hip: four waves per workgroup, one per SIMD, four chunks per workgroup, the recurrent products as 3-term fp16 splits
// on v_mfma_f32_16x16x32_f16 with rows scaled by powers of two, the state exchanged as packed hi / lo halves through LDS, a lane
// owning one (unit, chunk) pair.  The Lstm step has ONE matrix product (h(s-1) against the 4n x n matrix sW) and one exchange, so
// with the state image double buffered a step is one s_barrier:
//
//     barrier -> fetch h(s-1) (one operand per 32-wide K block: hi or lo by column group) -> 4 gates x K blocks x 2 MFMAs on
//     independent accumulators -> gate arithmetic of my (unit, chunk): five transcendentals -> my half of a state dword into the OTHER image,
//     the output row to HBM
//
// N = 64: wave w owns units 16w .. 16w+15, 64 weight registers; N = 128: units 32w .. 32w+31, two unit tiles, all 256 weight registers
// in accumulation registers (their MFMAs are asm).  Sizes in between (a multiple of 16) run with zero weights for the missing units:
// their state stays 0.  Products take two MFMAs each (the state's hi and lo halves in different column groups, bar16_common.h).  The input projection vW = x.iW^T + b comes from HBM (the row GEMM writes it): one 16-byte load per lane and step --
// the four gate pre-activations of a unit are neighbours (row = 4*unit + gate, layers.py:682-690) -- requested three steps ahead
// with asm loads the kernel counts itself (gru_scan16.hip).  The cell state of (unit, chunk) never leaves its lane's registers.
//
// lstm_mfma.hip (float32 MFMA 4x4x1, the SLOIKA_AMD_EXACT_F32 arithmetic) remains the all-fp32 path.
#include <limits.h>

#include "bar16_common.h"

__device__ __forceinline__ void lstm_gload4(f32x4 &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void lstm_pin4(f32x4 &v) { asm volatile("" : "+v"(v)); }

// one MFMA of a gate tile: weights as a builtin operand (N = 64: 64 weight registers) or named in accumulation registers (N = 128: 256)
template <bool AG, bool FIRST>
__device__ __forceinline__ void lstm_mma(f32x4 &acc, const half8 &wv, const half8 &bm)
{
    if constexpr (AG) {
        if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(wv), "v"(bm));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(wv), "v"(bm));
    } else {
        if constexpr (FIRST) acc = f32x4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, bm, acc, 0, 0, 0);
    }
}

// gemm_rows_f16x3.hip: a freshly defined builtin-MFMA destination must not take over operands that are still live
template <bool AGW>
__device__ __forceinline__ void lstm_keep(f32x4 &acc, const half8 &a, const half8 &b, const half8 &c)
{
    if constexpr (!AGW) asm volatile("" : "+v"(acc) : "v"(a), "v"(b), "v"(c));
}

template <int N>
__global__ void __launch_bounds__(256, 1) lstm_scan16_kernel(const float *__restrict__ vW, const float *__restrict__ sW,
                                                             const float *__restrict__ peep, float *__restrict__ h_out, long ldh, int T, int B,
                                                             int n, int reverse, const int *__restrict__ lens)
{
    static_assert(N == 64 || N == 128, "four waves of 16 or 32 units");
    constexpr int KBS = N / 32;
    constexpr int NUP = N / 64;                          // unit tiles per wave
    constexpr bool AG = N == 128;                        // weights in accumulation registers

    // [image parity][hi image 2N dwords | lo image 2N dwords]: element (k block kb, k group g, chunk c, r) = dword ((kb*4+g)*4+c)*4 + r
    // holds units 32kb+4g+r (low half) and 32kb+16+4g+r (high half) of chunk c -- the order in which a lane's eight B-operand
    // halves are consecutive.  The lanes of column groups q = 0, 1 fetch the hi image, q = 2, 3 the lo image (bar16_common.h,
    // mfma2x2: two MFMAs per product, pick_mix sums the column groups).
    __shared__ __attribute__((aligned(16))) unsigned h_img[2][2 * 2 * N];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < 2 * 2 * N; i += 256) { h_img[0][i] = 0u; h_img[1][i] = 0u; }                                   // h(-1) = 0
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // recurrent weights: A operands of the four gate tiles of each of my unit tiles (row = 4*unit + gate), rows scaled to [1, 2)
    half8 w_hi[NUP][4][KBS], w_lo[NUP][4][KBS];
    float inv[NUP][4];
#pragma unroll
    for (int p = 0; p < NUP; p++) {
        const int unit = 16 * NUP * w + 16 * p + (lane & 15);
        const bool uok = unit < n;
#pragma unroll
        for (int gt = 0; gt < 4; gt++) {
            const float *row = sW + (size_t)(4 * (uok ? unit : 0) + gt) * n;
            float v[KBS][8];
            float m = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                    v[kb][j] = (uok && k < n) ? row[k] : 0.0f;
                    m = fmaxf(m, fabsf(v[kb][j]));
                }
            }
            float iv;
            const float sc = pow2_scale(kgroup_max(m), iv);
            inv[p][gt] = __shfl(iv, 4 * g + q);
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a = v[kb][j] * sc;
                    const _Float16 h = (_Float16)a;
                    w_hi[p][gt][kb][j] = h;
                    w_lo[p][gt][kb][j] = (_Float16)(a - (float)h);
                }
            }
        }
    }
    // N = 128: every gate tile but the first lives in accumulation registers (224 of the 256; the first stays in ordinary ones so that
    // the compiler keeps a few accumulation registers for itself)
    if constexpr (AG) {
#pragma unroll
        for (int p = 0; p < NUP; p++)
#pragma unroll
            for (int gt = 0; gt < 4; gt++)
#pragma unroll
                for (int kb = 0; kb < KBS; kb++) {
                    if (p == 0 && gt == 0) continue;
                    w_hi[p][gt][kb] = to_acc_regs(w_hi[p][gt][kb]);
                    w_lo[p][gt][kb] = to_acc_regs(w_lo[p][gt][kb]);
                }
    }
    int moff[KBS];
#pragma unroll
    for (int kb = 0; kb < KBS; kb++) moff[kb] = (q >> 1) * 2 * N + ((kb * 4 + g) * 4 + c) * 4;    // in dwords, my column group's image
    // my (unit, chunk) pairs: unit u0 + 16p.  N = 128: wave w is K block w, both halves of dword wdw are mine; N = 64: wave w is
    // half (w & 1) of K block w >> 1
    const int u0 = 16 * NUP * w + 4 * g + q;
    bool uok[NUP];
    float p0[NUP], p1[NUP], p2[NUP];
#pragma unroll
    for (int p = 0; p < NUP; p++) {
        uok[p] = u0 + 16 * p < n;
        p0[p] = (peep && uok[p]) ? peep[u0 + 16 * p] : 0.0f;
        p1[p] = (peep && uok[p]) ? peep[n + u0 + 16 * p] : 0.0f;
        p2[p] = (peep && uok[p]) ? peep[2 * n + u0 + 16 * p] : 0.0f;
    }
    const int wdw = (((NUP == 2 ? w : (w >> 1)) * 4 + g) * 4 + c) * 4 + q;

    // my chunk's rows (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
    const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
    float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldh + (uok[0] ? u0 : 0);
    // vW of step s for my units: four consecutive floats each; steps past the chunk's end re-read its last row (never stored).
    // Four register sets, step s uses set s % 4 and requests step s + 3 into the set step s - 1 used; loads complete in order among
    // themselves, so once at most 3 * NUP memory operations are outstanding the current step's have arrived, whatever the stores do.
    f32x4 vs[4][NUP];
    const long ldv = 4L * n;
    unsigned voff = (unsigned)((((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldv + 4 * (uok[0] ? u0 : 0)) * sizeof(float));
    const unsigned vstep = (unsigned)((size_t)B * ldv * sizeof(float));
    const unsigned tile1 = (NUP == 2 && uok[NUP - 1]) ? 16u * 4u * (unsigned)sizeof(float) : 0u;     // bytes to my unit of the second tile
    int vnext = 0;
    auto load_v = [&](f32x4 (&v)[NUP]) {
        lstm_gload4(v[0], voff, vW);
        if constexpr (NUP == 2) lstm_gload4(v[1], voff + tile1, vW);
        vnext++;
        if (vnext < Tc) voff = reverse ? voff - vstep : voff + vstep;
    };
    load_v(vs[0]);
    load_v(vs[1]);
    load_v(vs[2]);

    float cell[NUP];
#pragma unroll
    for (int p = 0; p < NUP; p++) cell[p] = 0.0f;
    auto step = [&](auto PHC, const int s) {
        constexpr int ph = decltype(PHC)::value;
        constexpr int par = ph & 1;                      // h(s-1) is in image `par`, h(s) goes to the other one
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bm[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) bm[kb] = ldH(h_img[par], moff[kb]);
        load_v(vs[(ph + 3) & 3]);                        // three steps ahead
        f32x4 acc[NUP][4];
        // lo half of the weights first (small terms first); consecutive MFMAs go to different accumulators
        static_for<0, KBS>([&](auto KC) {
            constexpr int kb = decltype(KC)::value;
            static_for<0, 4 * NUP>([&](auto TC) {
                constexpr int p = decltype(TC)::value / 4, gt = decltype(TC)::value % 4;
                constexpr bool ag = AG && !(p == 0 && gt == 0);
                lstm_mma<ag, kb == 0>(acc[p][gt], w_lo[p][gt][kb], bm[kb]);
                if constexpr (kb == 0) lstm_keep<ag>(acc[p][gt], w_hi[p][gt][0], w_lo[p][gt][0], bm[0]);
            });
            static_for<0, 4 * NUP>([&](auto TC) {
                constexpr int p = decltype(TC)::value / 4, gt = decltype(TC)::value % 4;
                lstm_mma<(AG && !(p == 0 && gt == 0)), false>(acc[p][gt], w_hi[p][gt][kb], bm[kb]);
            });
        });
        if constexpr (NUP == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this step's vW (see above)
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        // pick_mix reads the accumulators from asm: let the matrix pipe drain (neither the compiler nor the asm MFMAs keep a distance)
        mfma_drain2(acc[NUP - 1][2], acc[NUP - 1][3]);
        float hn[NUP];
#pragma unroll
        for (int p = 0; p < NUP; p++) {
            f32x4 &cur = vs[ph][p];
            lstm_pin4(cur);
            // layers.py:686-691
            const float a0 = fmaf(pick_mix(acc[p][0]), inv[p][0], cur[0]), a1 = fmaf(pick_mix(acc[p][1]), inv[p][1], cur[1]);
            const float a2 = fmaf(pick_mix(acc[p][2]), inv[p][2], cur[2]), a3 = fmaf(pick_mix(acc[p][3]), inv[p][3], cur[3]);
            const float fg = sigmoid4(fmaf(cell[p], p1[p], a2));                 // forget gate
            const float ig = sigmoid4(fmaf(cell[p], p0[p], a1));                 // input gate
            const float cn = uok[p] ? fmaf(fg, cell[p], tanh5(a0) * ig) : 0.0f;  // new cell state
            const float og = sigmoid4(fmaf(cn, p2[p], a3));                      // output gate peeps at the NEW state
            hn[p] = uok[p] ? tanh5(cn) * og : 0.0f;
            cell[p] = cn;
        }
        if constexpr (NUP == 2) {
            unsigned hi, lo;
            split2(hn[0], hn[1], hi, lo);
            h_img[par ^ 1][wdw] = hi;
            h_img[par ^ 1][2 * N + wdw] = lo;
        } else {
            float hv = hn[0];
            asm volatile("" : "+v"(hv));                 // split2's note on v_fma_mixlo_f16 applies
            const _Float16 hh = (_Float16)hv;
            const _Float16 hl = (_Float16)(hv - (float)hh);
            reinterpret_cast<unsigned short *>(&h_img[par ^ 1][wdw])[w & 1] = __builtin_bit_cast(unsigned short, hh);
            reinterpret_cast<unsigned short *>(&h_img[par ^ 1][2 * N + wdw])[w & 1] = __builtin_bit_cast(unsigned short, hl);
        }
        if (live && s < Tc) {
            if (uok[0]) hp[0] = hn[0];
            if constexpr (NUP == 2) {
                if (uok[1]) hp[16] = hn[1];
            }
        }
        hp += hstep;
    };
    __syncthreads();                                     // LDS initialised
    for (int s = 0; s < T; s += 4) {
        step(ic<0>{}, s);
        if (s + 1 < T) step(ic<1>{}, s + 1);
        if (s + 2 < T) step(ic<2>{}, s + 2);
        if (s + 3 < T) step(ic<3>{}, s + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of mine may land in registers after the wave has ended
}

// One workgroup per CU (each wave is compiled for a whole SIMD): ask for enough dynamic LDS that two cannot share one.
template <int N>
static size_t lstm_scan16_exclusive_lds()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(lstm_scan16_kernel<N>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_scan16_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

// include/sloika_amd.h
extern "C" int slk_lstm_scan16_f32(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B, int n,
                                   int reverse, int act, int gate_act, const int32_t *lens, slk_stream_t stream)
{
    if (!vW || !sW || !out || T < 1 || B < 1 || n < 1 || ldo < n) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if (n % 16 || n > 128) return SLK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(vW) & 15) != 0) return SLK_ERR_UNSUPPORTED;
    if ((unsigned long long)T * B * 4 * n * sizeof(float) >= (1ull << 32)) return SLK_ERR_UNSUPPORTED;       // 32-bit lane offsets
    if (n > 64) {
        const size_t dyn = SLK_PER_DEVICE(size_t, lstm_scan16_exclusive_lds<128>());
        hipLaunchKernelGGL((lstm_scan16_kernel<128>), dim3((B + 3) / 4), dim3(256), dyn, slk_stream(stream), vW, sW, p, out, ldo, T, B, n,
                           reverse & 1, lens);
        return slk_launch_status();
    }
    const size_t dyn = SLK_PER_DEVICE(size_t, lstm_scan16_exclusive_lds<64>());
    hipLaunchKernelGGL((lstm_scan16_kernel<64>), dim3((B + 3) / 4), dim3(256), dyn, slk_stream(stream), vW, sW, p, out, ldo, T, B, n,
                       reverse & 1, lens);
    return slk_launch_status();
}
