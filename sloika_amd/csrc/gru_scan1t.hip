// gru_scan1t.hip -- the scan of Gru.step (sloika/layers.py:1010-1021) for wide layers with ONE 16-neuron tile per wave: the plan
// of gru_scan16.hip (four chunks per workgroup, two s_barrier per step, recurrent products as two fp16-split MFMAs per K block with
// the state's hi and lo halves in different column groups, the projection vI = x.iW^T + b read from HBM) on n / 16 waves instead of
// four.  A step is a latency chain; with two tiles per wave both tiles' activations, image writes and stores sit on it one after
// the other, with one tile per wave two waves share a SIMD and fill each other's waits (csrc/gru_bwd16.hip measured the same
// trade: 3350 against 2330 cycles per step).  Measured on whole layers (projection + scan, B = 1024, T' = 800): n = 112 1.11-1.19 ms
// against 1.21-1.28 for gru_scan16.hip, n = 128 1.19-1.22 against 1.23-1.26; `pretrained` architecture 709 -> 740 M samples/s.
// n = 144 is nine waves, three of them on one SIMD: 168 registers each.  As five K blocks of 32 the weights alone are 120 of them
// (round 3 fetched the candidate's from LDS every step instead: it spilled and ran 2.09 ms against 1.93 for gru_scan16.hip's ninth
// tile).  Round 5: 144 = four K blocks of 32 and ONE OF 16 -- v_mfma_f32_16x16x16_f16 for neurons 128 .. 143, whose halves the ninth
// wave writes side by side in the image so that one 8-byte LDS read is the B operand -- 108 registers of weights, no scratch:
// a 144-wide layer's scan 1.37 -> 0.94 ms (B = 1024, T' = 800, tools/scan_ab.py).
//
// A wave's recurrent weights are 3 gates x ceil(n / 32) K blocks x (hi, lo) x 4 registers = 96 of the 256 registers a wave has when
// two share a SIMD.
#include <limits.h>

#include "bar16_common.h"

__device__ __forceinline__ void g1_gload(float &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
template <bool FIRST>
__device__ __forceinline__ void g1_mma(f32x4 &acc, const half8 &wa, const half8 &bm)
{
    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wa), "v"(bm));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(bm));
}
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
// the K block of 16 (n = 144): lane (g, .) holds k = 4 g + i, i = 0 .. 3, of A's row and of B's column
__device__ __forceinline__ void g1_mma16(f32x4 &acc, const half4 &wa, const half4 &bm)
{
    asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(bm));
}

template <int N>
__global__ void __launch_bounds__(4 * N, 1) gru_scan1t_kernel(const float *__restrict__ vI, long ldv, const float *__restrict__ sW,
                                                              const float *__restrict__ sW2, float *__restrict__ h_out, long ldh, int T,
                                                              int B, int n, int reverse, const int *__restrict__ lens)
{
    static_assert(N % 16 == 0 && N <= 144, "one wave per 16 neurons, at most nine waves");
    // KBS K blocks of 32 as MFMA operands; HALF: one more of 16 (the image keeps a whole block for it)
    constexpr bool HALF = N == 144;                       // (112 = 3 x 32 + 16 as well, but measured 658 us against 628 as four blocks of 32)
    constexpr int NW = N / 16, KBS = HALF ? N / 32 : (N + 31) / 32, KBI = KBS + (HALF ? 1 : 0), NTH = 64 * NW, KP = 32 * KBI;

    // state images: [hi image | lo image], each KP x 4 chunks halves; element (k block kb, k group g, chunk c, r) = dword
    // ((kb*4+g)*4+c)*4 + r holds neuron 32kb+4g+r (low half) and 32kb+16+4g+r (high half) of chunk c
    // the lo image 32 banks behind the hi image (back to back on the same banks the mixed-operand reads conflict two ways: gru_bar16.hip)
    constexpr int LO = 2 * KP + (2 * KP % 64 == 32 ? 0 : 32);
    __shared__ __attribute__((aligned(16))) unsigned h_img[LO + 2 * KP], rh_img[LO + 2 * KP];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < LO + 2 * KP; i += NTH) { h_img[i] = 0u; rh_img[i] = 0u; }                // h(-1) = 0
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // A operands: row = neuron 16w + (lane & 15), element (kb, e) = W[row][k] with k = 32kb + 16(e&1) + 4g + (e>>1); rows scaled to [1, 2)
    half8 wz_hi[KBS], wz_lo[KBS], wr_hi[KBS], wr_lo[KBS], wc_hi[KBS], wc_lo[KBS];
    half4 wzh_hi = {}, wzh_lo = {}, wrh_hi = {}, wrh_lo = {}, wch_hi = {}, wch_lo = {};      // HALF: k = 32 KBS + 4 g + i
    float inv_z, inv_r, inv_c;
    {
        const int unit = 16 * w + (lane & 15);
        const bool uk = unit < n;
        auto prep = [&](const float *row, half8 *wh, half8 *wl, half4 &whh, half4 &wlh, float &invq) {
            float v[KBS][8];
            float vh[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            float m = 0.0f;
            if constexpr (HALF) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int k = 32 * KBS + 4 * g + e;
                    vh[e] = (uk && k < n) ? row[k] : 0.0f;
                    m = fmaxf(m, fabsf(vh[e]));
                }
            }
#pragma unroll
            for (int kb = 0; kb < KBS; kb++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = 32 * kb + 16 * (e & 1) + 4 * g + (e >> 1);
                    v[kb][e] = (uk && k < n) ? row[k] : 0.0f;
                    m = fmaxf(m, fabsf(v[kb][e]));
                }
            float iv;
            const float sc = pow2_scale(kgroup_max(m), iv);
            invq = __shfl(iv, 4 * g + q);
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
                half8 hi, lo;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float a = v[kb][e] * sc;
                    const _Float16 hh = (_Float16)a;
                    hi[e] = hh;
                    lo[e] = (_Float16)(a - (float)hh);
                }
                wh[kb] = hi;
                wl[kb] = lo;
            }
            if constexpr (HALF) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float a = vh[e] * sc;
                    const _Float16 hh = (_Float16)a;
                    whh[e] = hh;
                    wlh[e] = (_Float16)(a - (float)hh);
                }
            }
        };
        const int ur = uk ? unit : 0;
        prep(sW + (size_t)ur * n, wz_hi, wz_lo, wzh_hi, wzh_lo, inv_z);
        prep(sW + (size_t)(n + ur) * n, wr_hi, wr_lo, wrh_hi, wrh_lo, inv_r);
        prep(sW2 + (size_t)ur * n, wc_hi, wc_lo, wch_hi, wch_lo, inv_c);
    }
    int moff[KBS];
#pragma unroll
    for (int kb = 0; kb < KBS; kb++) moff[kb] = (q >> 1) * LO + ((kb * 4 + g) * 4 + c) * 4;      // my column group's image, in dwords
    const int moffh = (q >> 1) * LO + ((KBS * 4 + g) * 4 + c) * 4;                                // HALF: the block of 16
    auto ldH4 = [](const unsigned *img, int off) { return *reinterpret_cast<const half4 *>(img + off); };
    // my neuron u = 16w + 4g + q -> K block w >> 1, half w & 1, k group g, r = q
    const int u0 = 16 * w + 4 * g + q;
    const bool uok = u0 < n;
    // in halves; the wave of the block of 16 (HALF, w = 2 KBS) writes its four r side by side: one 8-byte read is the K = 16 operand
    const int wpos = (HALF && w == 2 * KBS) ? ((((w >> 1) * 4 + g) * 4 + c) * 4) * 2 + q
                                            : ((((w >> 1) * 4 + g) * 4 + c) * 4 + q) * 2 + (w & 1);
    // my chunk's rows (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
    const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
    float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldh + (uok ? u0 : 0);
    // vI of a step: z | r | c blocks of n floats per row; requests run three steps ahead in four register sets (gru_scan16.hip);
    // steps past the chunk's end re-read its last row
    struct VI { float z, r, c; };
    VI vs[4];
    const float *sb_z = vI, *sb_r = vI + n, *sb_c = vI + 2 * n;
    unsigned voff = (unsigned)((((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldv + (uok ? u0 : 0)) * sizeof(float));
    const unsigned vstep = (unsigned)((size_t)B * ldv * sizeof(float));
    int vnext = 0;
    auto load_vi = [&](VI &v) {
        g1_gload(v.z, voff, sb_z);
        g1_gload(v.r, voff, sb_r);
        g1_gload(v.c, voff, sb_c);
        vnext++;
        if (vnext < Tc) voff = reverse ? voff - vstep : voff + vstep;
    };
    load_vi(vs[0]);
    load_vi(vs[1]);
    load_vi(vs[2]);
    __syncthreads();                                     // LDS initialised

    float hold = 0.0f;
    auto step = [&](auto PHC, const int s) {
        constexpr int ph = decltype(PHC)::value;
        VI &cur = vs[ph];
        // ---- barrier A: h(s-1) is in its image ----
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bh[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) bh[kb] = ldH(h_img, moff[kb]);
        half4 bhh = {};
        if constexpr (HALF) bhh = ldH4(h_img, moffh);
        load_vi(vs[(ph + 3) & 3]);
        f32x4 accR, accZ;
        static_for<0, KBS>([&](auto KC) {
            constexpr int kb = decltype(KC)::value;
            g1_mma<kb == 0>(accR, wr_lo[kb], bh[kb]);
            g1_mma<kb == 0>(accZ, wz_lo[kb], bh[kb]);
            g1_mma<false>(accR, wr_hi[kb], bh[kb]);
            g1_mma<false>(accZ, wz_hi[kb], bh[kb]);
        });
        if constexpr (HALF) {
            // an MFMA of another shape reading an accumulator is a READER like any vector instruction: seven wait states behind the
            // 16x16x32 that wrote it (tools/mfma_result_hazard_scan.py; one MFMA in between is four)
            asm volatile("s_nop 2" : "+v"(accR), "+v"(accZ));
            g1_mma16(accR, wrh_lo, bhh);
            g1_mma16(accZ, wzh_lo, bhh);
            g1_mma16(accR, wrh_hi, bhh);
            g1_mma16(accZ, wzh_hi, bhh);
        }
        asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // this step's vI: the loads of the three younger steps may be outstanding
        asm volatile("" : "+v"(cur.z), "+v"(cur.r), "+v"(cur.c));
        mfma_drain2(accR, accZ);                         // pick_mix reads the accumulators from asm
        // layers.py:1012-1016
        const float rg = sigmoid4(fmaf(pick_mix(accR), inv_r, cur.r));
        const float rh = uok ? rg * hold : 0.0f;
        {
            float hv = rh;
            asm volatile("" : "+v"(hv));                 // split2's note on v_fma_mixlo_f16 applies
            const _Float16 h16 = (_Float16)hv;
            const _Float16 l16 = (_Float16)(hv - (float)h16);
            reinterpret_cast<unsigned short *>(&rh_img[0])[wpos] = __builtin_bit_cast(unsigned short, h16);
            reinterpret_cast<unsigned short *>(&rh_img[LO])[wpos] = __builtin_bit_cast(unsigned short, l16);
        }
        const float zg = sigmoid4(fmaf(pick_mix(accZ), inv_z, cur.z));
        // ---- barrier B: r * h(s-1) is in its image ----
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bm[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) bm[kb] = ldH(rh_img, moff[kb]);
        half4 bmh = {};
        if constexpr (HALF) bmh = ldH4(rh_img, moffh);
        f32x4 accC;
        static_for<0, KBS>([&](auto KC) {
            constexpr int kb = decltype(KC)::value;
            g1_mma<kb == 0>(accC, wc_lo[kb], bm[kb]);
            g1_mma<false>(accC, wc_hi[kb], bm[kb]);
        });
        f32x4 accCh;
        if constexpr (HALF) {
            // its own accumulator (the candidate has no second gate to put between the two shapes), added behind the drain
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, 0" : "=&v"(accCh) : "v"(wch_lo), "v"(bmh));
            g1_mma16(accCh, wch_hi, bmh);
        }
        mfma_drain(accC);
        if constexpr (HALF) {
            asm volatile("" : "+v"(accCh));
            accC += accCh;
        }
        // layers.py:1017-1021
        const float hb = tanh5(fmaf(pick_mix(accC), inv_c, cur.c));
        const float hn = uok ? fmaf(1.0f - zg, hb, zg * hold) : 0.0f;
        hold = hn;
        {
            float hv = hn;
            asm volatile("" : "+v"(hv));
            const _Float16 h16 = (_Float16)hv;
            const _Float16 l16 = (_Float16)(hv - (float)h16);
            reinterpret_cast<unsigned short *>(&h_img[0])[wpos] = __builtin_bit_cast(unsigned short, h16);
            reinterpret_cast<unsigned short *>(&h_img[LO])[wpos] = __builtin_bit_cast(unsigned short, l16);
        }
        if (live && s < Tc && uok) hp[0] = hn;
        hp += hstep;
    };
    for (int s = 0; s < T; s += 4) {
        step(ic<0>{}, s);
        if (s + 1 < T) step(ic<1>{}, s + 1);
        if (s + 2 < T) step(ic<2>{}, s + 2);
        if (s + 3 < T) step(ic<3>{}, s + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of mine may land in registers after the wave has ended
}

template <int N>
static size_t scan1t_exclusive_lds()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(gru_scan1t_kernel<N>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(gru_scan1t_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

// called by slk_gru_scan16_f32 (gru_scan16.hip) for the sizes this plan is the faster one for; same contract
extern "C" int slk_gru_scan1t_launch(const float *vI, long ldv, const float *sW, const float *sW2, float *y, long ldy, int T, int B, int n,
                                     int reverse, const int32_t *lens, hipStream_t s)
{
    if (n % 16 || n > 144) return SLK_ERR_UNSUPPORTED;
#define G1_LAUNCH(NN)                                                                                                        \
    {                                                                                                                        \
        const size_t dyn = SLK_PER_DEVICE(size_t, scan1t_exclusive_lds<NN>());                                               \
        hipLaunchKernelGGL((gru_scan1t_kernel<NN>), dim3((B + 3) / 4), dim3(4 * NN), dyn, s, vI, ldv, sW, sW2, y, ldy, T, B, n,  \
                           reverse & 1, lens);                                                                               \
        return slk_launch_status();                                                                                          \
    }
    if (n <= 112) G1_LAUNCH(112)
    if (n <= 128) G1_LAUNCH(128)
    if (n != 144) return SLK_ERR_UNSUPPORTED;              // (the block of 16 of <144> is neurons 128 .. 143)
    G1_LAUNCH(144)
#undef G1_LAUNCH
}
