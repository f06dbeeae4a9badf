// lstm_scan16.hip -- the scan of Lstm.step (sloika/layers.py:677-691, peepholes, interleaved gate rows) on the execution plan of
// gru_scan16.hip: four waves per workgroup, one per SIMD, four chunks per workgroup, the recurrent products as 3-term fp16 splits
// on v_mfma_f32_16x16x32_f16 with rows scaled by powers of two, the state exchanged as packed hi / lo halves through LDS, a lane
// owning one (unit, chunk) pair.  The Lstm step has ONE matrix product (h(s-1) against the 4n x n matrix sW) and one exchange, so
// with the state image double buffered a step is one s_barrier:
//
//     barrier -> fetch h(s-1) (two 32-wide K blocks, hi and lo) -> 4 gates x 2 blocks x 3 terms = 24 MFMAs on four independent
//     accumulators -> gate arithmetic of my (unit, chunk): five transcendentals -> my half of a state dword into the OTHER image,
//     the output row to HBM
//
// Wave w owns units 16w .. 16w+15 (sizes below 64, a multiple of 16, run with zero weights for the missing units: their state
// stays 0).  The input projection vW = x.iW^T + b comes from HBM (the row GEMM writes it): one 16-byte load per lane and step --
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

template <int N>
__global__ void __launch_bounds__(256, 1) lstm_scan16_kernel(const float *__restrict__ vW, const float *__restrict__ sW,
                                                             const float *__restrict__ peep, float *__restrict__ h_out, long ldh, int T, int B,
                                                             int n, int reverse, const int *__restrict__ lens)
{
    static_assert(N == 64, "four waves of 16 units");
    constexpr int KBS = N / 32;

    // [image parity][2N dwords]: element (k block kb, k group g, chunk c, r) = dword ((kb*4+g)*4+c)*4 + r holds units 32kb+4g+r (low
    // half) and 32kb+16+4g+r (high half) of chunk c -- the order in which a lane's eight B-operand halves are consecutive
    __shared__ __attribute__((aligned(16))) unsigned h_hi[2][2 * N], h_lo[2][2 * N];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < 2 * N; i += 256) { h_hi[0][i] = 0u; h_lo[0][i] = 0u; h_hi[1][i] = 0u; h_lo[1][i] = 0u; }       // h(-1) = 0
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // recurrent weights: A operands of the four gate tiles of my 16 units (row = 4*unit + gate), rows scaled to [1, 2)
    half8 w_hi[4][KBS], w_lo[4][KBS];
    float inv[4];
    {
        const int unit = 16 * w + (lane & 15);
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
            inv[gt] = __shfl(iv, 4 * g + q);
#pragma unroll
            for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float a = v[kb][j] * sc;
                    const _Float16 h = (_Float16)a;
                    w_hi[gt][kb][j] = h;
                    w_lo[gt][kb][j] = (_Float16)(a - (float)h);
                }
            }
        }
    }
    int boff[KBS];
#pragma unroll
    for (int kb = 0; kb < KBS; kb++) boff[kb] = ((kb * 4 + g) * 4 + c) * 4;                 // in dwords
    // my (unit, chunk): unit u0 lives in dword wdw, half (w & 1)
    const int u0 = 16 * w + 4 * g + q;
    const bool uok = u0 < n;
    const int wdw = (((w >> 1) * 4 + g) * 4 + c) * 4 + q;
    unsigned short *my_hi[2], *my_lo[2];
#pragma unroll
    for (int par = 0; par < 2; par++) {
        my_hi[par] = reinterpret_cast<unsigned short *>(&h_hi[par][wdw]) + (w & 1);
        my_lo[par] = reinterpret_cast<unsigned short *>(&h_lo[par][wdw]) + (w & 1);
    }
    const float p0 = (peep && uok) ? peep[u0] : 0.0f, p1 = (peep && uok) ? peep[n + u0] : 0.0f, p2 = (peep && uok) ? peep[2 * n + u0] : 0.0f;

    // my chunk's rows (ragged batch: chunk bc is Tc <= T steps long; a reversed scan starts at ITS last step)
    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    const int Tc = (lens && live) ? min(max(lens[bc], 1), T) : T;
    const long hstep = (reverse ? -1L : 1L) * (long)B * ldh;
    float *hp = h_out + ((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldh + (uok ? u0 : 0);
    // vW of step s for my unit: four consecutive floats; steps past the chunk's end re-read its last row (never stored).  Four
    // register sets, step s uses set s % 4 and requests step s + 3 into the set step s - 1 used; loads complete in order among
    // themselves, so once at most 3 memory operations are outstanding the current step's has arrived, whatever the stores do.
    f32x4 vs[4];
    const long ldv = 4L * n;
    unsigned voff = (unsigned)((((size_t)(reverse ? Tc - 1 : 0) * B + bcc) * ldv + 4 * (uok ? u0 : 0)) * sizeof(float));
    const unsigned vstep = (unsigned)((size_t)B * ldv * sizeof(float));
    int vnext = 0;
    auto load_v = [&](f32x4 &v) {
        lstm_gload4(v, voff, vW);
        vnext++;
        if (vnext < Tc) voff = reverse ? voff - vstep : voff + vstep;
    };
    load_v(vs[0]);
    load_v(vs[1]);
    load_v(vs[2]);

    float cell = 0.0f;
    auto step = [&](auto PHC, const int s) {
        constexpr int ph = decltype(PHC)::value;
        constexpr int par = ph & 1;                      // h(s-1) is in image `par`, h(s) goes to the other one
        f32x4 &cur = vs[ph];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bh[KBS], bl[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) { bh[kb] = ldH(h_hi[par], boff[kb]); bl[kb] = ldH(h_lo[par], boff[kb]); }
        load_v(vs[(ph + 3) & 3]);                        // three steps ahead
        f32x4 acc[4];
#pragma unroll
        for (int gt = 0; gt < 4; gt++) acc[gt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) {
            // small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[gt][kb], bl[kb], acc[gt], 0, 0, 0);
                if (kb == 0) asm volatile("" : "+v"(acc[gt]) : "v"(w_hi[gt][0]), "v"(w_lo[gt][0]), "v"(bl[0]), "v"(bh[0]));   // gemm_rows_f16x3.hip
            }
#pragma unroll
            for (int gt = 0; gt < 4; gt++) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[gt][kb], bh[kb], acc[gt], 0, 0, 0);
#pragma unroll
            for (int gt = 0; gt < 4; gt++) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[gt][kb], bh[kb], acc[gt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                 // this step's vW (see above)
        lstm_pin4(cur);
        // layers.py:686-691
        const float a0 = fmaf(sel4(acc[0], q), inv[0], cur[0]), a1 = fmaf(sel4(acc[1], q), inv[1], cur[1]);
        const float a2 = fmaf(sel4(acc[2], q), inv[2], cur[2]), a3 = fmaf(sel4(acc[3], q), inv[3], cur[3]);
        const float fg = sigmoid4(fmaf(cell, p1, a2));                   // forget gate
        const float ig = sigmoid4(fmaf(cell, p0, a1));                   // input gate
        const float cn = uok ? fmaf(fg, cell, tanh5(a0) * ig) : 0.0f;    // new cell state
        const float og = sigmoid4(fmaf(cn, p2, a3));                     // output gate peeps at the NEW state
        const float hn = uok ? tanh5(cn) * og : 0.0f;
        cell = cn;
        {
            float hv = hn;
            asm volatile("" : "+v"(hv));                 // split2's note on v_fma_mixlo_f16 applies
            const _Float16 hh = (_Float16)hv;
            const _Float16 hl = (_Float16)(hv - (float)hh);
            *my_hi[par ^ 1] = __builtin_bit_cast(unsigned short, hh);
            *my_lo[par ^ 1] = __builtin_bit_cast(unsigned short, hl);
        }
        if (live && s < Tc && uok) hp[0] = hn;
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
static size_t lstm_scan16_exclusive_lds()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(lstm_scan16_kernel<64>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_scan16_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
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
    if (n % 16 || n > 64) return SLK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(vW) & 15) != 0) return SLK_ERR_UNSUPPORTED;
    if ((unsigned long long)T * B * 4 * n * sizeof(float) >= (1ull << 32)) return SLK_ERR_UNSUPPORTED;       // 32-bit lane offsets
    const size_t dyn = SLK_PER_DEVICE(size_t, lstm_scan16_exclusive_lds());
    hipLaunchKernelGGL((lstm_scan16_kernel<64>), dim3((B + 3) / 4), dim3(256), dyn, slk_stream(stream), vW, sW, p, out, ldo, T, B, n,
                       reverse & 1, lens);
    return slk_launch_status();
}
