// lstm_bwd16.hip -- the reverse scan of an Lstm layer (the training step, csrc/train.hip has the maths and the portable kernel
// lstm_backward_kernel) on the execution plan of lstm_scan16.hip: four waves per workgroup, four chunks per workgroup, one
// s_barrier per step, the single matrix product of a step
//
//     carry_out[u] = sum over the 4n gate rows r of dsum[r] * sW[r][u]            (what flows back into out_{t-1})
//
// as fp16-split MFMAs (v_mfma_f32_16x16x32_f16, two per product: the hi and lo halves of dsum in different column groups,
// bar16_common.h) with the TRANSPOSED weights as A operands in registers -- wave w owns output units 16w .. 16w+15, K runs over the
// 4n gate rows -- and dsum of the step exchanged through a packed, double-buffered LDS image.  A lane owns one (unit, chunk) pair:
// it forms go = dL/dout_t + carry_out, the four gate gradients (dg', di', df', do'), the cell carry and its share of the peephole
// gradients, stores dsum to HBM and writes its four halves of the image.  Gradients have no natural range (the state of the forward
// scan lives in [-1, 1]; these may be 1e-9 or 1e+3), so the image of a step is scaled per chunk by the power of two that brings that
// chunk's largest |dsum| into [1, 2) -- what the weight rows get once, the operand columns get every step: a max over the wave's
// lanes of a chunk (two DPP steps and the k-group swap), four floats per wave through LDS and a second barrier.  Per step a lane reads dy, its four activated gates and two
// cell states; they are requested three steps ahead with asm loads the kernel counts itself (gru_scan16.hip).
#include <limits.h>

#include "bar16_common.h"

__device__ __forceinline__ void lb_gload1(float &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void lb_gload4(f32x4 &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

template <int N>
__global__ void __launch_bounds__(256, 1) lstm_bwd16_kernel(const float *__restrict__ dy, long lddy, const float *__restrict__ gates,
                                                            const float *__restrict__ cell, const float *__restrict__ sW,
                                                            const float *__restrict__ peep, float *__restrict__ dsum,
                                                            float *__restrict__ dpeep, int T, int B, int n, int reverse)
{
    static_assert(N == 64, "four waves of 16 units");
    constexpr int KBS = 4 * N / 32;                      // K = the 4N gate rows

    // [image parity][hi image | lo image], each 4 chunks x 4N halves = 8N dwords... element (k block kb, k group g, chunk c, r) =
    // dword ((kb*4+g)*4+c)*4 + r holds gate rows 32kb+4g+r (low half) and 32kb+16+4g+r (high half) of chunk c
    __shared__ __attribute__((aligned(16))) unsigned d_img[2][2 * 8 * N];
    __shared__ float smax[4][4];                         // [wave][chunk]: largest |dsum| of the step among the wave's units

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < 2 * 8 * N; i += 256) { d_img[0][i] = 0u; d_img[1][i] = 0u; }
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // A operands: row = output unit 16w + (lane & 15), element (kb, j) = sW[k][unit] with k = 32kb + 16(j&1) + 4g + (j>>1); rows scaled
    half8 w_hi[KBS], w_lo[KBS];
    float inv;
    {
        const int unit = 16 * w + (lane & 15);
        const bool uk = unit < n;
        float v[KBS][8];
        float m = 0.0f;
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int k = 32 * kb + 16 * (j & 1) + 4 * g + (j >> 1);
                v[kb][j] = (uk && k < 4 * n) ? sW[(size_t)k * n + unit] : 0.0f;
                m = fmaxf(m, fabsf(v[kb][j]));
            }
        }
        float iv;
        const float sc = pow2_scale(kgroup_max(m), iv);
        inv = __shfl(iv, 4 * g + q);
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float a = v[kb][j] * sc;
                const _Float16 h = (_Float16)a;
                w_hi[kb][j] = h;
                w_lo[kb][j] = (_Float16)(a - (float)h);
            }
        }
    }
    int moff[KBS];
#pragma unroll
    for (int kb = 0; kb < KBS; kb++) moff[kb] = (q >> 1) * 8 * N + ((kb * 4 + g) * 4 + c) * 4;    // in dwords, my column group's image
    // my (unit, chunk): gate rows 4*u0 .. 4*u0+3 -> K block u0 >> 3, half (u0 >> 2) & 1, k group u0 & 3, r = gate
    const int u0 = 16 * w + 4 * g + q;
    const bool uok = u0 < n;
    const int wbase = ((((u0 >> 3) * 4 + (u0 & 3)) * 4 + c) * 4) * 2 + ((u0 >> 2) & 1);         // in halves: + 2 * gate
    const float p0 = (peep && uok) ? peep[u0] : 0.0f, p1 = (peep && uok) ? peep[n + u0] : 0.0f, p2 = (peep && uok) ? peep[2 * n + u0] : 0.0f;

    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    // rows of scan step s: (reverse ? T-1-s : s) * B + chunk; the pass walks s = T-1 .. 0
    const long rstep = reverse ? (long)B : -(long)B;     // rows from scan step s to s - 1
    const long row0 = (long)(reverse ? 0 : T - 1) * B + bcc;
    const int uu = uok ? u0 : 0;
    // operands of a step: dy, the four activated gates (g, i, f, o), the cell state after and before the step.  Four register sets,
    // three steps ahead; four loads per step: once at most 12 memory operations are outstanding the current step's have arrived.
    struct Ops { float go, cn, cp; f32x4 gt; };
    Ops vs[4];
    unsigned off_dy = (unsigned)((row0 * lddy + uu) * (long)sizeof(float));
    unsigned off_gt = (unsigned)((row0 * 4L * n + 4 * uu) * (long)sizeof(float));
    unsigned off_cn = (unsigned)((row0 * (long)n + uu) * (long)sizeof(float));
    const unsigned st_dy = (unsigned)(rstep * lddy * (long)sizeof(float)), st_gt = (unsigned)(rstep * 4L * n * (long)sizeof(float));
    const unsigned st_cn = (unsigned)(rstep * (long)n * (long)sizeof(float));
    int vnext = 0;                                       // requests issued so far (request i is scan step T-1-i)
    auto load_v = [&](Ops &v) {
        lb_gload1(v.go, off_dy, dy);
        lb_gload4(v.gt, off_gt, gates);
        lb_gload1(v.cn, off_cn, cell);
        const bool more = vnext + 1 < T;                 // a step before this one exists: its cell state is my c_{t-1}
        lb_gload1(v.cp, more ? off_cn + st_cn : off_cn, cell);
        vnext++;
        if (more) { off_dy += st_dy; off_gt += st_gt; off_cn += st_cn; }
    };
    load_v(vs[0]);
    load_v(vs[1]);
    load_v(vs[2]);
    float *dp = dsum + (row0 * 4L * n + 4 * uu);
    const long dstep = rstep * 4L * n;

    float carry_c = 0.0f, ap0 = 0.0f, ap1 = 0.0f, ap2 = 0.0f;
    float inv_s = 1.0f;                                  // inverse of the scale my chunk's column of the image carries
    auto step = [&](auto PHC, const int i) {             // i-th step of the pass: scan step s = T-1-i
        constexpr int ph = decltype(PHC)::value;
        constexpr int par = ph & 1;                      // dsum of the step before (scan step s+1) is in image `par`
        Ops &cur = vs[ph];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bm[KBS];
#pragma unroll
        for (int kb = 0; kb < KBS; kb++) bm[kb] = ldH(d_img[par], moff[kb]);
        load_v(vs[(ph + 3) & 3]);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two chains over alternating K blocks
#pragma unroll
        for (int kb = 0; kb < KBS; kb += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[kb], bm[kb], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[kb + 1], bm[kb + 1], acc1, 0, 0, 0);
            if (kb == 0) asm volatile("" : "+v"(acc0), "+v"(acc1) : "v"(w_lo[0]), "v"(w_lo[1]), "v"(bm[0]), "v"(bm[1]));   // gemm_rows_f16x3.hip
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[kb], bm[kb], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[kb + 1], bm[kb + 1], acc1, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        asm volatile("" : "+v"(cur.go), "+v"(cur.cn), "+v"(cur.cp), "+v"(cur.gt));
        mfma_drain2(acc0, acc1);                         // pick_mix reads the accumulators from asm
        const float carry_out = (pick_mix(acc0) + pick_mix(acc1)) * inv * inv_s;
        // csrc/train.hip, lstm_backward_kernel
        const float go = cur.go + carry_out, gg = cur.gt[0], ig = cur.gt[1], fg = cur.gt[2], og = cur.gt[3], cn = cur.cn;
        const float cp = (i + 1 < T) ? cur.cp : 0.0f;
        const float tc = tanh5(cn);
        const float do_pre = go * tc * og * (1.0f - og);
        const float dc = go * og * (1.0f - tc * tc) + do_pre * p2 + carry_c;
        const float di_pre = dc * gg * ig * (1.0f - ig);
        const float df_pre = dc * cp * fg * (1.0f - fg);
        const float dg_pre = dc * ig * (1.0f - gg * gg);
        f32x4 d4 = {dg_pre, di_pre, df_pre, do_pre};
        if (!uok) d4 = f32x4{0.f, 0.f, 0.f, 0.f};
        carry_c = uok ? dc * fg + di_pre * p0 + df_pre * p1 : 0.0f;
        ap0 += d4[1] * cp; ap1 += d4[2] * cp; ap2 += d4[3] * cn;
        if (live && uok) *reinterpret_cast<f32x4 *>(dp) = d4;
        dp += dstep;
        // this step's column scale: max |dsum| over the units of my chunk -- over the k groups (lanes 16 apart), the lane quartets of a
        // row (DPP), then the four waves
        float mx = fmaxf(fmaxf(fabsf(d4[0]), fabsf(d4[1])), fmaxf(fabsf(d4[2]), fabsf(d4[3])));
        mx = kgroup_max(mx);
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x124, 0xf, 0xf, false)));      // row_ror:4
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x128, 0xf, 0xf, false)));      // row_ror:8
        if (lane < 4) smax[w][c] = mx;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float cm = fmaxf(fmaxf(smax[0][c], smax[1][c]), fmaxf(smax[2][c], smax[3][c]));
        const float sc = pow2_scale(cm, inv_s);
        {
            unsigned short *ih = reinterpret_cast<unsigned short *>(&d_img[par ^ 1][0]) + wbase;
            unsigned short *il = reinterpret_cast<unsigned short *>(&d_img[par ^ 1][8 * N]) + wbase;
#pragma unroll
            for (int gt = 0; gt < 4; gt++) {
                float hv = d4[gt] * sc;
                asm volatile("" : "+v"(hv));             // split2's note on v_fma_mixlo_f16 applies
                const _Float16 hh = (_Float16)hv;
                const _Float16 hl = (_Float16)(hv - (float)hh);
                ih[2 * gt] = __builtin_bit_cast(unsigned short, hh);
                il[2 * gt] = __builtin_bit_cast(unsigned short, hl);
            }
        }
    };
    __syncthreads();                                     // LDS initialised
    for (int i = 0; i < T; i += 4) {
        step(ic<0>{}, i);
        if (i + 1 < T) step(ic<1>{}, i + 1);
        if (i + 2 < T) step(ic<2>{}, i + 2);
        if (i + 3 < T) step(ic<3>{}, i + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of mine may land in registers after the wave has ended
    if (live && uok) {
        dpeep[((size_t)bc * 3 + 0) * n + u0] = ap0;
        dpeep[((size_t)bc * 3 + 1) * n + u0] = ap1;
        dpeep[((size_t)bc * 3 + 2) * n + u0] = ap2;
    }
}

static size_t lstm_bwd16_exclusive_lds()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(lstm_bwd16_kernel<64>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_bwd16_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

// include/sloika_amd.h
extern "C" int slk_lstm_backward16_f32(const float *dy, long lddy, const float *gates, const float *cell, const float *sW,
                                       const float *peep, float *dsum, float *dpeep, int T, int B, int n, int reverse, int act,
                                       int gate_act, slk_stream_t stream)
{
    if (!dy || !gates || !cell || !sW || !dsum || !dpeep || T < 1 || B < 1 || n < 1 || lddy < n) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if (n % 16 || n > 64) return SLK_ERR_UNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(gates) | reinterpret_cast<uintptr_t>(dsum)) & 15) != 0) return SLK_ERR_UNSUPPORTED;
    if ((unsigned long long)T * B * 4 * n * sizeof(float) >= (1ull << 32) || (unsigned long long)T * B * lddy * sizeof(float) >= (1ull << 32))
        return SLK_ERR_UNSUPPORTED;                      // 32-bit lane offsets
    const size_t dyn = SLK_PER_DEVICE(size_t, lstm_bwd16_exclusive_lds());
    hipLaunchKernelGGL((lstm_bwd16_kernel<64>), dim3((B + 3) / 4), dim3(256), dyn, slk_stream(stream), dy, lddy, gates, cell, sW, peep,
                       dsum, dpeep, T, B, n, reverse & 1);
    return slk_launch_status();
}
