// gru_bwd16.hip -- the reverse scan of a Gru layer (the training step; csrc/train.hip has the maths and the portable kernel,
// gru_backward_mfma.hip the fp32-MFMA one) on the execution plan of the forward kernels: four chunks per workgroup, two s_barrier
// per step, the two dependent products of a step
//
//     drh[i]    = sum_k dac[k] sW2[k][i]                      (n -> n;   dac = g (1-z) (1-c^2))
//     carry[i] += sum_k [daz | dar][k] sW[k][i]               (2n -> n;  daz = g (h-c) z (1-z), dar = drh h r (1-r))
//
// as fp16-split MFMAs (v_mfma_f32_16x16x32_f16, two per product: the hi and lo halves of the operand in different column groups,
// bar16_common.h) with the TRANSPOSED weights as A operands in registers.  The scan is a latency chain (its time does not depend on
// the batch: 1.10 ms at 64 chunks, 1.18 at 1024 for the fp32 kernel), so the step is kept short rather than the waves few: ONE
// 16-unit tile per wave, n / 16 waves per workgroup (six at n = 96: two SIMDs carry two waves, 256 registers each -- a wave needs
// 72 for its weights).  A lane owns one (unit, chunk) pair: it forms g = dL/dh_t + carry, recovers the candidate from the layer's
// own output, stores da = [daz | dar | dac] and r * h, and writes its halves of the two operand images in LDS.
//
// Gradients have no natural range (the forward state lives in [-1, 1]; these may be 1e-9 or 1e+3), so every image is scaled by a
// power of two per chunk and step.  lstm_bwd16.hip spends a barrier on the exact maximum; with two products that would be four
// barriers per step, so here the scale comes from a BOUND that every wave can form from ONE number per chunk exchanged at a barrier
// the step has anyway -- |dac|, 2 |daz| <= |g| <= Gb := 2 max_u (|dy_t + keep| + C1 |dzr_{t+1}|) and 4 |dar| <= |drh| <= C2 Gb with
// C1, C2 the largest column sums of |sW|, |sW2| -- and maps the bound to 2^14 (fp16 reaches 2^16): rigorous against overflow.  It is
// loose by a handful of bits (worst-case column sums where the products average out); what that costs is the range over which the
// lo half stays a normal number, i.e. an absolute error of 2^-25 of the BOUND on the smallest entries -- 2^-39+L relative to the
// largest entry for a bound L bits loose: float32-grade up to L = 15.
//
// Per step a lane reads dy, z, r, h_t, h_prev of its unit: asm loads the kernel counts itself (gru_scan16.hip), four steps ahead.
#include <limits.h>

#include "bar16_common.h"

__device__ __forceinline__ void gw_gload1(float &dst, unsigned voff, const float *sbase)
{
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// asm MFMA: the first one of a sum takes a zero C operand and an early-clobber destination (tools/mfma_overlap_scan.py)
template <bool FIRST>
__device__ __forceinline__ void gw_mma(f32x4 &acc, const half8 &wa, const half8 &bm)
{
    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(wa), "v"(bm));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(bm));
}

// power of two s with bound * s in [2^13, 2^14) (exponent kept inside [27, 240]); inv = 1 / s
__device__ __forceinline__ float gw_pow2_top(float bound, float &inv)
{
    const int e = min(max((int)((__float_as_uint(bound) >> 23) & 0xff), 27), 240);
    inv = __uint_as_float((unsigned)(e - 13) << 23);
    return __uint_as_float((unsigned)(267 - e) << 23);
}

// maximum over the lanes of my chunk (same lane & 3): over the k groups (lanes 16 apart), then over the lane quartets of a row
__device__ __forceinline__ float gw_chunk_max(float mx)
{
    mx = kgroup_max(mx);
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x124, 0xf, 0xf, false)));      // row_ror:4
    mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x128, 0xf, 0xf, false)));      // row_ror:8
    return mx;
}

#ifndef GW_ABL_SKIP
#define GW_ABL_SKIP 0                                    // timing experiments (results are garbage): waves >= this skip their dx MFMAs
#endif
#ifndef GW_ABL
#define GW_ABL 0                                         // timing experiments only (results are garbage): 1 no stores, 2 no loads, 4 fixed scales
#endif

// DX: the layer's dL/dx = da . iW (rows [daz | dar | dac] of da against iW[3n][insize]) comes out of the same pass: the two operand images
// of a step ARE da of that step as fp16 hi/lo pairs, so the product costs a wave the MFMAs of its own 16 inputs (transposed iW as A
// operands, 2 x 9 K blocks at n = 96) and nothing else -- no second pass over da (944 MB per layer read back by a separate GEMM).
// DA (with DX): the product leaves multiplied by fun'(.) of the layer BELOW, evaluated on that layer's output yref (the Gru's own input):
// it is dL/d(pre-activation) of that layer, what slk_gemm_dact_bf16x6 hands down (one more load per lane and step).
__device__ __forceinline__ float gw_dact(float v, int act)
{
    switch (act) {                                       // activation.py:8-57, as act_backward_kernel (csrc/train.hip)
    case SLK_ACT_TANH: return 1.0f - v * v;
    case SLK_ACT_SIGMOID: return v * (1.0f - v);
    case SLK_ACT_RELU: return v > 0.0f ? 1.0f : 0.0f;
    case SLK_ACT_ELU: return v > 0.0f ? 1.0f : v + 1.0f;
    default: return 1.0f;
    }
}

template <int N, bool DX, bool DA>
__global__ void __launch_bounds__(4 * N, 1) gru_bwd16_kernel(const float *__restrict__ dy, long lddy, const float *__restrict__ hprev,
                                                             long ldhp, const float *__restrict__ zr, const float *__restrict__ hout,
                                                             long ldh, const float *__restrict__ sW, const float *__restrict__ sW2,
                                                             float *__restrict__ da, float *__restrict__ rh, int T, int B, int n,
                                                             int reverse, const float *__restrict__ iW, float *__restrict__ dx,
                                                             long lddx, int insize, const float *__restrict__ yref, long ldyref,
                                                             int dact)
{
    static_assert(DX || !DA, "the activation's derivative rides on the dx product");
    static_assert(N % 32 == 0 && N <= 128, "K blocks of 32; one wave per 16 units, at most eight waves");
    constexpr int NW = N / 16, KB1 = N / 32, KB2 = 2 * N / 32, NTH = 64 * NW;
    constexpr int LPS = DA ? 6 : 5;                      // loads per step and lane

    // operand images: [hi image | lo image], each K x 4 chunks halves; element (k block kb, k group g, chunk c, r) = dword
    // ((kb*4+g)*4+c)*4 + r holds k = 32kb+4g+r (low half) and 32kb+16+4g+r (high half) of chunk c
    // (the lo image 32 banks behind the hi image, as in gru_bar16.hip: back to back they shared their banks and the mixed-operand reads
    //  conflicted two ways -- LDSBankConflict 21 % of the kernel's time, profiles/r04m_train_unit_utilisation.json)
    constexpr int CI = 2 * N + 32, ZI = 4 * N + 32;       // dwords from the hi to the lo image
    __shared__ __attribute__((aligned(16))) unsigned c_img[CI + 2 * N];            // dac, K = N
    __shared__ __attribute__((aligned(16))) unsigned z_img[ZI + 4 * N];            // [daz | dar], K = 2N
    // [step parity][chunk]: max |dy + keep|, max |dzr|, max |g| over ALL units of a chunk, as the bits of non-negative floats (which
    // order like unsigned integers): every lane adds its value with one LDS atomic maximum.  (Reducing inside the wave first -- two
    // lane swaps and two DPP steps per quantity -- and reading one number per wave back cost 800-960 of a step's 2100-2850 cycles.)
    __shared__ unsigned s_m[2][4];
    __shared__ float s_c[2][NW];                         // weight column sums, per wave

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    for (int i = tid; i < CI + 2 * N; i += NTH) c_img[i] = 0u;
    for (int i = tid; i < ZI + 4 * N; i += NTH) z_img[i] = 0u;
    auto ldH = [](const unsigned *img, int off) { return *reinterpret_cast<const half8 *>(img + off); };
    if (tid < 8) s_m[tid >> 2][tid & 3] = 0u;
    auto amax = [](unsigned *slot, float v) { atomicMax(slot, __float_as_uint(fabsf(v))); };

    const int c = lane & 3, q = (lane >> 2) & 3, g = lane >> 4;
    // A operands: row = output unit 16w + (lane & 15), element (kb, e) = W[k][unit] with k = 32kb + 16(e&1) + 4g + (e>>1); rows
    // scaled to [1, 2).  Image position k -> weight row: the first N positions are units 0 .. n-1 (of daz / dac), the next N dar's.
    half8 w1h[KB1], w1l[KB1], w2h[KB2], w2l[KB2];
    half8 w3h[DX ? KB2 : 1], w3l[DX ? KB2 : 1], w4h[DX ? KB1 : 1], w4l[DX ? KB1 : 1];       // iW^T: the [z | r] rows, the c rows
    float inv1, inv2, inv3 = 0.0f, inv4 = 0.0f, C1, C2;
    {
        const int unit = 16 * w + (lane & 15);
        auto prep = [&](const float *Wm, int ldw, int ncols, auto &wh, auto &wl, float &invq, float &colsum) {
            const bool uk = unit < ncols;
            constexpr int KBN = sizeof(wh) / sizeof(half8);
            float v[KBN][8];
            float m = 0.0f, sa = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KBN; kb++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = 32 * kb + 16 * (e & 1) + 4 * g + (e >> 1);
                    const int ku = k < N ? k : k - N, krow = k < N ? ku : n + ku;
                    v[kb][e] = (uk && ku < n) ? Wm[(size_t)krow * ldw + unit] : 0.0f;
                    m = fmaxf(m, fabsf(v[kb][e]));
                    sa += fabsf(v[kb][e]);
                }
            float iv;
            const float sc = pow2_scale(kgroup_max(m), iv);
            invq = __shfl(iv, 4 * g + q);
            sa += __shfl_xor(sa, 16);
            sa += __shfl_xor(sa, 32);                    // sum_k |W[k][unit]|
            colsum = sa;
#pragma unroll
            for (int kb = 0; kb < KBN; kb++)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float a = v[kb][e] * sc;
                    const _Float16 hh = (_Float16)a;
                    wh[kb][e] = hh;
                    wl[kb][e] = (_Float16)(a - (float)hh);
                }
        };
        float cs1, cs2;
        prep(sW2, n, n, w1h, w1l, inv1, cs2);
        prep(sW, n, n, w2h, w2l, inv2, cs1);
        if constexpr (DX) {
            float unused;
            prep(iW, insize, insize, w3h, w3l, inv3, unused);
            prep(iW + (size_t)2 * n * insize, insize, insize, w4h, w4l, inv4, unused);
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            cs1 = fmaxf(cs1, __shfl_xor(cs1, o));
            cs2 = fmaxf(cs2, __shfl_xor(cs2, o));
        }
        if (lane == 0) { s_c[0][w] = cs1; s_c[1][w] = cs2; }
        __syncthreads();                                 // (also: LDS images initialised)
        C1 = s_c[0][0];
        C2 = s_c[1][0];
#pragma unroll
        for (int k = 1; k < NW; k++) { C1 = fmaxf(C1, s_c[0][k]); C2 = fmaxf(C2, s_c[1][k]); }
        C1 *= 1.001f;                                    // a little above the exact sums: the products are rounded
        C2 *= 1.001f;
    }
    const float D2 = fmaxf(0.5f, 0.25f * C2);            // max |dzr| <= D2 max |g|
    int moff1[KB1], moff2[KB2];
#pragma unroll
    for (int kb = 0; kb < KB1; kb++) moff1[kb] = (q >> 1) * CI + ((kb * 4 + g) * 4 + c) * 4;   // in dwords, my column group's image
#pragma unroll
    for (int kb = 0; kb < KB2; kb++) moff2[kb] = (q >> 1) * ZI + ((kb * 4 + g) * 4 + c) * 4;

    const int bc = b0 + c;
    const bool live = bc < B;
    const int bcc = live ? bc : B - 1;
    // my unit u = 16w + 4g + q -> K block w >> 1, half w & 1, k group g, r = q
    const int u0 = 16 * w + 4 * g + q;
    const bool uok = u0 < n;
    const int uu = uok ? u0 : 0;
    const int wpos = ((((w >> 1) * 4 + g) * 4 + c) * 4 + q) * 2 + (w & 1);          // position of k = u in an image, in halves
    // rows of scan step s: (reverse ? T-1-s : s) * B + chunk; the pass walks s = T-1 .. 0 (request i is scan step T-1-i)
    const long rstep = reverse ? (long)B : -(long)B;
    const long row0 = (long)(reverse ? 0 : T - 1) * B + bcc;
    struct Ops { float dy, z, r, ht, hp, yb; };
    Ops vs[5];                                           // five register sets: requests run four steps ahead
    unsigned o_dy = (unsigned)((row0 * lddy + uu) * (long)sizeof(float)), o_z = (unsigned)((row0 * 2L * n + uu) * (long)sizeof(float));
    unsigned o_ht = (unsigned)((row0 * ldh + uu) * (long)sizeof(float)), o_hp = (unsigned)((row0 * ldhp + uu) * (long)sizeof(float));
    const unsigned st_dy = (unsigned)(rstep * lddy * (long)sizeof(float)), st_z = (unsigned)(rstep * 2L * n * (long)sizeof(float));
    const unsigned st_ht = (unsigned)(rstep * ldh * (long)sizeof(float)), st_hp = (unsigned)(rstep * ldhp * (long)sizeof(float));
    const unsigned r_off = (unsigned)(n * (int)sizeof(float));
    unsigned o_yb = DA ? (unsigned)((row0 * ldyref + (u0 < insize ? u0 : 0)) * (long)sizeof(float)) : 0u;
    const unsigned st_yb = DA ? (unsigned)(rstep * ldyref * (long)sizeof(float)) : 0u;
    int vnext = 0;
    auto load_v = [&](Ops &v) {                          // (requests past the first scan step re-read it: the count per step is fixed)
        gw_gload1(v.dy, o_dy, dy);
        gw_gload1(v.z, o_z, zr);
        gw_gload1(v.r, o_z + r_off, zr);
        gw_gload1(v.ht, o_ht, hout);
        gw_gload1(v.hp, o_hp, hprev);
        if constexpr (DA) gw_gload1(v.yb, o_yb, yref);
        vnext++;
        if (vnext < T) { o_dy += st_dy; o_z += st_z; o_ht += st_ht; o_hp += st_hp; o_yb += st_yb; }
    };
    load_v(vs[0]);
    load_v(vs[1]);
    load_v(vs[2]);
    load_v(vs[3]);
    float *dap = da + (row0 * 3L * n + uu), *rhp = rh + (row0 * (long)n + uu);
    const long dstep = rstep * 3L * n, rhstep = rstep * (long)n;

    // dx rows lag one step: the [daz | dar] half of a step's product is formed behind the NEXT step's barrier X
    const bool xok = u0 < insize;
    float *dxp = DX ? dx + (row0 * lddx + (xok ? u0 : 0)) : nullptr;
    const long dxstep = rstep * lddx;
    float dxc_kept = 0.0f;                               // the dac half of the step before, already unscaled
    float yb_prev = 0.0f;                                // (DA) the layer below's output at the step before
    float keep = 0.0f;
    float invs2 = 1.0f;                                  // inverse of the scale the [daz | dar] image in LDS carries
    // before the first step: max |dy| of the first request (keep = 0, no image yet)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
    asm volatile("" : "+v"(vs[0].dy));
    amax(&s_m[0][c], uok ? 2.0f * vs[0].dy : 0.0f);

    auto dx_zr = [&](f32x4 &xz, const half8 *bz) __attribute__((always_inline)) {
        if constexpr (DX) {
#if GW_ABL_SKIP
            if (w >= GW_ABL_SKIP) { xz = f32x4{0.f, 0.f, 0.f, 0.f}; return; }      // timing experiment: the waves that share a SIMD skip their tile
#endif
            static_for<0, KB2>([&](auto KC) {
                constexpr int kb = decltype(KC)::value;
                gw_mma<kb == 0>(xz, w3l[kb], bz[kb]);
                gw_mma<false>(xz, w3h[kb], bz[kb]);
            });
        }
    };
    auto dx_c = [&](f32x4 &xc, const half8 *bm) __attribute__((always_inline)) {
        if constexpr (DX) {
#if GW_ABL_SKIP
            if (w >= GW_ABL_SKIP) { xc = f32x4{0.f, 0.f, 0.f, 0.f}; return; }
#endif
            static_for<0, KB1>([&](auto KC) {
                constexpr int kb = decltype(KC)::value;
                gw_mma<kb == 0>(xc, w4l[kb], bm[kb]);
                gw_mma<false>(xc, w4h[kb], bm[kb]);
            });
        }
    };
    auto step = [&](auto PHC, const int par, const bool first_step) {   // one step of the pass; the i-th one (par = i & 1) handles scan step s = T-1-i
        constexpr int ph = decltype(PHC)::value;
        Ops &cur = vs[ph];
        Ops &nxt = vs[(ph + 1) % 5];
        // ---- barrier X: the [daz | dar] image of the step before and its maxima are there ----
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bz[KB2];
#pragma unroll
        for (int kb = 0; kb < KB2; kb++) bz[kb] = ldH(z_img, moff2[kb]);
#if GW_ABL & 4
        const float Gb = 1.0f;
#else
        const float Gb = __uint_as_float(s_m[par][c]);   // >= max |g| of this step over the chunk's units (see below)
        // the other parity's slot is empty again before it is filled behind this step's barrier Y (it was read last a whole barrier ago)
        if (tid < 4) s_m[par ^ 1][tid] = 0u;
#endif
#if !(GW_ABL & 2)
        load_v(vs[(ph + 4) % 5]);
#endif
        f32x4 a2;                                        // (two chains over alternating K blocks: measured, 3 % slower)
        static_for<0, KB2>([&](auto KC) {
            constexpr int kb = decltype(KC)::value;
            gw_mma<kb == 0>(a2, w2l[kb], bz[kb]);
            gw_mma<false>(a2, w2h[kb], bz[kb]);
        });
        // the operands of this step and dy of the next one have arrived once only the three youngest requests are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPS) : "memory");
        asm volatile("" : "+v"(cur.dy), "+v"(cur.z), "+v"(cur.r), "+v"(cur.ht), "+v"(cur.hp), "+v"(nxt.dy));
        if constexpr (DA) asm volatile("" : "+v"(cur.yb));
        f32x4 xz;                                        // the [daz | dar] half of dx of the step before (formed in front of barrier Y)
        mfma_drain(a2);                                  // pick_mix reads the accumulator from asm
        float invs1;
        const float sc1 = gw_pow2_top(Gb, invs1);
        // csrc/train.hip, gru_backward_kernel
        const float p2 = pick_mix(a2) * inv2 * invs2;
        const float invs2_prev = invs2;
        const float z = cur.z, r = cur.r, h = cur.hp;
        const float omz = 1.0f - z;
        const float cc = omz > 0.0f ? fminf(fmaxf((cur.ht - z * h) * __builtin_amdgcn_rcpf(omz), -1.0f), 1.0f) : 0.0f;
        const float gg = uok ? cur.dy + keep + p2 : 0.0f;
        const float dac = gg * omz * (1.0f - cc * cc);
        const float daz = gg * (h - cc) * z * omz;
#if !(GW_ABL & 1)
        if (live && uok) {
            dap[2 * n] = dac;
            dap[0] = daz;
        }
#endif
        {
            float hv = dac * sc1;
            asm volatile("" : "+v"(hv));                 // split2's note on v_fma_mixlo_f16 applies
            const _Float16 h16 = (_Float16)hv;
            const _Float16 l16 = (_Float16)(hv - (float)h16);
            reinterpret_cast<unsigned short *>(&c_img[0])[wpos] = __builtin_bit_cast(unsigned short, h16);
            reinterpret_cast<unsigned short *>(&c_img[CI])[wpos] = __builtin_bit_cast(unsigned short, l16);
        }
        // the [daz | dar] half of dx of the step before, on the operands fetched behind barrier X: its MFMAs behind the dac image's
        // write, under that write's way to LDS and the wait at barrier Y (right behind the chain's own MFMAs they cost 80 cycles more)
        dx_zr(xz, bz);
        // ---- barrier Y: the dac image and max |g| are there ----
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bm[KB1];
#pragma unroll
        for (int kb = 0; kb < KB1; kb++) bm[kb] = ldH(c_img, moff1[kb]);
        f32x4 a1;
        static_for<0, KB1>([&](auto KC) {
            constexpr int kb = decltype(KC)::value;
            gw_mma<kb == 0>(a1, w1l[kb], bm[kb]);
            gw_mma<false>(a1, w1h[kb], bm[kb]);
        });
        f32x4 xc;
        mfma_drain(a1);
        const float sc2 = gw_pow2_top(Gb * D2, invs2);
        const float drh = uok ? pick_mix(a1) * inv1 * invs1 : 0.0f;
        const float dar = drh * h * r * (1.0f - r);
        keep = gg * z + drh * r;
#if !(GW_ABL & 1)
        if (live && uok) {
            dap[n] = dar;
            rhp[0] = r * h;
        }
#else
        if (live && uok && T < 0) { dap[n] = dar + dac + daz; rhp[0] = r * h; }
#endif
        dap += dstep;
        rhp += rhstep;
        {
#if !(GW_ABL & 4)
            // |g| of the next step <= max_u |dy + keep| + C1 max_u |dzr| <= 2 max_u (|dy + keep| + C1 |dzr|): ONE number per chunk and
            // step.  (Round 3 exchanged the two maxima and the exact max |g| as well: three atomics, 2260 cycles per step; one: 1950;
            // none -- fixed scales, a timing experiment -- 1640.  Issuing it behind a barrier instead of in front of one, or from 16
            // lanes after a reduction over the k groups, changes nothing or costs: tools/bwd16_variants.py.)
            amax(&s_m[par ^ 1][c], uok ? 2.0f * fmaf(C1, fmaxf(fabsf(daz), fabsf(dar)), fabsf(nxt.dy + keep)) : 0.0f);
#endif
            float v0 = daz * sc2, v1 = dar * sc2;
            asm volatile("" : "+v"(v0), "+v"(v1));
            const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
            const _Float16 l0 = (_Float16)(v0 - (float)h0), l1 = (_Float16)(v1 - (float)h1);
            unsigned short *ih = reinterpret_cast<unsigned short *>(&z_img[0]), *il = reinterpret_cast<unsigned short *>(&z_img[ZI]);
            ih[wpos] = __builtin_bit_cast(unsigned short, h0);                         // k = u
            il[wpos] = __builtin_bit_cast(unsigned short, l0);
            ih[wpos + KB1 * 128] = __builtin_bit_cast(unsigned short, h1);              // k = N + u: KB1 K blocks further
            il[wpos + KB1 * 128] = __builtin_bit_cast(unsigned short, l1);
        }
        if constexpr (DX) {
            // (behind the chain's own work of this half: the image write above is on its way to LDS meanwhile)
            // dx of the step before: its dac half (kept) + its [daz | dar] half (xz, long complete: a barrier and this step's first half ago)
            float dxv = fmaf(pick_mix(xz) * inv3, invs2_prev, dxc_kept);
            if constexpr (DA) {
                dxv *= gw_dact(yb_prev, dact);
                yb_prev = cur.yb;
            }
#if !(GW_ABL & 1)
            if (live && xok && !first_step) dxp[-dxstep] = dxv;
#else
            if (live && xok && T < 0) dxp[0] = dxv;
#endif
        }
        if constexpr (DX) {
            dx_c(xc, bm);                                  // this step's dac half, under the [daz | dar] image's way to LDS
            mfma_drain(xc);
            dxc_kept = pick_mix(xc) * inv4 * invs1;        // ... unscaled (invs1: the scale of this step's dac image)
            dxp += dxstep;
        }
    };
    for (int i = 0; i < T; i += 5) {
        step(ic<0>{}, i & 1, i == 0);
        if (i + 1 < T) step(ic<1>{}, (i + 1) & 1, false);
        if (i + 2 < T) step(ic<2>{}, i & 1, false);
        if (i + 3 < T) step(ic<3>{}, (i + 1) & 1, false);
        if (i + 4 < T) step(ic<4>{}, i & 1, false);
    }
    if constexpr (DX) {
        // the last step's [daz | dar] half: one more exchange
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        half8 bz[KB2];
#pragma unroll
        for (int kb = 0; kb < KB2; kb++) bz[kb] = ldH(z_img, moff2[kb]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 xz;
        dx_zr(xz, bz);
        mfma_drain(xz);
        float dxv = fmaf(pick_mix(xz) * inv3, invs2, dxc_kept);
        if constexpr (DA) dxv *= gw_dact(yb_prev, dact);
        if (live && xok) dxp[-dxstep] = dxv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing of mine may land in registers after the wave has ended
}

template <int N, bool DX, bool DA>
static size_t gru_bwd16_exclusive_lds()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(gru_bwd16_kernel<N, DX, DA>)) != hipSuccess) return 0;
    const size_t half_cu = 80 * 1024 + 512;
    const size_t dyn = attr.sharedSizeBytes >= half_cu ? 0 : half_cu - attr.sharedSizeBytes;
    if (dyn && hipFuncSetAttribute(reinterpret_cast<const void *>(gru_bwd16_kernel<N, DX, DA>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)dyn) != hipSuccess)
        return 0;
    return dyn;
}

template <int N, bool DX, bool DA = false>
static int gru_bwd16_launch(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h, long ldh,
                            const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n, int reverse, const float *iW,
                            float *dx, long lddx, int insize, const float *yref, long ldyref, int dact, hipStream_t s)
{
    const size_t dyn = SLK_PER_DEVICE(size_t, (gru_bwd16_exclusive_lds<N, DX, DA>()));
    hipLaunchKernelGGL((gru_bwd16_kernel<N, DX, DA>), dim3((B + 3) / 4), dim3(4 * N), dyn, s, dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2,
                       da, rh, T, B, n, reverse & 1, iW, dx, lddx, insize, yref, ldyref, dact);
    return slk_launch_status();
}

static int gru_bwd16_entry(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h, long ldh,
                           const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n, int reverse, int act,
                           int gate_act, const float *iW, float *dx, long lddx, int insize, const float *yref, long ldyref, int dact,
                           slk_stream_t stream)
{
    if (!dy || !hprev || !zr || !h || !sW || !sW2 || !da || !rh || T < 1 || B < 1 || n < 1 || lddy < n || ldh < n || ldhp < n)
        return SLK_ERR_INVALID_ARG;
    if (dx && (!iW || insize < 1 || lddx < insize)) return SLK_ERR_INVALID_ARG;
    if (yref && (!dx || ldyref < insize || !slk_act_valid(dact))) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if (n % 16 || n > 128) return SLK_ERR_UNSUPPORTED;
    const unsigned long long rows = (unsigned long long)T * B, lim = 1ull << 32;
    if (rows * 3 * n * sizeof(float) >= lim || rows * lddy * sizeof(float) >= lim || rows * ldh * sizeof(float) >= lim ||
        rows * ldhp * sizeof(float) >= lim || (yref && rows * ldyref * sizeof(float) >= lim))
        return SLK_ERR_UNSUPPORTED;                      // 32-bit lane offsets
    hipStream_t s = slk_stream(stream);
    if (dx) {
        // a wave forms the product's 16 columns of its own tile: at most as many inputs as the layer has units (rounded up to the kernel's
        // width), and the weights of both products in a wave's 256 registers (n <= 96)
        const int N = n <= 64 ? 64 : 96;
        if (n > 96 || insize > N) return SLK_ERR_UNSUPPORTED;
#define GW_GO(NN, DAF) gru_bwd16_launch<NN, true, DAF>(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, iW, dx, lddx, insize, yref, ldyref, dact, s)
        if (n <= 64) return yref ? GW_GO(64, true) : GW_GO(64, false);
        return yref ? GW_GO(96, true) : GW_GO(96, false);
#undef GW_GO
    }
    if (n <= 64) return gru_bwd16_launch<64, false>(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, nullptr, nullptr, 0, 0, nullptr, 0, 0, s);
    if (n <= 96) return gru_bwd16_launch<96, false>(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, nullptr, nullptr, 0, 0, nullptr, 0, 0, s);
    return gru_bwd16_launch<128, false>(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, nullptr, nullptr, 0, 0, nullptr, 0, 0, s);
}

// include/sloika_amd.h
extern "C" int slk_gru_backward16_f32(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                                      long ldh, const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n,
                                      int reverse, int act, int gate_act, slk_stream_t stream)
{
    return gru_bwd16_entry(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, act, gate_act, nullptr, nullptr, 0, 0,
                           nullptr, 0, 0, stream);
}

extern "C" int slk_gru_backward16_dx_f32(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                                         long ldh, const float *sW, const float *sW2, const float *iW, float *da, float *rh, float *dx,
                                         long lddx, int T, int B, int n, int insize, int reverse, int act, int gate_act,
                                         const float *yref, long ldyref, int dact, slk_stream_t stream)
{
    if (!dx || !iW) return SLK_ERR_INVALID_ARG;
    return gru_bwd16_entry(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, act, gate_act, iW, dx, lddx, insize,
                           yref, ldyref, dact, stream);
}
