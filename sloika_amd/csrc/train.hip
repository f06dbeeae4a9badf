// train.hip -- the training step of bin/train_network.py:124-142 (`wrap_network`: loss, accuracy, th.grad of the loss,
// updates.adam) on gfx950.  SURVEY.md section 8 row f2.
//
// The reference gets its gradient from Theano's automatic differentiation; here the reverse pass is written out for the
// three layer types of the raw models (Convolution layers.py:417-419, Gru.step layers.py:1010-1021, Softmax layers.py:
// 309-314).  Design:
//   * the forward pass IS the inference path (conv kernel, gru_fused.hip, gemm_rows_f16x3.hip) -- nothing is saved
//     inside the recurrent kernel.  What the reverse scan needs (gates z, r and candidate c of every step) is a
//     function of (x_t, h_{t-1}) only, and every h_{t-1} is known once the forward scan has finished, so the gates of
//     ALL steps are recomputed time-parallel: [z r] as one GEMM over packed rows [x_t | h_{t-1}] (slk_train_pack_xh_f32 +
//     slk_gemm_bias_act_f16x3), the candidate inside the reverse scan as c = (h_t - z h_{t-1}) / (1 - z);
//   * the only sequential part is gru_backward_kernel: per step two dependent matrix-vector products with sW2^T and
//     sW^T (float32 FMA, weights held in registers, one workgroup per chunk) producing the pre-activation gradients
//     da = dL/d(vI) for every step;
//   * every weight gradient is then a time-parallel contraction over all T*B rows, C = A^T B (gemm_tn_kernel, float32
//     MFMA 32x32x2 reading both operands straight from their row-major layout, split over M with a deterministic
//     second-stage sum), bias gradients are the same contraction against a column of ones, and dL/dx of a layer is
//     slk_gemm_bias_act_f32 with the transposed weight;
//   * softmax + weighted cross-entropy + its gradient is one pass over the logits (softmax_xent_grad_kernel), in place;
//   * ADAMski (updates.py:36-89) is one element-wise kernel over the flat parameter / gradient / moment buffers; the
//     data-parallel all-reduce of the flat gradient (RCCL) happens on the host side between the two.
#include <type_traits>

#include "common.h"
#include "mfma4.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef GBWD_DIAG
#define GBWD_DIAG 0
#endif

// ---------------------------------------------------------------------------------------------------------------
// Packed rows for the gate recompute: xh[m] = [x[m] | h_prev[m]], h_prev(t, b) = h at the previous SCAN step (zero at
// the scan start: layers.py:85-88), the scan running backwards in time when `reverse` (layers.py:1449-1450).
// ---------------------------------------------------------------------------------------------------------------
// V = 4: float4 per thread (sizes, strides and pointers multiples of 4 floats), V = 1: any
template <int V>
__global__ void __launch_bounds__(256) pack_xh_kernel(const float *__restrict__ x, long ldx, const float *__restrict__ h,
                                                      long ldh, float *__restrict__ xh, int T, int B, int I, int N,
                                                      int reverse)
{
    typedef float vec __attribute__((ext_vector_type(V)));
    const int W = (I + N) / V;
    const size_t total = (size_t)T * B * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t m = e / W;
        const int j = (int)(e - m * W) * V;
        vec v = 0.0f;
        if (j < I) {
            v = *reinterpret_cast<const vec *>(x + m * ldx + j);
        } else {
            const int t = (int)(m / B), b = (int)(m - (size_t)t * B);
            const int tp = reverse ? t + 1 : t - 1;
            if (tp >= 0 && tp < T) v = *reinterpret_cast<const vec *>(h + ((size_t)tp * B + b) * ldh + (j - I));
        }
        *reinterpret_cast<vec *>(xh + e * V) = v;
    }
}

// xrh[m] = [x[m] | r[m] * h_prev[m]] from xh and the gates zr[m] = [z | r]
template <int V>
__global__ void __launch_bounds__(256) pack_xrh_kernel(const float *__restrict__ xh, const float *__restrict__ zr,
                                                       float *__restrict__ xrh, size_t M, int I, int N)
{
    typedef float vec __attribute__((ext_vector_type(V)));
    const int W = (I + N) / V;
    const size_t total = M * W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t m = e / W;
        const int j = (int)(e - m * W) * V;
        vec v = *reinterpret_cast<const vec *>(xh + e * V);
        if (j >= I) v *= *reinterpret_cast<const vec *>(zr + m * (2 * (size_t)N) + N + (j - I));
        *reinterpret_cast<vec *>(xrh + e * V) = v;
    }
}

static unsigned elementwise_grid(size_t total)
{
    size_t blocks = (total + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks));
}

extern "C" int slk_train_pack_xh_f32(const float *x, long ldx, const float *h, long ldh, float *xh, int T, int B, int I,
                                     int N, int reverse, slk_stream_t stream)
{
    if (!x || !h || !xh || T < 1 || B < 1 || I < 1 || N < 1 || ldx < I || ldh < N) return SLK_ERR_INVALID_ARG;
    const bool v4 = I % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldh % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(xh)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(pack_xh_kernel<4>, dim3(elementwise_grid((size_t)T * B * (I + N) / 4)), dim3(256), 0,
                           slk_stream(stream), x, ldx, h, ldh, xh, T, B, I, N, reverse);
    else
        hipLaunchKernelGGL(pack_xh_kernel<1>, dim3(elementwise_grid((size_t)T * B * (I + N))), dim3(256), 0, slk_stream(stream),
                           x, ldx, h, ldh, xh, T, B, I, N, reverse);
    return slk_launch_status();
}

extern "C" int slk_train_pack_xrh_f32(const float *xh, const float *zr, float *xrh, long M, int I, int N,
                                      slk_stream_t stream)
{
    if (!xh || !zr || !xrh || M < 1 || I < 1 || N < 1) return SLK_ERR_INVALID_ARG;
    const bool v4 = I % 4 == 0 && N % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(xh) | reinterpret_cast<uintptr_t>(zr) | reinterpret_cast<uintptr_t>(xrh)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(pack_xrh_kernel<4>, dim3(elementwise_grid((size_t)M * (I + N) / 4)), dim3(256), 0, slk_stream(stream),
                           xh, zr, xrh, (size_t)M, I, N);
    else
        hipLaunchKernelGGL(pack_xrh_kernel<1>, dim3(elementwise_grid((size_t)M * (I + N))), dim3(256), 0, slk_stream(stream), xh,
                           zr, xrh, (size_t)M, I, N);
    return slk_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Reverse scan of one Gru layer.  Forward step (layers.py:1010-1021), h = h_{t-1}:
//     z, r = sigmoid(vI[:2n] + h sW^T) ; c = tanh(vI[2n:] + (r*h) sW2^T) ; h_t = z*h + (1-z)*c
// Reverse step, g = dL/dh_t (from the layer above) + the carry from step t+1:
//     dac = g (1-z) (1-c^2) ; daz = g (h-c) z (1-z) ; drh = dac sW2 ; dar = drh h r (1-r)
//     carry = g z + drh r + [daz dar] sW ;  da = [daz dar dac] is dL/dvI of this step.
// One workgroup per chunk, 4*N threads: thread (i, q) holds column i of the K-quarter q of sW2 and of sW in registers
// (N/4 + 2N/4 floats) and produces a partial sum; the N "owner" threads (q = 0) do the element-wise part.  Operands of
// the next step are loaded while the current one computes (their addresses do not depend on the recursion).
// ---------------------------------------------------------------------------------------------------------------
// The candidate c is not an operand: h_t = z h + (1-z) c gives c = (h_t - z h) / (1 - z) from the layer's own output.
// Where z -> 1 the quotient is ill-conditioned, but c only enters through g (1-z) (1-c^2) and g (h-c) z (1-z), both
// carrying the factor (1-z): with c clamped to tanh's range the error stays below g (1-z), i.e. negligible exactly there.
__device__ __forceinline__ float gru_candidate(float h_t, float z, float h)
{
    const float omz = 1.0f - z;
    return omz > 0.0f ? slk_clip((h_t - z * h) / omz, -1.0f, 1.0f) : 0.0f;
}

template <int N>
__global__ void __launch_bounds__(4 * N) gru_backward_kernel(const float *__restrict__ dy, long lddy,
                                                             const float *__restrict__ hprev, long ldhp,
                                                             const float *__restrict__ zr, const float *__restrict__ hout,
                                                             long ldh, const float *__restrict__ sW,
                                                             const float *__restrict__ sW2, float *__restrict__ da,
                                                             float *__restrict__ rh, int T, int B, int reverse)
{
    constexpr int Q2 = N / 4, Q1 = 2 * N / 4;
    __shared__ float v_dac[N], v_dzr[2 * N], part[4][N];
    const int tid = threadIdx.x, i = tid % N, q = tid / N, b = blockIdx.x;
    const bool owner = q == 0;
    float w2[Q2], w1[Q1];
#pragma unroll
    for (int j = 0; j < Q2; j++) w2[j] = sW2[(size_t)(q * Q2 + j) * N + i];      // drh[i] = sum_k dac[k] sW2[k][i]
#pragma unroll
    for (int j = 0; j < Q1; j++) w1[j] = sW[(size_t)(q * Q1 + j) * N + i];       // carry[i] += sum_k dzr[k] sW[k][i]
    float carry = 0.0f;
    // operands of scan step s (owner threads only)
    auto row = [&](int s) { return (size_t)(reverse ? T - 1 - s : s) * B + b; };
    float n_g = 0.f, n_z = 0.f, n_r = 0.f, n_c = 0.f, n_h = 0.f;
    auto fetch = [&](int s) {
        const size_t m = row(s);
        n_g = dy[m * lddy + i];
        n_z = zr[m * (2 * N) + i];
        n_r = zr[m * (2 * N) + N + i];
        n_c = hout[m * ldh + i];
        n_h = hprev[m * ldhp + i];
    };
    if (owner) fetch(T - 1);
    for (int s = T - 1; s >= 0; s--) {
        float g = 0.f, z = 0.f, r = 0.f, cc = 0.f, h = 0.f, daz = 0.f;
        if (owner) {
            g = n_g + carry; z = n_z; r = n_r; h = n_h; cc = gru_candidate(n_c, z, h);
            if (s > 0) fetch(s - 1);
            const float dac = g * (1.0f - z) * (1.0f - cc * cc);
            daz = g * (h - cc) * z * (1.0f - z);
            v_dac[i] = dac;
            v_dzr[i] = daz;
            da[row(s) * (3 * N) + 2 * N + i] = dac;
            da[row(s) * (3 * N) + i] = daz;
        }
        __syncthreads();
        {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < Q2; j++) acc = fmaf(v_dac[q * Q2 + j], w2[j], acc);
            part[q][i] = acc;
        }
        __syncthreads();
        float keep = 0.0f;
        if (owner) {
            const float drh = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
            const float dar = drh * h * r * (1.0f - r);
            v_dzr[N + i] = dar;
            da[row(s) * (3 * N) + N + i] = dar;
            rh[row(s) * N + i] = r * h;
            keep = g * z + drh * r;
        }
        __syncthreads();
        {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < Q1; j++) acc = fmaf(v_dzr[q * Q1 + j], w1[j], acc);
            part[q][i] = acc;
        }
        __syncthreads();
        if (owner) carry = keep + ((part[0][i] + part[1][i]) + (part[2][i] + part[3][i]));
        // the next iteration writes v_dac / v_dzr[0:N] (read in this iteration before the last two barriers) and part[]
        // only after its own first barrier, which every reader of part[] above has to reach first: no barrier needed here
    }
}

// Same scan, restructured (the step is a chain of dependent LDS round trips and barriers, and with one output per lane
// every FMA needs its own LDS operand -- the vector broadcast alone costs 72 floats per lane and step, 110 KB per chunk,
// i.e. ~860 cycles of LDS bandwidth):
//   * lane (row g, slice sl) of wave w accumulates FOUR outputs (16w + 4g + 0..3) over ONE SIXTEENTH of k, so a vector
//     element read from LDS feeds four FMAs (two v_pk_fma_f32) and the LDS traffic drops fourfold;
//   * the 16 partial sums of an output meet inside a DPP row: two reduce-scatter steps inside the quad (each lane keeps
//     half of what it holds, which halves the adds; the accumulator order is pre-permuted per lane through the weight
//     layout, so no selects are needed) and two rotations across the quads: 5 instructions for 4 outputs, no LDS, no
//     barrier.  Afterwards lane sl holds output 4g + m(sl), m = bit-swap of sl & 3, replicated in the four quads;
//   * two chunks per workgroup share the weights in registers (quad q of every row finishes chunk q), so a batch of
//     1024 is resident at once instead of in two rounds;
//   * operands of every step arrive through a dedicated loader wave and LDS-DMA (global_load_lds) D steps ahead.  In
//     the plain kernel the owner threads' loads and stores share one vmcnt counter, so "wait for this step's operands"
//     also waits for everything younger; here the compute waves never wait on memory (their da stores just drain), the
//     loader issues loads only, so s_waitcnt vmcnt(10*(D-1)) means exactly "the step needed next has landed";
//   * two LDS-only barriers per step; the vectors the next step overwrites are double-buffered by step parity.
// Needs 16-byte aligned operand rows.
__device__ __forceinline__ float dpp_add_xor1(float keep, float send)
{
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1 /* quad_perm 1,0,3,2 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_xor2(float keep, float send)
{
    return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E /* quad_perm 2,3,0,1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_ror(float v, int /*4 or 8*/ n)
{
    return n == 4 ? v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /* row_ror:4 */, 0xf, 0xf, false))
                  : v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
// acc01/acc23: this lane's partial sums of its four outputs in permuted order (position p <-> output p ^ m).  Returns the
// complete sum of output m, the same in all four quads of the row.
__device__ __forceinline__ float row_reduce4(f32x2 acc01, f32x2 acc23)
{
    const float a = dpp_add_xor1(acc01.x, acc23.x);
    const float b = dpp_add_xor1(acc01.y, acc23.y);
    const float c = dpp_add_xor2(a, b);
    return dpp_add_ror(dpp_add_ror(c, 4), 8);
}

template <int N>
__global__ void __launch_bounds__(4 * N + 64) gru_backward_dma_kernel(const float *__restrict__ dy, long lddy,
                                                                      const float *__restrict__ hprev, long ldhp,
                                                                      const float *__restrict__ zr,
                                                                      const float *__restrict__ hout, long ldh,
                                                                      const float *__restrict__ sW,
                                                                      const float *__restrict__ sW2, float *__restrict__ da,
                                                                      float *__restrict__ rh, int T, int B, int reverse)
{
    constexpr int CH = 2, KA = N / 16, KB = 2 * N / 16, D = 6;
    static_assert(5 * CH * (D - 1) <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(16))) float ring[D][CH][5][N];
    __shared__ __attribute__((aligned(16))) float v_dac[2][CH][N], v_dzr[2][CH][2 * N];
    const int tid = threadIdx.x, b0 = blockIdx.x * CH, lane = tid & 63;
    const bool loader = tid >= 4 * N;
    const int sl = lane & 15, quad = sl >> 2, m = ((sl & 1) << 1) | ((sl >> 1) & 1);
    const int ibase = loader ? 0 : 16 * (tid >> 6) + 4 * (lane >> 4);       // first of this lane's four outputs
    const int i = ibase + m;                                                 // the output this lane finishes
    const bool owner = !loader && quad < CH && b0 + quad < B;
    const int bq = min(b0 + (quad < CH ? quad : 0), B - 1);                  // the chunk this lane finishes
    auto row = [&](int s, int bb) { return (size_t)(reverse ? T - 1 - s : s) * B + bb; };
    auto issue = [&](int sp) {                                    // loader wave: the operand rows of scan step sp
        if (lane < N / 4) {
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                const size_t mm = row(sp, min(b0 + ch, B - 1));
                float *dst = &ring[sp % D][ch][0][0];
                const float *src[5] = {dy + mm * lddy, zr + mm * (2 * N), zr + mm * (2 * N) + N, hout + mm * ldh,
                                       hprev + mm * ldhp};
#pragma unroll
                for (int a = 0; a < 5; a++)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[a] + 4 * lane),
                                                     (__attribute__((address_space(3))) void *)(dst + a * N), 16, 0, 0);
            }
        }
    };
    // weights of positions (0,1) and (2,3), position p <-> output ibase + (p ^ m)
    f32x2 wa01[KA], wa23[KA], wb01[KB], wb23[KB];
    if (!loader) {
#pragma unroll
        for (int k = 0; k < KA; k++) {                                                 // drh[i] = sum_k dac[k] sW2[k][i]
            const float *w = sW2 + (size_t)(sl * KA + k) * N + ibase;
            wa01[k].x = w[0 ^ m]; wa01[k].y = w[1 ^ m]; wa23[k].x = w[2 ^ m]; wa23[k].y = w[3 ^ m];
        }
#pragma unroll
        for (int k = 0; k < KB; k++) {                                                 // carry[i] += sum_k dzr[k] sW[k][i]
            const float *w = sW + (size_t)(sl * KB + k) * N + ibase;
            wb01[k].x = w[0 ^ m]; wb01[k].y = w[1 ^ m]; wb23[k].x = w[2 ^ m]; wb23[k].y = w[3 ^ m];
        }
    } else {
        for (int sp = T - 1; sp >= 0 && sp > T - 1 - D; sp--) issue(sp);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    float carry = 0.0f;
    for (int s = T - 1; s >= 0; s--) {
        const int par = s & 1;
        float g = 0.f, z = 0.f, r = 0.f, h = 0.f;
        if (owner) {
            const float *op = &ring[s % D][quad][0][0];
            g = op[i] + carry; z = op[N + i]; r = op[2 * N + i]; h = op[4 * N + i];
            const float cc = gru_candidate(op[3 * N + i], z, h);
            const float dac = g * (1.0f - z) * (1.0f - cc * cc);
            const float daz = g * (h - cc) * z * (1.0f - z);
            v_dac[par][quad][i] = dac;
            v_dzr[par][quad][i] = daz;
            if (GBWD_DIAG != 1) {
            da[row(s, bq) * (3 * N) + 2 * N + i] = dac;
            da[row(s, bq) * (3 * N) + i] = daz;
            }
        }
        lds_barrier();                                            // 1: dac visible; ring slot s % D is free again
        float keep = 0.0f;
        if (loader) {
            // step s-1 is read after barrier 2; the D-1 younger steps (5*CH loads each) may stay in flight -- unless
            // fewer than that were issued (the last D steps), where everything is awaited
            if (GBWD_DIAG == 2) {
            } else if (s - D >= 0) {
                issue(s - D);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * CH * (D - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            float drh[CH];
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                const float *vp = &v_dac[par][ch][sl * KA];
                f32x2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < (GBWD_DIAG == 3 ? 1 : KA); k++) {
                    const f32x2 vv = {vp[k], vp[k]};
                    a01 = __builtin_elementwise_fma(vv, wa01[k], a01);
                    a23 = __builtin_elementwise_fma(vv, wa23[k], a23);
                }
                drh[ch] = row_reduce4(a01, a23);
            }
            if (owner) {
                const float mine = quad == 0 ? drh[0] : drh[CH - 1];
                const float dar = mine * h * r * (1.0f - r);
                v_dzr[par][quad][N + i] = dar;
                if (GBWD_DIAG != 1) {
                    da[row(s, bq) * (3 * N) + N + i] = dar;
                    rh[row(s, bq) * N + i] = r * h;
                }
                keep = g * z + mine * r;
            }
        }
        lds_barrier();                                            // 2: [daz dar] visible; ring slot (s-1) % D has landed
        if (!loader) {
            float tot[CH];
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                const float *vp = &v_dzr[par][ch][sl * KB];
                f32x2 a01 = {0.0f, 0.0f}, a23 = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < (GBWD_DIAG == 3 ? 1 : KB); k++) {
                    const f32x2 vv = {vp[k], vp[k]};
                    a01 = __builtin_elementwise_fma(vv, wb01[k], a01);
                    a23 = __builtin_elementwise_fma(vv, wb23[k], a23);
                }
                tot[ch] = row_reduce4(a01, a23);
            }
            carry = keep + (quad == 0 ? tot[0] : tot[CH - 1]);
        }
    }
}

template <int N>
static int launch_gru_backward(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                               long ldh, const float *sW, const float *sW2, float *da, float *rh, int T, int B, int reverse,
                               hipStream_t s)
{
    const bool aligned = lddy % 4 == 0 && ldhp % 4 == 0 && ldh % 4 == 0 &&
                         ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(hprev) |
                           reinterpret_cast<uintptr_t>(zr) | reinterpret_cast<uintptr_t>(h)) & 15) == 0;
    if (aligned)
        hipLaunchKernelGGL((gru_backward_dma_kernel<N>), dim3((B + 1) / 2), dim3(4 * N + 64), 0, s, dy, lddy, hprev, ldhp, zr, h,
                           ldh, sW, sW2, da, rh, T, B, reverse);
    else
        hipLaunchKernelGGL((gru_backward_kernel<N>), dim3(B), dim3(4 * N), 0, s, dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da,
                           rh, T, B, reverse);
    return slk_launch_status();
}

int slk_gru_backward_mfma_dispatch(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                                   long ldh, const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n,
                                   int reverse, hipStream_t s);                                  // gru_backward_mfma.hip

extern "C" int slk_gru_backward_f32(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                                    long ldh, const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n,
                                    int reverse, int act, int gate_act, slk_stream_t stream)
{
    if (!dy || !hprev || !zr || !h || !sW || !sW2 || !da || !rh || T < 1 || B < 1 || n < 1 || lddy < n || ldh < n || ldhp < n)
        return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    hipStream_t s = slk_stream(stream);
    {
        const int rc = slk_gru_backward_mfma_dispatch(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, n, reverse, s);
        if (rc != SLK_ERR_UNSUPPORTED) return rc;              // gru_backward_mfma.hip: aligned rows, n <= 128
    }
    switch (n) {
#define GRU_BWD_CASE(NN) \
    case NN: return launch_gru_backward<NN>(dy, lddy, hprev, ldhp, zr, h, ldh, sW, sW2, da, rh, T, B, reverse, s)
        GRU_BWD_CASE(16); GRU_BWD_CASE(32); GRU_BWD_CASE(48); GRU_BWD_CASE(64); GRU_BWD_CASE(96); GRU_BWD_CASE(112);
        GRU_BWD_CASE(128); GRU_BWD_CASE(144);
#undef GRU_BWD_CASE
    default: return SLK_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Lstm (layers.py:677-697; gate rows interleaved, row j*4 + gate: 0 candidate, 1 input, 2 forget, 3 output; peepholes p).
// The forward kernels keep only the layer output.  Every out_{t-1} is known after the forward pass, so the summed gate
// inputs of ALL steps are one time-parallel GEMM over [x_t | out_{t-1}]; what remains sequential is the cell recursion
// c_t = c_{t-1} f + g i through the peepholes, which is element-wise: lstm_gates_kernel scans it with one thread per
// (chunk, neuron) and leaves the activated gates and the cell states for the reverse scan.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) lstm_gates_kernel(const float *__restrict__ sum, const float *__restrict__ peep,
                                                         float *__restrict__ gates, float *__restrict__ cell, int T, int B,
                                                         int n, int reverse)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * n) return;
    const int b = idx / n, j = idx - b * n;
    const float p0 = peep ? peep[j] : 0.0f, p1 = peep ? peep[n + j] : 0.0f, p2 = peep ? peep[2 * n + j] : 0.0f;
    float c = 0.0f;
#pragma unroll 4
    for (int s = 0; s < T; s++) {
        const size_t m = (size_t)(reverse ? T - 1 - s : s) * B + b;
        const float4 v = *reinterpret_cast<const float4 *>(sum + m * (4 * (size_t)n) + 4 * j);
        const float g = slk_tanh(v.x);
        const float i = slk_sigmoid(v.y + c * p0);                                   // layers.py:688
        const float f = slk_sigmoid(v.z + c * p1);                                   // layers.py:686
        const float cn = c * f + g * i;
        const float o = slk_sigmoid(v.w + cn * p2);                                  // layers.py:690
        *reinterpret_cast<float4 *>(gates + m * (4 * (size_t)n) + 4 * j) = make_float4(g, i, f, o);
        cell[m * n + j] = cn;
        c = cn;
    }
}

extern "C" int slk_lstm_gates_f32(const float *sum, const float *peep, float *gates, float *cell, int T, int B, int n,
                                  int reverse, slk_stream_t stream)
{
    if (!sum || !gates || !cell || T < 1 || B < 1 || n < 1) return SLK_ERR_INVALID_ARG;
    if (((reinterpret_cast<uintptr_t>(sum) | reinterpret_cast<uintptr_t>(gates)) & 15) != 0) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(lstm_gates_kernel, dim3((unsigned)(((size_t)B * n + 255) / 256)), dim3(256), 0, slk_stream(stream), sum,
                       peep, gates, cell, T, B, n, reverse);
    return slk_launch_status();
}

// Reverse scan of one Lstm layer.  With go = dL/dout_t + carry_out, tc = tanh(c_t):
//     do' = go tc o(1-o) ; dc = go o (1-tc^2) + do' p2 + carry_c ; di' = dc g i(1-i) ; df' = dc c_{t-1} f(1-f)
//     dg' = dc i (1-g^2) ; carry_c = dc f + di' p0 + df' p1 ; carry_out = [dg' di' df' do'] . sW   (one 4n -> n product)
// dsum:[M][4n] (interleaved like the gates) is dL/d(summed gate inputs); dpeep:[B][3][n] the per-chunk peephole
// gradients (summed over chunks by the caller).  One workgroup per chunk, 4n threads, thread (k, quarter) holds column k
// of one quarter of sW's rows in registers; the n owner threads (quarter 0) do the element-wise part.
template <int N>
__global__ void __launch_bounds__(4 * N) lstm_backward_kernel(const float *__restrict__ dy, long lddy,
                                                              const float *__restrict__ gates, const float *__restrict__ cell,
                                                              const float *__restrict__ sW, const float *__restrict__ peep,
                                                              float *__restrict__ dsum, float *__restrict__ dpeep, int T, int B,
                                                              int reverse)
{
    __shared__ __attribute__((aligned(16))) float v[4 * N];
    __shared__ float part[4][N];
    const int tid = threadIdx.x, i = tid % N, q = tid / N, b = blockIdx.x;
    const bool owner = q == 0;
    float w[N];
#pragma unroll
    for (int r = 0; r < N; r++) w[r] = sW[(size_t)(q * N + r) * N + i];              // carry_out[i] = sum_r dsum[r] sW[r][i]
    const float p0 = peep ? peep[i] : 0.0f, p1 = peep ? peep[N + i] : 0.0f, p2 = peep ? peep[2 * N + i] : 0.0f;
    auto row = [&](int s) { return (size_t)(reverse ? T - 1 - s : s) * B + b; };
    float carry_out = 0.0f, carry_c = 0.0f, ap0 = 0.0f, ap1 = 0.0f, ap2 = 0.0f;
    float n_go = 0.f, n_cn = 0.f, n_cp = 0.f;
    float4 n_gt = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int s) {
        const size_t m = row(s);
        n_go = dy[m * lddy + i];
        n_gt = *reinterpret_cast<const float4 *>(gates + m * (4 * N) + 4 * i);
        n_cn = cell[m * N + i];
        n_cp = s > 0 ? cell[row(s - 1) * N + i] : 0.0f;
    };
    if (owner) fetch(T - 1);
    for (int s = T - 1; s >= 0; s--) {
        if (owner) {
            const float go = n_go + carry_out, g = n_gt.x, ig = n_gt.y, f = n_gt.z, o = n_gt.w, cn = n_cn, cp = n_cp;
            const size_t m = row(s);
            if (s > 0) fetch(s - 1);
            const float tc = slk_tanh(cn);
            const float do_pre = go * tc * o * (1.0f - o);
            const float dc = go * o * (1.0f - tc * tc) + do_pre * p2 + carry_c;
            const float di_pre = dc * g * ig * (1.0f - ig);
            const float df_pre = dc * cp * f * (1.0f - f);
            const float dg_pre = dc * ig * (1.0f - g * g);
            carry_c = dc * f + di_pre * p0 + df_pre * p1;
            ap0 += di_pre * cp; ap1 += df_pre * cp; ap2 += do_pre * cn;
            const float4 d4 = make_float4(dg_pre, di_pre, df_pre, do_pre);
            *reinterpret_cast<float4 *>(&v[4 * i]) = d4;
            *reinterpret_cast<float4 *>(dsum + m * (4 * N) + 4 * i) = d4;
        }
        __syncthreads();
        {
            float acc = 0.0f;
#pragma unroll
            for (int r = 0; r < N; r++) acc = fmaf(v[q * N + r], w[r], acc);
            part[q][i] = acc;
        }
        __syncthreads();
        if (owner) carry_out = (part[0][i] + part[1][i]) + (part[2][i] + part[3][i]);
    }
    if (owner) {
        dpeep[((size_t)b * 3 + 0) * N + i] = ap0;
        dpeep[((size_t)b * 3 + 1) * N + i] = ap1;
        dpeep[((size_t)b * 3 + 2) * N + i] = ap2;
    }
}

extern "C" int slk_lstm_backward_f32(const float *dy, long lddy, const float *gates, const float *cell, const float *sW,
                                     const float *peep, float *dsum, float *dpeep, int T, int B, int n, int reverse, int act,
                                     int gate_act, slk_stream_t stream)
{
    if (!dy || !gates || !cell || !sW || !dsum || !dpeep || T < 1 || B < 1 || n < 1 || lddy < n) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if (((reinterpret_cast<uintptr_t>(gates) | reinterpret_cast<uintptr_t>(dsum)) & 15) != 0) return SLK_ERR_INVALID_ARG;
    hipStream_t s = slk_stream(stream);
    switch (n) {
#define LSTM_BWD_CASE(NN)                                                                                                  \
    case NN:                                                                                                               \
        hipLaunchKernelGGL((lstm_backward_kernel<NN>), dim3(B), dim3(4 * NN), 0, s, dy, lddy, gates, cell, sW, peep, dsum,   \
                           dpeep, T, B, reverse);                                                                          \
        return slk_launch_status()
        LSTM_BWD_CASE(16); LSTM_BWD_CASE(32); LSTM_BWD_CASE(48); LSTM_BWD_CASE(64); LSTM_BWD_CASE(96); LSTM_BWD_CASE(128);
#undef LSTM_BWD_CASE
    default: return SLK_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Loss of train_network.py:128-136 and its gradient with respect to the logits, in place.  One wave per row (t, b):
//     p = exp(l - max) * inv_sum ; post = min_prob + (1 - min_prob) p ; row loss = -w log(post[label]) / count
//     dl_j = (w / count) (1 - min_prob) p[label] / post[label] * (p_j - [j == label])
// for drop <= t < T - drop, zero gradient and zero loss outside; count = (T - 2 drop) B (the T.mean).  correct[m] = 1
// when the first maximum of the row is the label (T.argmax/T.eq, :134).  Columns nstate..ld-1 are zeroed so the
// gradient can be contracted with a padded row length.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_xent_grad_kernel(float *__restrict__ logits, long ld,
                                                                const float *__restrict__ stats,
                                                                const int32_t *__restrict__ labels,
                                                                const float *__restrict__ weights, int T, int B, int nstate,
                                                                int drop, float min_prob, float *__restrict__ loss_rows,
                                                                float *__restrict__ correct_rows)
{
    const int lane = threadIdx.x & 63;
    const size_t m = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (size_t)T * B) return;
    const int t = (int)(m / B);
    const bool counted = t >= drop && t < T - drop;
    float *row = logits + m * ld;
    const float mx = stats[2 * m], inv = stats[2 * m + 1];
    const int label = labels[m];
    const float count = (float)(T - 2 * drop) * (float)B;
    const float p_lab = __expf(row[label] - mx) * inv;
    const float post_lab = min_prob + (1.0f - min_prob) * p_lab;
    const float w = counted ? weights[m] / count : 0.0f;
    const float coef = w * (1.0f - min_prob) * p_lab / post_lab;
    int first = 0x7fffffff;
    auto grad = [&](float l, int j) {
        float d = 0.0f;
        if (j < nstate) {
            if (l == mx && j < first) first = j;
            d = coef * (__expf(l - mx) * inv - (j == label ? 1.0f : 0.0f));
        }
        return d;
    };
    if ((ld & 3) == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) {
        // 16-byte accesses, all loads of the row in flight before the first store
        float4 *row4 = reinterpret_cast<float4 *>(row);
        const int n4 = (int)(ld >> 2);
        for (int j0 = 0; j0 < n4; j0 += 64 * 5) {
            float4 v[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int j4 = j0 + 64 * k + lane;
                v[k] = row4[min(j4, n4 - 1)];                      // clamped, not conditional: the loads stay back to back
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const int j4 = j0 + 64 * k + lane;
                if (j4 < n4)
                    row4[j4] = make_float4(grad(v[k].x, 4 * j4), grad(v[k].y, 4 * j4 + 1), grad(v[k].z, 4 * j4 + 2),
                                           grad(v[k].w, 4 * j4 + 3));
            }
        }
    } else {
        for (int j = lane; j < (int)ld; j += 64) row[j] = grad(row[j], j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
    if (lane == 0) {
        loss_rows[m] = counted ? -w * logf(post_lab) : 0.0f;
        correct_rows[m] = (counted && first == label) ? 1.0f / count : 0.0f;
    }
}

extern "C" int slk_softmax_xent_grad_f32(float *logits, long ld, const float *stats, const int32_t *labels,
                                         const float *weights, int T, int B, int nstate, int drop, float min_prob,
                                         float *loss_rows, float *correct_rows, slk_stream_t stream)
{
    if (!logits || !stats || !labels || !weights || !loss_rows || !correct_rows || T < 1 || B < 1 || nstate < 1 ||
        ld < nstate || drop < 0 || 2 * drop >= T || !(min_prob >= 0.0f && min_prob < 1.0f))
        return SLK_ERR_INVALID_ARG;
    const size_t M = (size_t)T * B;
    hipLaunchKernelGGL(softmax_xent_grad_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, slk_stream(stream), logits, ld,
                       stats, labels, weights, T, B, nstate, drop, min_prob, loss_rows, correct_rows);
    return slk_launch_status();
}

// sum of x[0..n) (square = 0) or of x^2 (square = 1: updates.param_sqr, updates.py:92-103), accumulated in float64 in a
// fixed order: one workgroup, result in out[0]
__global__ void __launch_bounds__(1024) reduce_sum_kernel(const float *__restrict__ x, size_t n, int square,
                                                          double *__restrict__ out)
{
    __shared__ double part[1024];
    double acc = 0.0;
    size_t e = threadIdx.x;
    for (; e + 7 * 1024 < n; e += 8 * 1024) {            // eight loads in flight, added in index order
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = x[e + (size_t)k * 1024];
#pragma unroll
        for (int k = 0; k < 8; k++) acc += square ? (double)v[k] * v[k] : (double)v[k];
    }
    for (; e < n; e += 1024) {
        const double v = x[e];
        acc += square ? v * v : v;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

extern "C" int slk_reduce_sum_f32(const float *x, size_t n, int square, double *out, slk_stream_t stream)
{
    if (!x || !out) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, slk_stream(stream), x, n, square, out);
    return slk_launch_status();
}

// The sums of nrow arrays x[r][0..n) in two launches: 256 workgroups per array each add a contiguous piece (thread-strided, eight loads
// in flight, float64, then the tree of reduce_sum_kernel), one workgroup per array adds the 256 pieces in a fixed tree.  Deterministic
// like reduce_sum_kernel -- whose single workgroup needs 74 us for the 819200 loss terms of a training step, twice per step.
#define RS_PIECES 256
__global__ void __launch_bounds__(256) reduce_rows_partial_kernel(const float *__restrict__ x, size_t n, double *__restrict__ scratch)
{
    __shared__ double part[256];
    const float *row = x + (size_t)blockIdx.y * n;
    const size_t per = (n + RS_PIECES - 1) / RS_PIECES, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double acc = 0.0;
    size_t e = lo + threadIdx.x;
    for (; e + 7 * 256 < hi; e += 8 * 256) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = row[e + (size_t)k * 256];
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (double)v[k];
    }
    for (; e < hi; e += 256) acc += (double)row[e];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) scratch[(size_t)blockIdx.y * RS_PIECES + blockIdx.x] = part[0];
}
__global__ void __launch_bounds__(RS_PIECES) reduce_rows_final_kernel(const double *__restrict__ scratch, double *__restrict__ out)
{
    __shared__ double part[RS_PIECES];
    part[threadIdx.x] = scratch[(size_t)blockIdx.x * RS_PIECES + threadIdx.x];
    __syncthreads();
    for (int s = RS_PIECES / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

// out[r] = sum of x[r][0..n), r < nrow <= 16; scratch: nrow * 256 doubles
extern "C" int slk_reduce_rows_sum_f32(const float *x, int nrow, size_t n, double *out, double *scratch, slk_stream_t stream)
{
    if (!x || !out || !scratch || nrow < 1 || nrow > 16 || n < 1) return SLK_ERR_INVALID_ARG;
    hipStream_t s = slk_stream(stream);
    hipLaunchKernelGGL(reduce_rows_partial_kernel, dim3(RS_PIECES, nrow), dim3(256), 0, s, x, n, scratch);
    hipLaunchKernelGGL(reduce_rows_final_kernel, dim3(nrow), dim3(RS_PIECES), 0, s, (const double *)scratch, out);
    return slk_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// C[N1][N2] = A^T B over M rows (A:[M][N1], B:[M][N2], row-major): every weight gradient of the reverse pass.
// v_mfma_f32_32x32x2_f32 contracts over k = the ROW index m, and both operands want "32 consecutive columns of one row
// per half-wave": exactly what row-major A and B give, so operands go from global memory to the MFMA without LDS or
// transposes.  One wave per workgroup owns a 96 x 96 block of C (3 x 3 accumulators) for one slice of rows and writes a
// partial block; tn_reduce_kernel adds the slices in a fixed order (deterministic, unlike atomics).  The slice height
// is chosen per call so that a small C still gives every SIMD of the chip a wave or two (tn_slice_rows).
// ---------------------------------------------------------------------------------------------------------------
#define TN_ROWS 8192           /* tallest slice */
#define TN_MIN_ROWS 256
#define TN_WAVES 2048          /* waves wanted per launch: 256 CUs x 4 SIMDs x 2 */
#define TN_BLK 96
#define TN_UNROLL 4
#ifndef TN_DEAL
#define TN_DEAL 6              /* vector instructions asked for behind every MFMA of gemm_tn_bf16_kernel */
#endif

// Which block of C and which slice of rows this workgroup takes.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and
// b + 8 share one; observed, speed only), each with an L2 of its own, and the g1 * g2 column blocks of ONE slice read the same rows
// of A and B: numbered along the launch order they would sit on 8 different XCDs and every operand row would cross the fabric up to
// 8 times.  Launch index l -> work item (l % 8) * ceil(W / 8) + l / 8: consecutive work items (the blocks of a slice) run on the
// same XCD one after the other and find each other's operand rows in its L2.
__device__ __forceinline__ bool tn_block_of(int g1, int g2, int nslice, int &bx, int &by, int &bz)
{
    const long W = (long)g1 * g2 * nslice, per_xcd = (W + 7) / 8;
    const long l = blockIdx.x;
    const long work = (l & 7) * per_xcd + (l >> 3);
    if (work >= W) return false;
    bx = (int)(work % g1);
    const long rest = work / g1;
    bz = (int)(rest % g2);
    by = (int)(rest / g2);
    return true;
}

// CS: also produce the column sums of A (the bias gradient, da^T 1) from the operands already in registers
template <bool CS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) gemm_tn_kernel(const float *__restrict__ A, long lda, const float *__restrict__ Bm,
                                                     long ldb, float *__restrict__ partial, long M, int N1, int N2,
                                                     float *__restrict__ cs_partial, int slice_rows, int g1, int g2, int nslice)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    int bx, by, bz;
    if (!tn_block_of(g1, g2, nslice, bx, by, bz)) return;
    const int n1_0 = bx * TN_BLK, n2_0 = bz * TN_BLK;
    const long m_lo = (long)by * slice_rows, m_hi = min(m_lo + slice_rows, M);
    const int ta = min(3, (N1 - n1_0 + 31) / 32), tb = min(3, (N2 - n2_0 + 31) / 32);    // live 32-wide tiles (uniform)
    const float *pa[3], *pb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        pa[i] = A + min(n1_0 + 32 * i + r, N1 - 1);
        pb[i] = Bm + min(n2_0 + 32 * i + r, N2 - 1);
    }
    f32x16 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    float cs[3] = {0.0f, 0.0f, 0.0f};

    // two register buffers of 2*TN_UNROLL rows each: the loads of the next block are in flight while the matrix pipe
    // works through the current one (one wave per SIMD at this register footprint, so the overlap has to be in-wave).
    // Rows past the slice are read from a clamped (valid) address and zeroed when consumed -- masking at load time
    // would make the wave wait for the load immediately.
    float a0[TN_UNROLL][3], b0[TN_UNROLL][3], a1[TN_UNROLL][3], b1[TN_UNROLL][3];
    auto load = [&](float (&av)[TN_UNROLL][3], float (&bv)[TN_UNROLL][3], long m0) {
#pragma unroll
        for (int u = 0; u < TN_UNROLL; u++) {
            const long mc = min(m0 + 2 * u + h, M - 1);
#pragma unroll
            for (int i = 0; i < 3; i++) {
                av[u][i] = pa[i][mc * lda];               // (unconditional: see gemm_tn_bf16_kernel)
                bv[u][i] = pb[i][mc * ldb];
            }
        }
    };
    auto mma = [&](float (&av)[TN_UNROLL][3], float (&bv)[TN_UNROLL][3], long m0) {
#pragma unroll
        for (int u = 0; u < TN_UNROLL; u++) {
            const float live = (m0 + 2 * u + h) < m_hi ? 1.0f : 0.0f;
            float a[3];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                a[i] = av[u][i] * live;
                if (CS) cs[i] += a[i];
            }
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (i < ta && j < tb) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[u][j], acc[i][j], 0, 0, 0);
        }
    };
    constexpr int STEP = 2 * TN_UNROLL;
    load(a0, b0, m_lo);
    for (long m0 = m_lo; m0 < m_hi; m0 += 2 * STEP) {
        load(a1, b1, m0 + STEP);
        __builtin_amdgcn_sched_barrier(0);
        mma(a0, b0, m0);
        __builtin_amdgcn_sched_barrier(0);
        load(a0, b0, m0 + 2 * STEP);
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1, m0 + STEP);
        __builtin_amdgcn_sched_barrier(0);
    }
    // partial[slice][N1][N2]; D[row = (e&3) + 8*(e>>2) + 4*h][col = r]; columns clamped above are dropped here
    float *out = partial + (size_t)by * N1 * N2;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int col = n2_0 + 32 * j + r;
            if (i < ta && j < tb && col < N2) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int rowc = n1_0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (rowc < N1) out[(size_t)rowc * N2 + col] = acc[i][j][e];
                }
            }
        }
    if (CS && bz == 0) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float tot = cs[i] + __shfl_xor(cs[i], 32);
            const int colc = n1_0 + 32 * i + r;
            if (h == 0 && i < ta && colc < N1) cs_partial[(size_t)by * N1 + colc] = tot;
        }
    }
}

// The same contraction on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate): every float32 operand is
// cut into three bf16 pieces v = b1 + b2 + b3 (8 significand bits each: the top 16 bits of v, of v - b1, of v - b1 - b2 --
// exact, and bf16 has float32's exponent range, so no scaling is needed and gradients of 1e-30 are as safe as 1e+3) and a
// product is the six terms b1c1 + (b1c2 + b2c1) + (b1c3 + b2c2 + b3c1) in float32 accumulators; what is dropped is below
// 2^-24 of the product.  A lane supplies 8 consecutive ROWS m of one column per operand tile -- again straight from global
// memory (a half-wave reads 32 consecutive columns of one row per load).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_bf16x3(const float (&v)[8], bf16x8 &p1, bf16x8 &p2, bf16x8 &p3)
{
    union { bf16x8 v; unsigned u[4]; } o1, o2, o3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // a piece = the top 16 bits of what is left; v_perm_b32 packs the top halves of a pair of values in one instruction (no
        // masking, shifting and or-ing of the pieces themselves: 11 instructions per pair instead of 16)
        float r1[2], r2[2];
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const float x = v[2 * j + e];
            r1[e] = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
            r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);
        }
        o1.u[j] = __builtin_amdgcn_perm(__float_as_uint(v[2 * j + 1]), __float_as_uint(v[2 * j]), 0x07060302u);
        o2.u[j] = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
        o3.u[j] = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
    }
    p1 = o1.v; p2 = o2.v; p3 = o3.v;
}

// one 96 x 96 block (bx, bz) of C for slice `by` of the rows; CS: the column sums of A as well (cs_partial may still be null)
template <bool CS>
__device__ __forceinline__ void tn_bf16_block(const float *__restrict__ A, long lda, const float *__restrict__ Bm, long ldb,
                                              float *__restrict__ partial, long M, int N1, int N2, float *__restrict__ cs_partial,
                                              int slice_rows, int bx, int by, int bz)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int n1_0 = bx * TN_BLK, n2_0 = bz * TN_BLK;
    const long m_lo = (long)by * slice_rows, m_hi = min(m_lo + slice_rows, M);
    const int ta = min(3, (N1 - n1_0 + 31) / 32), tb = min(3, (N2 - n2_0 + 31) / 32);    // live 32-wide tiles (uniform)
    // Addresses: the slice's first row as a uniform base, the rest as a 32-bit byte offset per lane -- row within the slice (clamped
    // to the last row of the matrix) times the row length (v_mad_u32_u24) plus the lane's column: 64 vector instructions per block
    // where 64-bit row * length + column arithmetic took ~190 (the launcher checks that the offsets fit).
    const char *const Abase = reinterpret_cast<const char *>(A + m_lo * lda), *const Bbase = reinterpret_cast<const char *>(Bm + m_lo * ldb);
    const unsigned lastrel = (unsigned)(M - 1 - m_lo), lda4 = (unsigned)lda * 4u, ldb4 = (unsigned)ldb * 4u;
    unsigned ca[3], cb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ca[i] = (unsigned)min(n1_0 + 32 * i + r, N1 - 1) * 4u;
        cb[i] = (unsigned)min(n2_0 + 32 * i + r, N2 - 1) * 4u;
    }
    f32x16 acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    float cs[3] = {0.0f, 0.0f, 0.0f};
    // one K block = 16 rows: this lane holds rows m0 + 8h + (0..7) of its column in each of the six operand tiles.  Two
    // register buffers: the loads of the next block are in flight while the matrix pipe works through the current one.
    float a0[3][8], b0[3][8], a1[3][8], b1[3][8];
    auto load = [&](float (&av)[3][8], float (&bv)[3][8], long m0) __attribute__((always_inline)) {
        const unsigned rel0 = (unsigned)(m0 - m_lo) + 8u * h;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned rel = min(rel0 + j, lastrel);
#pragma unroll
            for (int i = 0; i < 3; i++) {                // (unconditional, from clamped rows and columns: a load inside a branch --
                // even a wave-uniform one -- makes the compiler wait for EVERYTHING in flight before the next use, the prefetch included)
                av[i][j] = *reinterpret_cast<const float *>(Abase + (__umul24(rel, lda4) + ca[i]));
                bv[i][j] = *reinterpret_cast<const float *>(Bbase + (__umul24(rel, ldb4) + cb[i]));
            }
        }
    };
    // A block is cut into pieces by ~310 vector instructions.  Round 3 cut a block completely and then multiplied it, smallest terms
    // first -- the first product needs the LAST piece -- so the matrix pipe idled through the cutting (MfmaUtil 33 %); and the test on
    // the live tile counts, uniform as it is, put every MFMA into a basic block of its own with nothing schedulable in between.
    // Now the terms run largest first, each level of pieces is cut under the MFMAs of the level before, and the first level of the
    // NEXT block (one v_perm per pair of values) under the last MFMAs of this one:
    //     9 MFMAs  a1.b1             | v -= top(v): second level          (the raw values are peeled in place)
    //    18 MFMAs  a1.b2, a2.b1      | v -= top(v): third level
    //              -- the raw registers are free: request the block after next into them --
    //    27 MFMAs  a2.b2, a1.b3, a3.b1 | rows past the slice -> 0, column sums, first level of the next block
    // (The running sums already hold the blocks before: which term of a block is added first no longer decides what is absorbed.)
    struct Pieces { bf16x8 a1[3], a2[3], a3[3], b1[3], b2[3], b3[3]; };
    auto top = [](const float (&v)[8]) __attribute__((always_inline)) {                 // the top 16 bits of eight values: one v_perm_b32 per pair
        union { bf16x8 v; unsigned u[4]; } o;
#pragma unroll
        for (int j = 0; j < 4; j++) o.u[j] = __builtin_amdgcn_perm(__float_as_uint(v[2 * j + 1]), __float_as_uint(v[2 * j]), 0x07060302u);
        return o.v;
    };
    auto peel = [](float (&v)[8]) __attribute__((always_inline)) {                      // what is left below the top 16 bits (exact)
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = v[j] - __uint_as_float(__float_as_uint(v[j]) & 0xffff0000u);
    };
    auto prep = [&](float (&av)[3][8], float (&bv)[3][8], long m0, Pieces &P) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                av[i][j] = (m0 + 8 * h + j) < m_hi ? av[i][j] : 0.0f;       // rows past the slice (read from a clamped address)
                if (CS) cs[i] += av[i][j];
            }
            P.a1[i] = top(av[i]);
            P.b1[i] = top(bv[i]);
        }
    };
    auto level = [&](float (&av)[3][8], float (&bv)[3][8], bf16x8 (&pa)[3], bf16x8 (&pb)[3]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            peel(av[i]);
            peel(bv[i]);
            pa[i] = top(av[i]);
            pb[i] = top(bv[i]);
        }
    };
#define TN_TERM(PA, PB)                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 3; i++) _Pragma("unroll") for (int j = 0; j < 3; j++)                \
        if (FULL || (i < ta && j < tb)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(P.PA[i], P.PB[j], acc[i][j], 0, 0, 0);
    // one block: its values in (av, bv), its first-level pieces in P; the next block's values in (an, bn) have arrived by the third
    // phase and get their first-level pieces into Q; (av, bv) are requested again for the block at m_next
    auto block = [&](float (&av)[3][8], float (&bv)[3][8], Pieces &P, float (&an)[3][8], float (&bn)[3][8], Pieces &Q, long m_this,
                     auto fullc) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(fullc)::value;
        TN_TERM(a1, b1)
        level(av, bv, P.a2, P.b2);
        if constexpr (FULL) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
            }
        }
        TN_TERM(a1, b2) TN_TERM(a2, b1)
        level(av, bv, P.a3, P.b3);
        if constexpr (FULL) {
#pragma unroll
            for (int i = 0; i < 18; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);               // (the requests must not rise above the last use of the registers they fill)
        TN_TERM(a2, b2) TN_TERM(a1, b3) TN_TERM(a3, b1)
        load(av, bv, m_this + 2 * 16);
        prep(an, bn, m_this + 16, Q);
        if constexpr (FULL) {
#pragma unroll
            for (int i = 0; i < 24; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // two of the 48 requests
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
#pragma unroll
            for (int i = 0; i < 3; i++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int STEP = 16;
    Pieces P0, P1;
    load(a0, b0, m_lo);
    load(a1, b1, m_lo + STEP);
    prep(a0, b0, m_lo, P0);
    auto run = [&](auto fullc) __attribute__((always_inline)) {
        for (long m0 = m_lo; m0 < m_hi; m0 += 2 * STEP) {
            block(a0, b0, P0, a1, b1, P1, m0, fullc);
            block(a1, b1, P1, a0, b0, P0, m0 + STEP, fullc);
        }
    };
    if (ta == 3 && tb == 3) run(std::true_type{});
    else run(std::false_type{});
#undef TN_TERM
    float *out = partial + (size_t)by * N1 * N2;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int col = n2_0 + 32 * j + r;
            if (i < ta && j < tb && col < N2) {
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const int rowc = n1_0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (rowc < N1) out[(size_t)rowc * N2 + col] = acc[i][j][e];
                }
            }
        }
    if (CS && bz == 0 && cs_partial) {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float tot = cs[i] + __shfl_xor(cs[i], 32);
            const int colc = n1_0 + 32 * i + r;
            if (h == 0 && i < ta && colc < N1) cs_partial[(size_t)by * N1 + colc] = tot;
        }
    }
}

template <bool CS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_tn_bf16_kernel(const float *__restrict__ A, long lda, const float *__restrict__ Bm,
                                                          long ldb, float *__restrict__ partial, long M, int N1, int N2,
                                                          float *__restrict__ cs_partial, int slice_rows, int g1, int g2, int nslice)
{
    int bx, by, bz;
    if (!tn_block_of(g1, g2, nslice, bx, by, bz)) return;
    tn_bf16_block<CS>(A, lda, Bm, ldb, partial, M, N1, N2, cs_partial, slice_rows, bx, by, bz);
}

// Several contractions over the SAME rows in one launch (the three weight gradients of a Gru layer all contract da -- with x, with
// h(t-1), with r * h(t-1) -- and as three launches da crossed the memory bus twice): the blocks of all problems for one slice of rows are
// consecutive work items, so they run at the same time on one XCD and whichever reads a row of an operand first brings it into that
// XCD's L2 for the others.
#define TN_MAXPROB 4
struct TnProblems {
    const float *A[TN_MAXPROB], *B[TN_MAXPROB];
    float *partial[TN_MAXPROB], *cs_partial[TN_MAXPROB];
    long lda[TN_MAXPROB], ldb[TN_MAXPROB];
    int N1[TN_MAXPROB], N2[TN_MAXPROB], g1[TN_MAXPROB], first[TN_MAXPROB + 1];      // first[q]: work item of problem q's first block within a slice
    int nprob;
};
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) gemm_tn_bf16_multi_kernel(TnProblems pr, long M, int slice_rows, int nslice)
{
    const int per_slice = pr.first[pr.nprob];
    const long W = (long)per_slice * nslice, per_xcd = (W + 7) / 8;
    const long l = blockIdx.x, work = (l & 7) * per_xcd + (l >> 3);         // tn_block_of's dealing over the XCDs
    if (work >= W) return;
    const int by = (int)(work / per_slice), rem = (int)(work % per_slice);
    int q = 0;
#pragma unroll
    for (int k = 1; k < TN_MAXPROB; k++) q = (k < pr.nprob && rem >= pr.first[k]) ? k : q;
    const int local = rem - pr.first[q], bx = local % pr.g1[q], bz = local / pr.g1[q];
    tn_bf16_block<true>(pr.A[q], pr.lda[q], pr.B[q], pr.ldb[q], pr.partial[q], M, pr.N1[q], pr.N2[q], pr.cs_partial[q], slice_rows, bx, by, bz);
}

// C = sum over slices of partial, in a fixed order: 64 outputs x 16 slice groups per workgroup; group sg adds slices
// sg, sg+16, ... (eight loads in flight), then the 16 group sums are added in group order.
__global__ void __launch_bounds__(1024) tn_reduce_kernel(const float *__restrict__ partial, int nslice, int N1, int N2,
                                                         float *__restrict__ C, long ldc)
{
    __shared__ double part[16][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + o, total = (size_t)N1 * N2;
    double acc = 0.0;
    if (e < total) {
        int s = sg;
        for (; s + 7 * 16 < nslice; s += 8 * 16) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = partial[(size_t)(s + 16 * k) * total + e];
#pragma unroll
            for (int k = 0; k < 8; k++) acc += v[k];
        }
        for (; s < nslice; s += 16) acc += partial[(size_t)s * total + e];
    }
    part[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && e < total) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) tot += part[k][o];
        C[(e / N2) * ldc + (e % N2)] = (float)tot;
    }
}

// several of those sums in one launch (slk_gemm_tn_multi_bf16x6_f32: up to four products and their column sums)
struct TnReduceJobs {
    const float *partial[2 * TN_MAXPROB];
    float *C[2 * TN_MAXPROB];
    long ldc[2 * TN_MAXPROB];
    int N1[2 * TN_MAXPROB], N2[2 * TN_MAXPROB], first[2 * TN_MAXPROB + 1];      // first[k]: first workgroup of job k
    int njob;
};
__global__ void __launch_bounds__(1024) tn_reduce_jobs_kernel(TnReduceJobs jobs, int nslice)
{
    __shared__ double part[16][64];
    int k = 0;
#pragma unroll
    for (int q = 1; q < 2 * TN_MAXPROB; q++) k = (q < jobs.njob && (int)blockIdx.x >= jobs.first[q]) ? q : k;
    const float *partial = jobs.partial[k];
    const int N2 = jobs.N2[k];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const size_t e = (size_t)((int)blockIdx.x - jobs.first[k]) * 64 + o, total = (size_t)jobs.N1[k] * N2;
    double acc = 0.0;
    if (e < total) {                                     // (the order of tn_reduce_kernel)
        int s = sg;
        for (; s + 7 * 16 < nslice; s += 8 * 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = partial[(size_t)(s + 16 * j) * total + e];
#pragma unroll
            for (int j = 0; j < 8; j++) acc += v[j];
        }
        for (; s < nslice; s += 16) acc += partial[(size_t)s * total + e];
    }
    part[sg][o] = acc;
    __syncthreads();
    if (sg == 0 && e < total) {
        double tot = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) tot += part[j][o];
        jobs.C[k][(e / N2) * jobs.ldc[k] + (e % N2)] = (float)tot;
    }
}

static int tn_multi_slice_rows(long M, long blocks);
static int tn_slice_rows(long M, int N1, int N2)
{
    const long blocks = (long)((N1 + TN_BLK - 1) / TN_BLK) * ((N2 + TN_BLK - 1) / TN_BLK);
    return tn_multi_slice_rows(M, blocks);
}

extern "C" size_t slk_gemm_tn_workspace_bytes(long M, int N1, int N2)
{
    if (M < 1 || N1 < 1 || N2 < 1) return 0;
    const int rows = tn_slice_rows(M, N1, N2);
    return (size_t)((M + rows - 1) / rows) * N1 * ((size_t)N2 + 1) * sizeof(float);
}

static int gemm_tn_launch(bool bf16, const float *A, long lda, const float *B, long ldb, float *C, long ldc, long M, int N1, int N2,
                          float *colsum, void *workspace, size_t workspace_bytes, slk_stream_t stream)
{
    if (!A || !B || !C || M < 1 || N1 < 1 || N2 < 1 || lda < N1 || ldb < N2 || ldc < N2) return SLK_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < slk_gemm_tn_workspace_bytes(M, N1, N2)) return SLK_ERR_WORKSPACE;
    const int rows = tn_slice_rows(M, N1, N2);
    const long nslice = (M + rows - 1) / rows;
    const int g1 = (N1 + TN_BLK - 1) / TN_BLK, g2 = (N2 + TN_BLK - 1) / TN_BLK;
    const long W = (long)g1 * g2 * nslice;
    if (W > 0x7ffffff0L) return SLK_ERR_UNSUPPORTED;
    hipStream_t s = slk_stream(stream);
    float *partial = (float *)workspace, *cs_partial = partial + (size_t)nslice * N1 * N2;
    const dim3 grid((unsigned)(8 * ((W + 7) / 8)));                       // tn_block_of: 8 XCDs x ceil(W / 8) work items
#define TN_LAUNCH(K) hipLaunchKernelGGL(K, grid, dim3(64), 0, s, A, lda, B, ldb, partial, M, N1, N2, cs_partial, rows, g1, g2, (int)nslice)
    // the bf16 kernel's 32-bit byte offsets within a slice (24-bit row length): rows too long for them take the float32 kernel
    const unsigned long long ldmax = (unsigned long long)(lda > ldb ? lda : ldb) * 4ull;
    if (ldmax >= (1ull << 24) || ((unsigned long long)rows + 64ull) * ldmax + ldmax >= (1ull << 32)) bf16 = false;
    if (bf16) {
        if (colsum) TN_LAUNCH(gemm_tn_bf16_kernel<true>);
        else TN_LAUNCH(gemm_tn_bf16_kernel<false>);
    } else {
        if (colsum) TN_LAUNCH(gemm_tn_kernel<true>);
        else TN_LAUNCH(gemm_tn_kernel<false>);
    }
#undef TN_LAUNCH
    const size_t total = (size_t)N1 * N2;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, s, (const float *)partial,
                       (int)nslice, N1, N2, C, ldc);
    if (colsum)
        hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)((N1 + 63) / 64)), dim3(1024), 0, s, (const float *)cs_partial,
                           (int)nslice, N1, 1, colsum, 1L);
    return slk_launch_status();
}

extern "C" int slk_gemm_tn_f32(const float *A, long lda, const float *B, long ldb, float *C, long ldc, long M, int N1, int N2,
                               float *colsum, void *workspace, size_t workspace_bytes, slk_stream_t stream)
{
    return gemm_tn_launch(false, A, lda, B, ldb, C, ldc, M, N1, N2, colsum, workspace, workspace_bytes, stream);
}

// the same with every product as six bf16 terms (float32-grade, see gemm_tn_bf16_kernel); same workspace
extern "C" int slk_gemm_tn_bf16x6_f32(const float *A, long lda, const float *B, long ldb, float *C, long ldc, long M, int N1,
                                      int N2, float *colsum, void *workspace, size_t workspace_bytes, slk_stream_t stream)
{
    return gemm_tn_launch(true, A, lda, B, ldb, C, ldc, M, N1, N2, colsum, workspace, workspace_bytes, stream);
}

// nprob (1..4) contractions over the same M rows in one launch: C[q] = A[q]^T B[q], colsum[q] (or NULL) = A[q]^T 1 (arrays of nprob
// entries in HOST memory).  Six bf16 terms per product as slk_gemm_tn_bf16x6_f32, same results.
static long tn_multi_blocks(int nprob, const int *N1, const int *N2)
{
    long blocks = 0;
    for (int q = 0; q < nprob; q++) blocks += (long)((N1[q] + TN_BLK - 1) / TN_BLK) * ((N2[q] + TN_BLK - 1) / TN_BLK);
    return blocks;
}
static int tn_multi_slice_rows(long M, long blocks)
{
    long rows = M * blocks / TN_WAVES;
    rows = (rows + 31) / 32 * 32;
    rows = rows < TN_MIN_ROWS ? TN_MIN_ROWS : (rows > TN_ROWS ? TN_ROWS : rows);
    // not one work item more than the waves wanted: 2052 items on 1024 wave slots are THREE rounds, the last one of four waves (the six
    // blocks of a Gru layer at M = 819200 had exactly that: train_wgrad 4.1 -> 3.4 ms per step)
    while (rows < TN_ROWS && blocks * ((M + rows - 1) / rows) > TN_WAVES) rows += 32;
    return (int)rows;
}

extern "C" size_t slk_gemm_tn_multi_workspace_bytes(long M, int nprob, const int *N1, const int *N2)
{
    if (M < 1 || nprob < 1 || nprob > TN_MAXPROB || !N1 || !N2) return 0;
    for (int q = 0; q < nprob; q++)
        if (N1[q] < 1 || N2[q] < 1) return 0;
    const int rows = tn_multi_slice_rows(M, tn_multi_blocks(nprob, N1, N2));
    const size_t nslice = (size_t)((M + rows - 1) / rows);
    size_t floats = 0;
    for (int q = 0; q < nprob; q++) floats += nslice * N1[q] * ((size_t)N2[q] + 1);
    return floats * sizeof(float);
}

extern "C" int slk_gemm_tn_multi_bf16x6_f32(int nprob, const float *const *A, const long *lda, const float *const *B, const long *ldb,
                                            float *const *C, const long *ldc, long M, const int *N1, const int *N2, float *const *colsum,
                                            void *workspace, size_t workspace_bytes, slk_stream_t stream)
{
    if (nprob < 1 || nprob > TN_MAXPROB || !A || !lda || !B || !ldb || !C || !ldc || !N1 || !N2 || M < 1) return SLK_ERR_INVALID_ARG;
    for (int q = 0; q < nprob; q++)
        if (!A[q] || !B[q] || !C[q] || N1[q] < 1 || N2[q] < 1 || lda[q] < N1[q] || ldb[q] < N2[q] || ldc[q] < N2[q]) return SLK_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < slk_gemm_tn_multi_workspace_bytes(M, nprob, N1, N2)) return SLK_ERR_WORKSPACE;
    const long blocks = tn_multi_blocks(nprob, N1, N2);
    const int rows = tn_multi_slice_rows(M, blocks);
    const long nslice = (M + rows - 1) / rows, W = blocks * nslice;
    if (W > 0x7ffffff0L) return SLK_ERR_UNSUPPORTED;
    TnProblems pr;
    float *ws = (float *)workspace;
    pr.nprob = nprob;
    pr.first[0] = 0;
    for (int q = 0; q < TN_MAXPROB; q++) {
        const int k = q < nprob ? q : 0;
        const unsigned long long ldmax = (unsigned long long)(lda[k] > ldb[k] ? lda[k] : ldb[k]) * 4ull;
        if (ldmax >= (1ull << 24) || ((unsigned long long)rows + 64ull) * ldmax + ldmax >= (1ull << 32)) return SLK_ERR_UNSUPPORTED;   // 32-bit offsets
        pr.A[q] = A[k]; pr.B[q] = B[k]; pr.lda[q] = lda[k]; pr.ldb[q] = ldb[k]; pr.N1[q] = N1[k]; pr.N2[q] = N2[k];
        pr.g1[q] = (N1[k] + TN_BLK - 1) / TN_BLK;
        if (q < nprob) {
            pr.partial[q] = ws;
            ws += (size_t)nslice * N1[q] * N2[q];
            pr.cs_partial[q] = (colsum && colsum[q]) ? ws : nullptr;
            ws += (size_t)nslice * N1[q];
            pr.first[q + 1] = pr.first[q] + pr.g1[q] * ((N2[q] + TN_BLK - 1) / TN_BLK);
        } else {
            pr.partial[q] = nullptr; pr.cs_partial[q] = nullptr;
            pr.first[q + 1] = pr.first[q];
        }
    }
    pr.first[TN_MAXPROB] = pr.first[nprob];
    hipStream_t s = slk_stream(stream);
    hipLaunchKernelGGL(gemm_tn_bf16_multi_kernel, dim3((unsigned)(8 * ((W + 7) / 8))), dim3(64), 0, s, pr, M, rows, (int)nslice);
    // the sums over the slices of every problem (and of its column sums) in one launch
    TnReduceJobs jobs;
    int nj = 0, blk = 0;
    for (int q = 0; q < nprob; q++) {
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && !pr.cs_partial[q]) continue;
            jobs.partial[nj] = pass ? pr.cs_partial[q] : pr.partial[q];
            jobs.C[nj] = pass ? colsum[q] : C[q];
            jobs.N1[nj] = N1[q];
            jobs.N2[nj] = pass ? 1 : N2[q];
            jobs.ldc[nj] = pass ? 1L : ldc[q];
            jobs.first[nj] = blk;
            blk += (int)(((size_t)N1[q] * (pass ? 1 : N2[q]) + 63) / 64);
            nj++;
        }
    }
    for (int k = nj; k < 2 * TN_MAXPROB; k++) { jobs.partial[k] = nullptr; jobs.C[k] = nullptr; jobs.N1[k] = jobs.N2[k] = 0; jobs.ldc[k] = 0; jobs.first[k] = blk; }
    jobs.first[2 * TN_MAXPROB] = blk;
    jobs.njob = nj;
    hipLaunchKernelGGL(tn_reduce_jobs_kernel, dim3((unsigned)blk), dim3(1024), 0, s, jobs, (int)nslice);
    return slk_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// dL/d(pre-activation) = dL/dy * fun'(.) written in terms of the layer OUTPUT y (what the forward pass keeps):
// tanh 1-y^2, sigmoid y(1-y), linear 1, relu [y>0], elu (y>0 ? 1 : y+1)    (activation.py:8-57)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) act_backward_kernel(const float *__restrict__ dy, const float *__restrict__ y,
                                                           float *__restrict__ out, size_t n, int act)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float v = y[e];
        float d;
        switch (act) {
        case SLK_ACT_TANH: d = 1.0f - v * v; break;
        case SLK_ACT_SIGMOID: d = v * (1.0f - v); break;
        case SLK_ACT_RELU: d = v > 0.0f ? 1.0f : 0.0f; break;
        case SLK_ACT_ELU: d = v > 0.0f ? 1.0f : v + 1.0f; break;
        default: d = 1.0f; break;
        }
        out[e] = dy[e] * d;
    }
}

extern "C" int slk_act_backward_f32(const float *dy, const float *y, float *out, size_t n, int act, slk_stream_t stream)
{
    if (!dy || !y || !out) return SLK_ERR_INVALID_ARG;
    if (act != SLK_ACT_TANH && act != SLK_ACT_SIGMOID && act != SLK_ACT_RELU && act != SLK_ACT_ELU && act != SLK_ACT_LINEAR)
        return SLK_ERR_UNSUPPORTED;
    if (n == 0) return SLK_OK;
    hipLaunchKernelGGL(act_backward_kernel, dim3(elementwise_grid(n)), dim3(256), 0, slk_stream(stream), dy, y, out, n, act);
    return slk_launch_status();
}

// y += x (the contributions of the branches of a Parallel layer to dL/d(input): layers.py:1486-1487 concatenates forward)
__global__ void __launch_bounds__(256) add_inplace_kernel(float *__restrict__ y, const float *__restrict__ x, size_t n)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) y[e] += x[e];
}

extern "C" int slk_add_inplace_f32(float *y, const float *x, size_t n, slk_stream_t stream)
{
    if (!y || !x) return SLK_ERR_INVALID_ARG;
    if (n == 0) return SLK_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(elementwise_grid(n)), dim3(256), 0, slk_stream(stream), y, x, n);
    return slk_launch_status();
}

// Window rows of a one-feature convolution (conv.py:66-111 with insize 1): cols[(t*B + b)][k] = x(b, t*stride + k - pad_lo),
// zero outside the signal; x addressed as x[t*x_t_stride + b*x_b_stride] like slk_conv1d_f32.  dL/dW = dpre^T cols.
__global__ void __launch_bounds__(256) im2col_cin1_kernel(const float *__restrict__ x, long xts, long xbs, int T, int B,
                                                          int Tout, int winlen, int stride, int pad_lo,
                                                          float *__restrict__ cols)
{
    const size_t total = (size_t)Tout * B * winlen;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t m = e / winlen;
        const int k = (int)(e - m * winlen), t = (int)(m / B), b = (int)(m - (size_t)t * B);
        const int src = t * stride + k - pad_lo;
        cols[e] = (src >= 0 && src < T) ? x[(size_t)src * xts + (size_t)b * xbs] : 0.0f;
    }
}

extern "C" int slk_train_im2col_cin1_f32(const float *x, long x_t_stride, long x_b_stride, int T, int B, int winlen,
                                         int stride, int pad_lo, int pad_hi, float *cols, slk_stream_t stream)
{
    if (!x || !cols || T < 1 || B < 1 || winlen < 1 || stride < 1 || pad_lo < 0 || pad_hi < 0) return SLK_ERR_INVALID_ARG;
    const int Tout = slk_conv1d_out_len(T, winlen, stride, pad_lo, pad_hi);
    if (Tout < 1) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(im2col_cin1_kernel, dim3(elementwise_grid((size_t)Tout * B * winlen)), dim3(256), 0,
                       slk_stream(stream), x, x_t_stride, x_b_stride, T, B, Tout, winlen, stride, pad_lo, cols);
    return slk_launch_status();
}

// The same for any number of input features: x:[T][B][Cin] (rows of Cin floats, ldx apart); cols[(t*B + b)][c*winlen + k] =
// x(t*stride + k - pad_lo, b, c), zero outside the signal -- the column order of Convolution.W:[Cout][Cin][winlen] flattened, so
// dL/dW = dpre^T cols and dL/dcols = dpre . W.
__global__ void __launch_bounds__(256) im2col_kernel(const float *__restrict__ x, long ldx, int T, int B, int Cin, int Tout,
                                                     int winlen, int stride, int pad_lo, float *__restrict__ cols)
{
    const int K = Cin * winlen;
    const size_t total = (size_t)Tout * B * K;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t m = e / K;
        const int ck = (int)(e - m * K), c = ck / winlen, k = ck - c * winlen;
        const int t = (int)(m / B), b = (int)(m - (size_t)t * B);
        const int src = t * stride + k - pad_lo;
        cols[e] = (src >= 0 && src < T) ? x[((size_t)src * B + b) * ldx + c] : 0.0f;
    }
}

// ... and its adjoint: dx(t', b, c) = sum over the windows (t, k) that cover sample t' of dcols[(t*B + b)][c*winlen + k], as a
// gather in a fixed order (deterministic; at most ceil(winlen / stride) terms)
__global__ void __launch_bounds__(256) col2im_kernel(const float *__restrict__ dcols, int T, int B, int Cin, int Tout, int winlen,
                                                     int stride, int pad_lo, float *__restrict__ dx, long lddx)
{
    const int K = Cin * winlen;
    const size_t total = (size_t)T * B * Cin;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % Cin);
        const size_t tb = e / Cin;
        const int b = (int)(tb % B), tp = (int)(tb / B);
        float acc = 0.0f;
        for (int k = 0; k < winlen; k++) {
            const int num = tp + pad_lo - k;
            if (num < 0) break;                               // larger k only makes it smaller
            if (num % stride) continue;
            const int t = num / stride;
            if (t < Tout) acc += dcols[((size_t)t * B + b) * K + c * winlen + k];
        }
        dx[((size_t)tp * B + b) * lddx + c] = acc;
    }
}

extern "C" int slk_train_im2col_f32(const float *x, long ldx, int T, int B, int Cin, int winlen, int stride, int pad_lo,
                                    int pad_hi, float *cols, slk_stream_t stream)
{
    if (!x || !cols || T < 1 || B < 1 || Cin < 1 || ldx < Cin || winlen < 1 || stride < 1 || pad_lo < 0 || pad_hi < 0)
        return SLK_ERR_INVALID_ARG;
    const int Tout = slk_conv1d_out_len(T, winlen, stride, pad_lo, pad_hi);
    if (Tout < 1) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(im2col_kernel, dim3(elementwise_grid((size_t)Tout * B * Cin * winlen)), dim3(256), 0, slk_stream(stream), x,
                       ldx, T, B, Cin, Tout, winlen, stride, pad_lo, cols);
    return slk_launch_status();
}

extern "C" int slk_train_col2im_f32(const float *dcols, int T, int B, int Cin, int winlen, int stride, int pad_lo, int pad_hi,
                                    float *dx, long lddx, slk_stream_t stream)
{
    if (!dcols || !dx || T < 1 || B < 1 || Cin < 1 || lddx < Cin || winlen < 1 || stride < 1 || pad_lo < 0 || pad_hi < 0)
        return SLK_ERR_INVALID_ARG;
    const int Tout = slk_conv1d_out_len(T, winlen, stride, pad_lo, pad_hi);
    if (Tout < 1) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(col2im_kernel, dim3(elementwise_grid((size_t)T * B * Cin)), dim3(256), 0, slk_stream(stream), dcols, T, B,
                       Cin, Tout, winlen, stride, pad_lo, dx, lddx);
    return slk_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// updates.adam ("ADAMski", updates.py:77-87) over flat buffers; lr_t and momentum_decay are this step's scalars
// (updates.py:73-76, computed on the host in float32 like the reference's shared variables).  `l2` adds the gradient of
// the penalty l2 * param_sqr (train_network.py:130), `gscale` scales the incoming gradient (1/world after a sum
// all-reduce); the clip is applied to the result, as th.grad of the whole loss is clipped in the reference.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adamski_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                      float *__restrict__ momentum, float *__restrict__ variance, size_t n,
                                                      float lr_t, float momentum_decay, float decay1, float decay2,
                                                      float epsilon, float clip, float l2, float gscale)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float p = param[e];
        const float g = slk_clip(grad[e] * gscale + 2.0f * l2 * p, -clip, clip);
        const float mo = momentum_decay * momentum[e] + (1.0f - decay1) * g;
        const float va = decay2 * variance[e] + (1.0f - decay2) * g * g;
        momentum[e] = mo;
        variance[e] = va;
        param[e] = p - lr_t * mo / (sqrtf(va) + epsilon);
    }
}

extern "C" int slk_adamski_update_f32(float *param, const float *grad, float *momentum, float *variance, size_t n, float lr_t,
                                      float momentum_decay, float decay1, float decay2, float epsilon, float clip, float l2,
                                      float gscale, slk_stream_t stream)
{
    if (!param || !grad || !momentum || !variance) return SLK_ERR_INVALID_ARG;
    if (n == 0) return SLK_OK;
    hipLaunchKernelGGL(adamski_kernel, dim3(elementwise_grid(n)), dim3(256), 0, slk_stream(stream), param, grad, momentum,
                       variance, n, lr_t, momentum_decay, decay1, decay2, epsilon, clip, l2, gscale);
    return slk_launch_status();
}

// updates.sgd (updates.py:9-33): vel = momentum vel - rate clip(g) ; param += vel
__global__ void __launch_bounds__(256) sgd_kernel(float *__restrict__ param, const float *__restrict__ grad,
                                                  float *__restrict__ vel, size_t n, float rate, float momentum, float clip,
                                                  float l2, float gscale)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float p = param[e];
        const float g = slk_clip(grad[e] * gscale + 2.0f * l2 * p, -clip, clip);
        const float v = momentum * vel[e] - rate * g;
        vel[e] = v;
        param[e] = p + v;
    }
}

extern "C" int slk_sgd_update_f32(float *param, const float *grad, float *vel, size_t n, float rate, float momentum,
                                  float clip, float l2, float gscale, slk_stream_t stream)
{
    if (!param || !grad || !vel || momentum < 0.0f) return SLK_ERR_INVALID_ARG;
    if (n == 0) return SLK_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(elementwise_grid(n)), dim3(256), 0, slk_stream(stream), param, grad, vel, n, rate,
                       momentum, clip, l2, gscale);
    return slk_launch_status();
}
