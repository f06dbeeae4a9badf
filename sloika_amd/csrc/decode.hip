// decode.hip -- k-mer transducer decoding on gfx950.
//
//   decode.prepare_post   sloika/decode.py:21-36
//   decode.viterbi        sloika/decode.py:39-93     (forward DP :60-82, backtrace :84-91)
//   decode.argmax         sloika/decode.py:5-18
//
// This is integer/float32 max-plus dynamic programming: HBM- and latency-bound, no MFMA.  One 256-thread
// workgroup walks one chunk through time with the score vector in LDS.  Thread j owns the `nbase` to-states
// that share the same step predecessor set {a*nkmer/nbase + j}; the 16 (nbase^2) skip predecessors are reduced
// in two levels (step maxima first), which preserves numpy's first-maximum tie-breaking because the combined
// key (value desc, a*nbase+b asc) is what np.argmax over the reshaped (nbase^2, nkmer/nbase^2) view selects.
// Float arithmetic is max / add / sub in float32 only (no contraction possible), so scores are bit-identical
// to numpy's given the same log-posteriors.  The traceback (reference: an int32 from-state per (t, state)) is stored as ONE BYTE
// per (t, state) by the generic kernel:
//   0..nbase-1 = step from a, nbase..nbase+nbase^2-1 = skip from ab, 255 = stay
// and as ONE 16-BIT WORD per (t, four states) by the two nbase-4 kernels (see viterbi_forward4_kernel), and walked by a second
// kernel that stages time-blocks of it in LDS.
#include "common.h"
#include "decode_internal.h"

#define VIT_ETA 1e-10f
#define VIT_STAY 255

// exact restatement of numpy's float32 evaluation; __fmul_rn/__fadd_rn stop hipcc fusing the pair into an fma
__device__ __forceinline__ float prepare_post_val(float p, float min_prob, float one_m)
{
    return __fadd_rn(min_prob, __fmul_rn(one_m, p));       // decode.py:36
}

// natural log through v_log_f32 (log2) * ln2: ~1e-7 relative, 2 instructions instead of ~20 -- the decoder evaluates
// it once per (t, chunk, state), i.e. 840 M times per B=1024 batch
__device__ __forceinline__ float fast_logf(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

__device__ __forceinline__ float log_post_val(float p, int mode, float min_prob, float one_m)
{
    if (mode == SLK_POST_LOG) return p;
    if (mode == SLK_POST_LN) return fast_logf(p);          // transducer.py:30
    if (mode == SLK_POST_RAW) p = prepare_post_val(p, min_prob, one_m);
    return fast_logf(__fadd_rn(p, VIT_ETA));               // decode.py:56
}

// SLK_POST_LOGITS: the posterior is rebuilt from (logit, row max, 1/row sum) exactly as softmax_rows_kernel writes it
// (sloika/layers.py:311-314), then treated as SLK_POST_RAW.
__device__ __forceinline__ float log_logit_val(float l, float2 st, float min_prob, float one_m)
{
    const float p = __expf(l - st.x) * st.y;
    return fast_logf(__fadd_rn(prepare_post_val(p, min_prob, one_m), VIT_ETA));
}

// The same transforms on TWO values at a time: element-wise vector arithmetic compiles to v_pk_mul_f32 / v_pk_add_f32 (one
// instruction for both lanes of the pair, each component rounded exactly like the scalar instruction; no contraction:
// -ffp-contract=off), only exp and log stay scalar.  Bit-identical to log_post_val / log_logit_val by construction; the
// decoder's vector work per state drops by a fifth.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fast_logf2(f32x2 x)
{
    const f32x2 l = {__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
    return l * 0.6931471805599453f;
}
__device__ __forceinline__ f32x2 prepare_post_val2(f32x2 p, float min_prob, float one_m) { return min_prob + one_m * p; }
__device__ __forceinline__ f32x2 log_post_val2(f32x2 p, int mode, float min_prob, float one_m)
{
    if (mode == SLK_POST_LOG) return p;
    if (mode == SLK_POST_LN) return fast_logf2(p);
    if (mode == SLK_POST_RAW) p = prepare_post_val2(p, min_prob, one_m);
    return fast_logf2(p + VIT_ETA);
}
__device__ __forceinline__ f32x2 log_logit_val2(f32x2 l, float2 st, float min_prob, float one_m)
{
    const f32x2 a = (l - st.x) * 1.4426950408889634f;              // __expf(y) = v_exp_f32(y * log2 e)
    const f32x2 e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
    const f32x2 p = e * st.y;
    return fast_logf2(prepare_post_val2(p, min_prob, one_m) + VIT_ETA);
}

__global__ void log_post_kernel(const float *__restrict__ post, float *__restrict__ lpost, size_t count, int mode,
                                float min_prob, float one_m)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        lpost[i] = log_post_val(post[i], mode, min_prob, one_m);
}

__global__ void log_logits_kernel(const float *__restrict__ logits, long ld, const float2 *__restrict__ stats,
                                  float *__restrict__ lpost, size_t rows, int nst, float min_prob, float one_m)
{
    const size_t count = rows * nst;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / nst;
        lpost[i] = log_logit_val(logits[row * ld + (i - row * nst)], stats[row], min_prob, one_m);
    }
}

__global__ void prepare_post_kernel(const float *__restrict__ post, float *__restrict__ out, size_t count,
                                    float min_prob, float one_m)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        out[i] = prepare_post_val(post[i], min_prob, one_m);
}

static inline unsigned grid_for(size_t count)
{
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (unsigned)(blocks ? blocks : 1);
}

// (1.0 - min_prob) is evaluated in double by the reference and THEN cast to float32 (decode.py:36)
static inline float one_minus(float min_prob_f, double min_prob_d) { (void)min_prob_f; return (float)(1.0 - min_prob_d); }

extern "C" int slk_log_post_f32(const float *post, float *lpost, size_t count, int input_mode, float min_prob,
                                slk_stream_t stream)
{
    if (!post || !lpost || input_mode < 0 || input_mode > SLK_POST_LN) return SLK_ERR_INVALID_ARG;
    if (!count) return SLK_OK;
    hipLaunchKernelGGL(log_post_kernel, dim3(grid_for(count)), dim3(256), 0, slk_stream(stream), post, lpost, count,
                       input_mode, min_prob, one_minus(min_prob, (double)min_prob));
    return slk_launch_status();
}

extern "C" int slk_log_post_logits_f32(const float *logits, long ld, const float *stats, float *lpost, size_t rows,
                                       int nstate, float min_prob, slk_stream_t stream)
{
    if (!logits || !stats || !lpost || nstate < 1 || ld < nstate) return SLK_ERR_INVALID_ARG;
    if (!rows) return SLK_OK;
    hipLaunchKernelGGL(log_logits_kernel, dim3(grid_for(rows * nstate)), dim3(256), 0, slk_stream(stream), logits, ld,
                       reinterpret_cast<const float2 *>(stats), lpost, rows, nstate, min_prob,
                       one_minus(min_prob, (double)min_prob));
    return slk_launch_status();
}

extern "C" int slk_prepare_post_f32(const float *post, float *out, size_t count, float min_prob, slk_stream_t stream)
{
    if (!post || !out) return SLK_ERR_INVALID_ARG;
    if (!count) return SLK_OK;
    hipLaunchKernelGGL(prepare_post_kernel, dim3(grid_for(count)), dim3(256), 0, slk_stream(stream), post, out, count,
                       min_prob, one_minus(min_prob, (double)min_prob));
    return slk_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// forward DP
// ------------------------------------------------------------------------------------------------------
template <int NB, bool LOGITS>
__global__ void __launch_bounds__(1024) viterbi_forward_kernel(const float *__restrict__ post,
                                                               const float2 *__restrict__ stats, long ld, int T, int B,
                                                               int nkmer, float skip_pen, int mode, float min_prob,
                                                               float one_m, uint8_t *__restrict__ tb,
                                                               int32_t *__restrict__ best_out,
                                                               float *__restrict__ score_out, const int *__restrict__ lens)
{
    constexpr int NB2 = NB * NB;
    extern __shared__ float sm[];
    const int Tpad = T;                                        // row stride of the traceback; T = this chunk's own length
    if (lens) T = min(max(lens[blockIdx.x], 1), Tpad);
    const int nrem1 = nkmer / NB, nrem2 = nkmer / NB2;
    float *v = sm;                                             // [nkmer] scores of the previous step
    float *stepmax = sm + nkmer;                               // [nrem1]
    int *steparg = reinterpret_cast<int *>(stepmax + nrem1);   // [nrem1]
    float *redv = reinterpret_cast<float *>(steparg + nrem1);  // [16]
    int *redi = reinterpret_cast<int *>(redv + 16);            // [16]
    const int b = blockIdx.x, j = threadIdx.x;                 // thread j owns to-states j*NB .. j*NB+NB-1
    const bool active = j < nrem1;
    const int jj = active ? j : 0, j2 = jj / NB;
    const float *pb = post + (size_t)b * ld;            // rows (t, b) are `ld` floats apart (ld >= nkmer + 1)
    const size_t tstride = (size_t)B * ld;
    uint8_t *tbb = tb + (size_t)b * Tpad * nkmer;

    auto xform = [&](float val, float2 st) {
        return LOGITS ? log_logit_val(val, st, min_prob, one_m) : log_post_val(val, mode, min_prob, one_m);
    };
    // t = 0: v = lpost[0][1:]   (decode.py:57)
    float raw[NB], raw0;
    float2 rst = make_float2(0.f, 1.f);
    if (LOGITS) rst = stats[b];
#pragma unroll
    for (int c = 0; c < NB; c++) raw[c] = pb[1 + jj * NB + c];
    if (active) {
#pragma unroll
        for (int c = 0; c < NB; c++) v[jj * NB + c] = xform(raw[c], rst);
    }
    if (T > 1) {
#pragma unroll
        for (int c = 0; c < NB; c++) raw[c] = pb[tstride + 1 + jj * NB + c];
        raw0 = pb[tstride];
        if (LOGITS) rst = stats[(size_t)B + b];
    }
    __syncthreads();

    for (int t = 1; t < T; t++) {
        // prefetch the next row one full step ahead of its use
        float nxt[NB], nxt0 = 0.0f;
        float2 nst2 = make_float2(0.f, 1.f);
        if (t + 1 < T) {
            const float *pn = pb + (size_t)(t + 1) * tstride;
#pragma unroll
            for (int c = 0; c < NB; c++) nxt[c] = pn[1 + jj * NB + c];
            nxt0 = pn[0];
            if (LOGITS) nst2 = stats[(size_t)(t + 1) * B + b];
        }
        // ---- step maximum over a (first max wins: np.argmax, decode.py:67-68) ----
        float sstep = v[jj];
        int sarg = 0;
#pragma unroll
        for (int a = 1; a < NB; a++) {
            float c = v[a * nrem1 + jj];
            if (c > sstep) { sstep = c; sarg = a; }
        }
        if (active) { stepmax[jj] = sstep; steparg[jj] = sarg; }
        float lp[NB];
#pragma unroll
        for (int c = 0; c < NB; c++) lp[c] = xform(raw[c], rst);
        const float lp0 = xform(raw0, rst);
        __syncthreads();
        // ---- skip maximum over ab = a*NB + b (first max in ab order wins, decode.py:72-73) ----
        float kbest = stepmax[j2];
        int karg = steparg[j2] * NB;
#pragma unroll
        for (int bb = 1; bb < NB; bb++) {
            float c = stepmax[bb * nrem2 + j2];
            int key = steparg[bb * nrem2 + j2] * NB + bb;
            if (c > kbest || (c == kbest && key < karg)) { kbest = c; karg = key; }
        }
        const float sskip = kbest - skip_pen;                       // decode.py:72
        const float mx = fmaxf(sstep, sskip);
        const int code = sstep > sskip ? sarg : NB + karg;          // decode.py:76 (tie -> skip)
        if (active) {
            uint8_t codes[NB];
#pragma unroll
            for (int c = 0; c < NB; c++) {
                const int s = jj * NB + c;
                const float nv = lp[c] + mx;                        // decode.py:75
                const float stay = v[s] + lp0;                      // decode.py:80
                const bool move = nv > stay;                        // decode.py:81 (tie -> stay)
                codes[c] = move ? (uint8_t)code : (uint8_t)VIT_STAY;
                v[s] = move ? nv : stay;
            }
            uint8_t *dst = tbb + (size_t)t * nkmer + (size_t)jj * NB;
            if constexpr (NB == 4) {
                *reinterpret_cast<uint32_t *>(dst) = (uint32_t)codes[0] | ((uint32_t)codes[1] << 8) |
                                                     ((uint32_t)codes[2] << 16) | ((uint32_t)codes[3] << 24);
            } else {
#pragma unroll
                for (int c = 0; c < NB; c++) dst[c] = codes[c];
            }
        }
#pragma unroll
        for (int c = 0; c < NB; c++) raw[c] = nxt[c];
        raw0 = nxt0;
        rst = nst2;
        __syncthreads();
    }
    // ---- first argmax of v (np.argmax, decode.py:85) ----
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (active) {
#pragma unroll
        for (int c = 0; c < NB; c++) {
            float x = v[jj * NB + c];
            if (x > bv) { bv = x; bi = jj * NB + c; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = bv; redi[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; w++)
            if (redv[w] > bv || (redv[w] == bv && redi[w] < bi)) { bv = redv[w]; bi = redi[w]; }
        score_out[b] = bv;
        best_out[b] = bi;
    }
}

// ------------------------------------------------------------------------------------------------------
// forward DP, nbase = 4 (the DNA/RNA models): one barrier per step.
//   * scores ping-pong between two LDS vectors, so a step needs a single barrier (the generic kernel above publishes
//     the step maxima through LDS and needs two);
//   * thread j = 4q + c owns to-states 4j..4j+3.  Their 16 skip predecessors (a, b) are shared by the quad q: thread c
//     reduces the four with b = c, then the quad combines the partial (value, key = a*4 + b) pairs with two DPP
//     exchanges -- same first-maximum rule as np.argmax over the flattened (a, b) axis (decode.py:72-73);
//   * the blank column's log-posterior is the same for every thread: it is evaluated once per workgroup, a block of
//     blockDim.x steps at a time, instead of once per thread and step.
// ------------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) unaligned_f4 { float x, y, z, w; };
#define VIT_TBS 16

template <bool LOGITS>
__global__ void __launch_bounds__(1024) viterbi_forward4_kernel(const float *__restrict__ post,
                                                                const float2 *__restrict__ stats, long ld, int T, int B,
                                                                int nkmer, float skip_pen, int mode, float min_prob,
                                                                float one_m, uint8_t *__restrict__ tb,
                                                                int32_t *__restrict__ best_out,
                                                                float *__restrict__ score_out, const int *__restrict__ lens)
{
    constexpr int NB = 4;
    extern __shared__ float sm[];
    const int Tpad = T;                                        // row stride of the traceback; T = this chunk's own length
    if (lens) T = min(max(lens[blockIdx.x], 1), Tpad);
    const int nrem1 = nkmer / 4, nrem2 = nkmer / 16;
    const int b = blockIdx.x, j = threadIdx.x, nt = blockDim.x;
    float *vbuf0 = sm, *vbuf1 = sm + nkmer;                    // scores of even / odd steps
    float *lp0buf = sm + 2 * nkmer;                            // [nt] blank log-posteriors of the current block of steps
    float *redv = lp0buf + nt;                                 // [16]
    int *redi = reinterpret_cast<int *>(redv + 16);            // [16]
    // Traceback, PACKED: the four to-states of a thread share their step predecessor a and the sixteen of a quad their skip
    // predecessor (a, b), so a thread's step is 14 bits -- 2 bits per state (0 stay, 1 step, 2 skip), a in bits 8-9, a*4+b in
    // bits 10-13 -- one 16-bit word instead of four bytes: half the traceback traffic of a kernel that is HBM bound.
    // VIT_TBS steps are staged in LDS as an image of the global layout and copied out with 16-byte stores: a store every step
    // would make the loop-top wait for the prefetched row (vmcnt) also wait for that store
    uint16_t *tbs = reinterpret_cast<uint16_t *>(redi + 16);   // [VIT_TBS][nrem1] half words
    const bool active = j < nrem1;
    const int jj = active ? j : 0, q = jj >> 2, c = jj & 3;
    const float *pb = post + (size_t)b * ld;                   // rows (t, b) are `ld` floats apart (ld >= nkmer + 1)
    const size_t tstride = (size_t)B * ld;
    uint8_t *tbb = tb + (size_t)b * Tpad * (nkmer / 2);         // 2 bytes per thread (4 states) and step

    auto xform = [&](float val, float2 st) {
        return LOGITS ? log_logit_val(val, st, min_prob, one_m) : log_post_val(val, mode, min_prob, one_m);
    };
    auto xform2 = [&](f32x2 val, float2 st) {
        return LOGITS ? log_logit_val2(val, st, min_prob, one_m) : log_post_val2(val, mode, min_prob, one_m);
    };
    auto load_row = [&](int t) { return *reinterpret_cast<const unaligned_f4 *>(pb + (size_t)t * tstride + 1 + 4 * jj); };
    auto row_stats = [&](int t) { return LOGITS ? stats[(size_t)t * B + b] : make_float2(0.f, 1.f); };
    // blank column of step t0 + j (one step per thread)
    auto load_blank = [&](int t0, float &r0, float2 &st0) {
        const int tt = min(t0 + j, T - 1);
        r0 = pb[(size_t)tt * tstride];
        st0 = row_stats(tt);
    };

    // t = 0: v = lpost[0][1:]   (decode.py:57)
    unaligned_f4 raw = load_row(0);
    float2 rst = row_stats(0);
    float blank_raw;
    float2 blank_st;
    load_blank(0, blank_raw, blank_st);
    if (active)
        *reinterpret_cast<float4 *>(&vbuf0[4 * jj]) = make_float4(xform(raw.x, rst), xform(raw.y, rst), xform(raw.z, rst), xform(raw.w, rst));
    lp0buf[j] = xform(blank_raw, blank_st);
    if (T > nt) load_blank(nt, blank_raw, blank_st);           // next block, converted when it starts
    // Two row buffers that swap roles statically (time loop unrolled by two): copying a just-requested row into the
    // "current" registers would make every step wait for that request.  During step t `use` holds row t, `fill` gets t+1.
    // (requesting rows two steps ahead with a third buffer was measured: 1.064 against 1.041 ms -- memory latency is not what
    // the step waits for)
    unaligned_f4 rowA = raw, rowB = raw;
    float2 stA = rst, stB = rst;
    if (T > 1) { rowA = load_row(1); stA = row_stats(1); }
    __syncthreads();

    auto step = [&](int t, const unaligned_f4 &raw, const float2 &rst, unaligned_f4 &fill, float2 &st_fill) {
        const int tl = t & (nt - 1);                           // blockDim.x is a power of two here (64 .. 1024)
        if (tl == 0) {                                         // every thread is past the previous block (end-of-step barrier)
            lp0buf[j] = xform(blank_raw, blank_st);
            __syncthreads();
            if (t + nt < T) load_blank(t + nt, blank_raw, blank_st);
        }
        // prefetch the next row one full step ahead of its use (the tail re-reads the last row)
        fill = load_row(t + 1 < T ? t + 1 : T - 1);
        st_fill = row_stats(t + 1 < T ? t + 1 : T - 1);
        const float *vold = (t & 1) ? vbuf0 : vbuf1;
        float *vnew = (t & 1) ? vbuf1 : vbuf0;
        // ---- step maximum over a (first max wins: np.argmax, decode.py:67-68) ----
        float sstep = vold[jj];
        int sarg = 0;
#pragma unroll
        for (int a = 1; a < NB; a++) {
            const float x = vold[a * nrem1 + jj];
            if (x > sstep) { sstep = x; sarg = a; }
        }
        // ---- skip maximum over ab = a*4 + b: this thread's share is b = c ----
        float kbest = vold[c * nrem2 + q];
        int karg = c;
#pragma unroll
        for (int a = 1; a < NB; a++) {
            const float x = vold[a * nrem1 + c * nrem2 + q];
            if (x > kbest) { kbest = x; karg = a * NB + c; }
        }
        const float4 own = *reinterpret_cast<const float4 *>(&vold[4 * jj]);
        const float lp0 = lp0buf[tl];
        const f32x2 lp01 = xform2(f32x2{raw.x, raw.y}, rst), lp23 = xform2(f32x2{raw.z, raw.w}, rst);
        // quad exchange: lane ^ 1, then lane ^ 2 (first maximum in ab order wins, decode.py:72-73)
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int ctl = r == 0 ? 0xB1 : 0x4E;              // quad_perm [1,0,3,2] / [2,3,0,1]
            const float ov = r == 0 ? __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(kbest), 0xB1, 0xf, 0xf, false))
                                    : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(kbest), 0x4E, 0xf, 0xf, false));
            const int ok = r == 0 ? __builtin_amdgcn_update_dpp(0, karg, 0xB1, 0xf, 0xf, false)
                                  : __builtin_amdgcn_update_dpp(0, karg, 0x4E, 0xf, 0xf, false);
            (void)ctl;
            const bool take = (ov > kbest) | ((ov == kbest) & (ok < karg));     // branch-free
            kbest = take ? ov : kbest;
            karg = take ? ok : karg;
        }
        const float sskip = kbest - skip_pen;                       // decode.py:72
        const float mx = fmaxf(sstep, sskip);
        const uint32_t how = sstep > sskip ? 1u : 2u;               // decode.py:76 (tie -> skip)
        if (active) {
            const f32x2 nv01 = lp01 + mx, nv23 = lp23 + mx;         // decode.py:75
            const f32x2 st01 = f32x2{own.x, own.y} + lp0, st23 = f32x2{own.z, own.w} + lp0;     // decode.py:80
            const float nvv[4] = {nv01.x, nv01.y, nv23.x, nv23.y}, stv[4] = {st01.x, st01.y, st23.x, st23.y};
            float nw[4];
            uint32_t packed = ((uint32_t)sarg << 8) | ((uint32_t)karg << 10);
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const bool move = nvv[cc] > stv[cc];                // decode.py:81 (tie -> stay)
                packed |= (move ? how : 0u) << (2 * cc);
                nw[cc] = move ? nvv[cc] : stv[cc];
            }
            *reinterpret_cast<float4 *>(&vnew[4 * jj]) = make_float4(nw[0], nw[1], nw[2], nw[3]);
            tbs[(t % VIT_TBS) * nrem1 + jj] = (uint16_t)packed;
        }
        __syncthreads();
        if ((t % VIT_TBS) == VIT_TBS - 1 || t == T - 1) {
            // rows t0..t of the traceback are complete in LDS (t0 = first step of this block, >= 1)
            const int t0 = t - (t % VIT_TBS);
            const int first = t0 < 1 ? 1 : t0;
            const int nwords = (t - first + 1) * (nrem1 / 2);         // 32-bit words (nrem1 is a multiple of 16 here)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(tbs + (first - t0) * nrem1);
            uint32_t *dst = reinterpret_cast<uint32_t *>(tbb + (size_t)first * (nkmer / 2));
            if ((nrem1 & 7) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                for (int k = j; k < nwords / 4; k += nt)
                    reinterpret_cast<uint4 *>(dst)[k] = reinterpret_cast<const uint4 *>(src)[k];
            } else {
                for (int k = j; k < nwords; k += nt) dst[k] = src[k];
            }
            __syncthreads();               // the staging rows are rewritten from the next step on
        }
    };
    for (int t = 1; t < T; t += 2) {
        step(t, rowA, stA, rowB, stB);
        if (t + 1 < T) step(t + 1, rowB, stB, rowA, stA);
    }
    // ---- first argmax of v (np.argmax, decode.py:85) ----
    const float *v = ((T - 1) & 1) ? vbuf1 : vbuf0;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (active) {
#pragma unroll
        for (int cc = 0; cc < NB; cc++) {
            float x = v[jj * NB + cc];
            if (x > bv) { bv = x; bi = jj * NB + cc; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = bv; redi[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nwv = blockDim.x >> 6;
        for (int w = 1; w < nwv; w++)
            if (redv[w] > bv || (redv[w] == bv && redi[w] < bi)) { bv = redv[w]; bi = redi[w]; }
        score_out[b] = bv;
        best_out[b] = bi;
    }
}

// ------------------------------------------------------------------------------------------------------
// forward DP, nbase = 4, up to 1024 k-mers: ONE WAVE per chunk, no barriers at all.
//   Lane q owns the 16 to-states 16q..16q+15, i.e. the four step groups j = 4q..4q+3, which all share the same 16 skip
//   predecessors (a, b) -> v[a*nkmer/4 + b*nkmer/16 + q]: the skip arg-max is evaluated once per 16 states instead of
//   once per 4, the per-thread fixed work (addressing, row statistics, blank column) is amortised over 16 states, and
//   waves of different chunks drift freely, hiding each other's memory latency (4 waves per SIMD at B = 1024).
//   Scores live in LDS (one vector per wave: a wave reads every predecessor before it writes, in program order).
//   About half the vector instructions per (chunk, step) of the one-barrier workgroup kernel above.
// ------------------------------------------------------------------------------------------------------
template <bool LOGITS>
__global__ void __launch_bounds__(256) viterbi_forward4_wave_kernel(const float *__restrict__ post,
                                                                    const float2 *__restrict__ stats, long ld, int T, int B,
                                                                    int nkmer, float skip_pen, int mode, float min_prob,
                                                                    float one_m, uint8_t *__restrict__ tb,
                                                                    int32_t *__restrict__ best_out,
                                                                    float *__restrict__ score_out, const int *__restrict__ lens)
{
    extern __shared__ float sm[];
    const int Tpad = T;                                        // row stride of the traceback
    // the chunk index is wave-uniform: keeping it in a scalar register makes the row statistics scalar loads
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= B) return;                                        // whole wave leaves together; nothing below synchronises waves
    if (lens) T = min(max(lens[b], 1), Tpad);                  // ragged batch: this chunk's own length (wave-uniform)
    const int nrem1 = nkmer / 4, nrem2 = nkmer / 16;
    float *v = sm + wave * (nkmer + 64);                       // [nkmer] scores, then [64] blank log-posteriors
    float *lp0buf = v + nkmer;
    const bool active = lane < nrem2;
    const int q = active ? lane : 0;
    const float *pb = post + (size_t)b * ld;
    const size_t tstride = (size_t)B * ld;
    uint8_t *tbb = tb + (size_t)b * Tpad * (nkmer / 2);         // packed traceback, as viterbi_forward4_kernel writes it

    auto xform = [&](float val, float2 st) {
        return LOGITS ? log_logit_val(val, st, min_prob, one_m) : log_post_val(val, mode, min_prob, one_m);
    };
    auto row_stats = [&](int t) { return LOGITS ? stats[(size_t)t * B + b] : make_float2(0.f, 1.f); };
    struct row16 { unaligned_f4 g[4]; };
    auto load_row = [&](int t) {
        row16 r;
        const float *p = pb + (size_t)t * tstride + 1 + 16 * q;
#pragma unroll
        for (int i = 0; i < 4; i++) r.g[i] = *reinterpret_cast<const unaligned_f4 *>(p + 4 * i);
        return r;
    };
    auto load_blank = [&](int t0, float &r0, float2 &st0) {      // blank column of step t0 + lane
        const int tt = min(t0 + lane, T - 1);
        r0 = pb[(size_t)tt * tstride];
        st0 = row_stats(tt);
    };

    // Software pipeline: the log-posteriors of step t+1 are evaluated during step t (they do not depend on the scores,
    // so their exp/log work fills the waits of the dependent max-plus chain -- with one wave per SIMD nothing else
    // would), from a row that was requested during step t-1.
    auto xform_row = [&](const row16 &r, float2 st, float (&lp)[16]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lp[4 * i + 0] = xform(r.g[i].x, st);
            lp[4 * i + 1] = xform(r.g[i].y, st);
            lp[4 * i + 2] = xform(r.g[i].z, st);
            lp[4 * i + 3] = xform(r.g[i].w, st);
        }
    };
    // t = 0: v = lpost[0][1:]   (decode.py:57)
    float lp[16];
    {
        const row16 r0 = load_row(0);
        xform_row(r0, row_stats(0), lp);
    }
    float blank_raw;
    float2 blank_st;
    load_blank(0, blank_raw, blank_st);
    if (active) {
#pragma unroll
        for (int i = 0; i < 4; i++)
            *reinterpret_cast<float4 *>(&v[16 * q + 4 * i]) = make_float4(lp[4 * i], lp[4 * i + 1], lp[4 * i + 2], lp[4 * i + 3]);
    }
    lp0buf[lane] = xform(blank_raw, blank_st);
    if (T > 64) load_blank(64, blank_raw, blank_st);           // next block of 64 steps, converted when it starts
    // Two row buffers in rotation, WITHOUT register copies: a copy of a just-requested row would force the wave to wait
    // for that request (the whole memory latency, every step), so the time loop is unrolled by two and the buffers swap
    // roles statically.  While step t runs: `use` holds row t+1 (requested during step t-1), `fill` receives row t+2.
    row16 rowA, rowB;
    float2 stA, stB;
    if (T > 1) {
        const row16 r1 = load_row(1);
        xform_row(r1, row_stats(1), lp);
    }
    rowA = load_row(T > 2 ? 2 : T - 1);
    stA = row_stats(T > 2 ? 2 : T - 1);

    auto step = [&](int t, const row16 &use, const float2 &st_use, row16 &fill, float2 &st_fill) {
        const int tl = t & 63;
        if (tl == 0) {
            lp0buf[lane] = xform(blank_raw, blank_st);
            if (t + 64 < T) load_blank(t + 64, blank_raw, blank_st);
        }
        const int tf = t + 2 < T ? t + 2 : T - 1;               // clamped: the tail re-reads the last row
        fill = load_row(tf);
        st_fill = row_stats(tf);
        // ---- skip maximum over ab = a*4 + b, first maximum wins (decode.py:72-73) ----
        float kbest = v[q];
        int karg = 0;
#pragma unroll
        for (int ab = 1; ab < 16; ab++) {
            const float x = v[(ab >> 2) * nrem1 + (ab & 3) * nrem2 + q];
            const bool take = x > kbest;
            kbest = take ? x : kbest;
            karg = take ? ab : karg;
        }
        const float sskip = kbest - skip_pen;                       // decode.py:72
        const float lp0 = lp0buf[tl];
        // ---- the four step groups j = 4q + c ----
        float4 pred[4];                                              // pred[a] = v[a*nrem1 + 4q .. +3]
#pragma unroll
        for (int a = 0; a < 4; a++) pred[a] = *reinterpret_cast<const float4 *>(&v[a * nrem1 + 4 * q]);
        float4 own[4];
#pragma unroll
        for (int c = 0; c < 4; c++) own[c] = *reinterpret_cast<const float4 *>(&v[16 * q + 4 * c]);
        float4 out[4];
        uint32_t codes[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const float pa[4] = {c == 0 ? pred[0].x : c == 1 ? pred[0].y : c == 2 ? pred[0].z : pred[0].w,
                                 c == 0 ? pred[1].x : c == 1 ? pred[1].y : c == 2 ? pred[1].z : pred[1].w,
                                 c == 0 ? pred[2].x : c == 1 ? pred[2].y : c == 2 ? pred[2].z : pred[2].w,
                                 c == 0 ? pred[3].x : c == 1 ? pred[3].y : c == 2 ? pred[3].z : pred[3].w};
            float sstep = pa[0];                                     // first max wins: np.argmax, decode.py:67-68
            int sarg = 0;
#pragma unroll
            for (int a = 1; a < 4; a++) {
                const bool take = pa[a] > sstep;
                sstep = take ? pa[a] : sstep;
                sarg = take ? a : sarg;
            }
            const float mx = fmaxf(sstep, sskip);
            const uint32_t how = sstep > sskip ? 1u : 2u;           // decode.py:76 (tie -> skip)
            const float ownv[4] = {own[c].x, own[c].y, own[c].z, own[c].w};
            float nw[4];
            uint32_t packed = ((uint32_t)sarg << 8) | ((uint32_t)karg << 10);
#pragma unroll
            for (int cc = 0; cc < 4; cc++) {
                const float nv = lp[4 * c + cc] + mx;               // decode.py:75
                const float stay = ownv[cc] + lp0;                  // decode.py:80
                const bool move = nv > stay;                        // decode.py:81 (tie -> stay)
                packed |= (move ? how : 0u) << (2 * cc);
                nw[cc] = move ? nv : stay;
            }
            out[c] = make_float4(nw[0], nw[1], nw[2], nw[3]);
            codes[c] = packed;                                       // 16 bits: step group 4q + c
        }
        if (active) {
            // every read of the old scores above precedes these writes in program order (one wave, in-order LDS)
#pragma unroll
            for (int c = 0; c < 4; c++) *reinterpret_cast<float4 *>(&v[16 * q + 4 * c]) = out[c];
            *reinterpret_cast<uint2 *>(tbb + (size_t)t * (nkmer / 2) + 8 * (size_t)q) =
                make_uint2(codes[0] | (codes[1] << 16), codes[2] | (codes[3] << 16));
        }
        xform_row(use, st_use, lp);                                  // log-posteriors of step t+1
    };
    for (int t = 1; t < T; t += 2) {
        step(t, rowA, stA, rowB, stB);
        if (t + 1 < T) step(t + 1, rowB, stB, rowA, stA);
    }
    // ---- first argmax of v (np.argmax, decode.py:85) ----
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (active) {
#pragma unroll
        for (int cc = 0; cc < 16; cc++) {
            const float x = v[16 * q + cc];
            if (x > bv) { bv = x; bi = 16 * q + cc; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(bv, o);
        int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        score_out[b] = bv;
        best_out[b] = bi;
    }
}

// ------------------------------------------------------------------------------------------------------
// backtrace: one lane walks the byte traceback from the last step to the first (decode.py:84-91).  The walk is a chain
// of dependent 1-byte LDS reads (~120 cycles per step), so everything else must stay out of its way: waves 1-3 stream
// the traceback into a ring of VBT_RING LDS buffers of VBT_BLOCK bytes by LDS-DMA (global_load_lds), VBT_RING-1 blocks
// ahead of the walker, which hides the memory latency of a block behind the walk of the previous ones (the earlier
// 2-buffer version copied through registers and made the walker wait for a memory round trip per block).  Those waves
// issue loads only -- exactly VBT_BLOCK/1024/3 per wave and block -- so vmcnt(n) tells them which block has landed; the
// walker's wave owns the path stores.  The path is emitted right-aligned, then shifted left and padded with -1.
// ------------------------------------------------------------------------------------------------------
#define VBT_RING 3
#define VBT_BLOCK 12288            /* bytes; a multiple of 3 KiB (three stager waves) and of every 4^k k-mer count <= 4096 */
// PACKED false: a byte per state.  true: the traceback of viterbi_forward4_kernel, one 16-bit word per four states and step (two bits
// per to-state: 0 stay, 1 step, 2 skip; the step argument at bit 8, the skip argument at bit 10).  (The fused kernel of
// softmax_viterbi.hip writes a third format, one byte per four states, that viterbi_backtrace_rows_kernel below walks.)
template <int NB, bool PACKED>
__global__ void __launch_bounds__(256) viterbi_backtrace_kernel(const uint8_t *__restrict__ tb,
                                                                const int32_t *__restrict__ best, int T, int nkmer,
                                                                int tblk, int dma, int32_t *__restrict__ path_out,
                                                                int32_t *__restrict__ len_out, const int *__restrict__ lens)
{
    const int rowbytes = PACKED ? nkmer / 2 : nkmer;
    constexpr int NB2 = NB * NB;
    constexpr int PER_WAVE = VBT_BLOCK / 1024 / 3;
    const int Tpad = T;                                        // row strides of tb / path_out; T = this chunk's own length
    if (lens) T = min(max(lens[blockIdx.x], 1), Tpad);
    __shared__ __attribute__((aligned(16))) uint8_t blk[VBT_RING * VBT_BLOCK];
    __shared__ int sh_cur, sh_pos;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nrem1 = nkmer / NB, nrem2 = nkmer / NB2;
    const uint8_t *tbb = tb + (size_t)b * Tpad * rowbytes;
    int32_t *path = path_out + (size_t)b * Tpad;
    if (tid == 0) {
        sh_cur = best[b];
        sh_pos = T - 1;
        path[T - 1] = sh_cur;
    }
    // rows [t0, t1) of block i, counting blocks from the end of the chunk
    auto bounds = [&](int i, int &t0, int &t1) {
        t1 = T - i * tblk;
        t0 = max(1, t1 - tblk);
    };
    auto walk = [&](int i, const uint8_t *rows) {             // lane 0 of wave 0
        int t0, t1;
        bounds(i, t0, t1);
        // one lane walks: keeping the state in scalar registers (readfirstlane) moves the index arithmetic of this
        // dependent chain to the scalar unit
        int cur = __builtin_amdgcn_readfirstlane(sh_cur), pos = __builtin_amdgcn_readfirstlane(sh_pos);
        for (int t = t1 - 1; t >= t0; t--) {
            if constexpr (PACKED) {
                const int w = __builtin_amdgcn_readfirstlane(
                    (int)reinterpret_cast<const uint16_t *>(rows)[(t - t0) * (nkmer / 4) + (cur >> 2)]);
                const int how = (w >> (2 * (cur & 3))) & 3;        // 0 stay, 1 step, 2 skip
                if (how) {
                    cur = how == 1 ? ((w >> 8) & 3) * nrem1 + cur / NB : ((w >> 10) & 15) * nrem2 + cur / NB2;
                    path[--pos] = cur;                         // decode.py:88-90
                }
            } else {
                const int code = __builtin_amdgcn_readfirstlane((int)rows[(t - t0) * nkmer + cur]);
                if (code != VIT_STAY) {
                    cur = code < NB ? code * nrem1 + cur / NB : (code - NB) * nrem2 + cur / NB2;
                    path[--pos] = cur;                         // decode.py:88-90
                }
            }
        }
        sh_cur = cur;
        sh_pos = pos;
    };
    const int nblocks = (T - 1 + tblk - 1) / tblk;           // rows 1..T-1
    if (dma) {
        // block i -> ring slot i % VBT_RING; PER_WAVE 1-KiB instructions per stager wave, pieces past the block's end
        // re-read its first bytes into the unused tail of the slot
        auto request = [&](int i) {
            int t0, t1;
            bounds(i, t0, t1);
            const int nbytes = (t1 - t0) * rowbytes;
            const uint8_t *src = tbb + (size_t)t0 * rowbytes;
            uint8_t *dst = blk + (i % VBT_RING) * VBT_BLOCK;
#pragma unroll
            for (int k = 0; k < PER_WAVE; k++) {
                const int piece = k * 3 + (wave - 1);
                const int off = piece * 1024 + 16 * lane;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (off + 16 <= nbytes ? off : 0)),
                                                 (__attribute__((address_space(3))) void *)(dst + piece * 1024), 16, 0, 0);
            }
        };
        if (wave > 0)
            for (int i = 0; i < VBT_RING - 1 && i < nblocks; i++) request(i);
        for (int i = 0; i < nblocks; i++) {
            if (wave > 0) {
                // block i must have landed before the barrier; the (up to RING-2) younger requests stay in flight
                const int younger = min(VBT_RING - 2, nblocks - 1 - i);
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // LDS-only barrier (block i is in LDS; the walker is done with block i-1): __syncthreads() would also make
            // the walker's wave wait for its path stores to be acknowledged, a memory round trip per block
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // slot (i + RING-1) % RING = slot of block i-1: free now
            if (wave > 0 && i + VBT_RING - 1 < nblocks) request(i + VBT_RING - 1);
            if (tid == 0) walk(i, blk + (i % VBT_RING) * VBT_BLOCK);
        }
        __syncthreads();
    } else {
        // generic staging (k-mer counts that do not divide the block size, unaligned buffers): one block at a time
        for (int i = 0; i < nblocks; i++) {
            int t0, t1;
            bounds(i, t0, t1);
            __syncthreads();
            const int nbytes = (t1 - t0) * rowbytes;
            for (int k = tid; k < nbytes; k += nt) blk[k] = tbb[(size_t)t0 * rowbytes + k];
            __syncthreads();
            if (tid == 0) walk(i, blk);
        }
        __syncthreads();
    }
    // shift left by pos (dst < src, ascending blocks with a barrier between read and write are safe)
    const int pos = sh_pos, len = T - pos;
    __threadfence_block();
    __syncthreads();
    for (int base = 0; base < Tpad; base += nt) {
        int i = base + tid;
        int32_t val = (i < len) ? path[pos + i] : -1;
        __syncthreads();
        if (i < Tpad) path[i] = val;
        __syncthreads();
    }
    if (tid == 0) len_out[b] = len;
}

// ------------------------------------------------------------------------------------------------------
// Backtrace of the fused kernel's traceback of 4^5 k-mers -- ONE BYTE per four states: bits 0-3 "to-state n moves", bit 4 "by step"
// (else by skip), bits 5-6 the step argument, bit 7 bit (j & 3) of the skip argument, which the four lanes of a quad share; 256 bytes
// per step and chunk -- with the rows in REGISTERS: a row is 256 bytes = one dword per lane,
// so a wave keeps sixteen rows in sixteen registers (and the next sixteen on their way from memory) and reads "the word of state
// cur" with v_readlane instead of a dependent LDS read.  One wave per chunk, four per workgroup (= one per SIMD: as workgroups of
// one wave, four chunks landed on one SIMD and took 90 us instead of 57), no LDS, no barrier, no stager waves.
// The walk is a serial chain and what it costs is the NUMBER of instructions on it (a wave issues a scalar instruction per ~7
// cycles here, a taken branch costs ~30): the same walk in the style of the kernel above (a dependent LDS read, ~31 instructions and three taken
// branches per moving row) took 102 us.  Here
// a row is one asm statement of twenty instructions without a branch; the state is kept as the five values the next row needs
// (cur, cur >> 2, the lane cur >> 4, the byte's shift, cur & 3).  The path is collected in a register (v_writelane at lane
// pos & 63) and stored at most once per sixteen rows, right-aligned as above, then shifted left and padded with -1 by the same
// wave (program order: the loads of the shift follow the stores they read).  1024 chunks x 800 rows: 102 -> ~50 us.
// ------------------------------------------------------------------------------------------------------
#define VBR_ROWS 16
__global__ void __launch_bounds__(256) viterbi_backtrace_rows_kernel(const uint8_t *__restrict__ tb, const int32_t *__restrict__ best,
                                                                    int T, int B, int nkmer, int32_t *path_out,
                                                                    int32_t *__restrict__ len_out, const int *__restrict__ lens)
{
    // four chunks per workgroup, one per wave: the four waves of a workgroup go to the four SIMDs of a CU
    const int b = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const int Tpad = T;                                        // row strides of tb / path_out; T = this chunk's own length
    if (lens) T = min(max(lens[b], 1), Tpad);
    T = __builtin_amdgcn_readfirstlane(T);
    const int rowdw = nkmer / 16;                              // dwords per row: 64
    const uint32_t *tbb = reinterpret_cast<const uint32_t *>(tb + (size_t)b * Tpad * (nkmer / 4));
    int32_t *path = path_out + (size_t)b * Tpad;
    unsigned ucur = (unsigned)__builtin_amdgcn_readfirstlane(best[b]);
    unsigned u2 = ucur >> 2, idx = ucur >> 4, b8 = (ucur << 1) & 24u, n = ucur & 3u;
    int pos = T - 1;                                           // path[pos .. T) is known
    int done = T;                                              // path[done .. T) is in memory
    int buf = (int)ucur;                                       // lane (p & 63): path[p], pos <= p < done (at most 64 of them)
    auto flush = [&]() {
        const int p = pos + ((lane - pos) & 63);
        if (p < done) path[p] = buf;
        done = pos;
    };
    // every lane loads, every row: rows below 1 and lanes past the row are clamped to valid addresses (conditional loads would
    // cost the compiler its count of the loads in flight: it then waits for the block it has just asked for)
    const int lcl = lane < rowdw ? lane : rowdw - 1;
    const int tlo = Tpad > 1 ? 1 : 0;                          // (a traceback of ONE row has no row 1: nothing is walked, row 0 is what there is to read)
    auto load = [&](uint32_t (&r)[VBR_ROWS], int ttop) {
#pragma unroll
        for (int k = 0; k < VBR_ROWS; k++) r[k] = tbb[(size_t)max(ttop - k, tlo) * rowdw + lcl];
    };
    auto row = [&](uint32_t rk) {                              // decode.py:84-91, one row
        unsigned q, t, g, a;
        // No branch: both targets are computed and one selected (a staying row writes path[pos] = cur once more).  Measured against
        // a version that branches around the 16 instructions of a move (four instructions and a taken branch per staying row):
        // 57 / 56 us (random weights / blank-dominated, 740 / 349 moves in 800 rows) against 62 / 51 -- a taken branch costs what
        // ten scalar instructions do.  1024 k-mers: the step argument (bits 5-6) goes to bits 8-9, the skip argument (bit 7 of the
        // four bytes) to bits 6-9 -- the high half of a product with 2^31 + 2^24 + 2^17 + 2^10 (sixteen distinct partial products).
        asm volatile("v_readlane_b32 %[q], %[r], %[idx]\n\t"
                     "s_lshr_b32 %[t], %[q], %[b8]\n\t"                 // the byte of cur's group of four at bit 0
                     "s_and_b32 %[g], %[q], 0x80808080\n\t"
                     "s_and_b32 %[a], %[t], 0x60\n\t"
                     "s_mul_hi_u32 %[g], %[g], 0x81020400\n\t"
                     "s_lshl3_add_u32 %[a], %[a], %[u2]\n\t"            // by step: (arg << 8) + (cur >> 2)
                     "s_and_b32 %[g], %[g], 0x3c0\n\t"
                     "s_add_u32 %[g], %[g], %[idx]\n\t"                 // by skip: (arg << 6) + (cur >> 4)
                     "s_bitcmp1_b32 %[t], 4\n\t"
                     "s_cselect_b32 %[a], %[a], %[g]\n\t"
                     "s_bitcmp1_b32 %[t], %[n]\n\t"
                     "s_cselect_b32 %[cur], %[a], %[cur]\n\t"
                     "s_subb_u32 %[pos], %[pos], 0\n\t"                 // pos -= moves
                     "s_lshr_b32 %[idx], %[cur], 4\n\t"
                     "s_mov_b32 m0, %[pos]\n\t"
                     "s_lshl_b32 %[b8], %[cur], 1\n\t"
                     "s_lshr_b32 %[u2], %[cur], 2\n\t"
                     "v_writelane_b32 %[buf], %[cur], m0\n\t"
                     "s_and_b32 %[b8], %[b8], 24\n\t"
                     "s_and_b32 %[n], %[cur], 3"
                     : [q] "=&s"(q), [t] "=&s"(t), [g] "=&s"(g), [a] "=&s"(a), [cur] "+s"(ucur), [u2] "+s"(u2), [idx] "+s"(idx),
                       [b8] "+s"(b8), [n] "+s"(n), [pos] "+s"(pos), [buf] "+v"(buf)
                     : [r] "v"(rk)
                     : "scc");        // and m0, which hipcc will not take as a clobber (reserved); nothing else in this kernel uses it
                                      // (no LDS-DMA, no s_movrel, no message): tests/test_isa_hygiene.py::test_backtrace_rows_kernel_owns_m0
    };
    uint32_t ra[VBR_ROWS], rb[VBR_ROWS];
    int ttop = T - 1;
    load(ra, ttop);
    load(rb, ttop - VBR_ROWS);
    while (ttop >= 2 * VBR_ROWS) {                             // rows ttop .. ttop - 31 are all >= 1
#pragma unroll
        for (int k = 0; k < VBR_ROWS; k++) row(ra[k]);
        load(ra, ttop - 2 * VBR_ROWS);
        if (done - pos > 64 - VBR_ROWS) flush();
#pragma unroll
        for (int k = 0; k < VBR_ROWS; k++) row(rb[k]);
        load(rb, ttop - 3 * VBR_ROWS);
        if (done - pos > 64 - VBR_ROWS) flush();
        ttop -= 2 * VBR_ROWS;
    }
#pragma unroll
    for (int k = 0; k < VBR_ROWS; k++)
        if (ttop - k >= 1) row(ra[k]);
    if (done - pos > 64 - VBR_ROWS) flush();
#pragma unroll
    for (int k = 0; k < VBR_ROWS; k++)
        if (ttop - VBR_ROWS - k >= 1) row(rb[k]);
    flush();
    // shift left by pos, pad with -1: eight batches of 64 entries are read, then written (dst <= src, ascending: a group reads
    // nothing an earlier one wrote; one memory round trip per group instead of one per batch)
    const int len = T - pos;
    for (int base = 0; base < Tpad; base += 8 * 64) {
        int32_t val[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = base + 64 * k + lane;
            val[k] = path[pos + min(i, len - 1)];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = base + 64 * k + lane;
            if (i < Tpad) path[i] = i < len ? val[k] : -1;
        }
    }
    if (lane == 0) len_out[b] = len;
}

static bool vit_dims(int nbase, int klen, int *nkmer_out)
{
    if (klen < 3 || nbase < 2) return false;
    long nk = 1;
    for (int i = 0; i < klen; i++) {
        nk *= nbase;
        if (nk > 4096) return false;     // one thread per step-predecessor group, <= 1024 threads
    }
    *nkmer_out = (int)nk;
    return true;
}

extern "C" size_t slk_viterbi_kmer_workspace_bytes(int T, int B, int nbase, int klen)
{
    int nkmer;
    if (T < 1 || B < 1 || !vit_dims(nbase, klen, &nkmer)) return 0;
    size_t tb = ((size_t)B * T * nkmer + 255) & ~(size_t)255;
    return tb + sizeof(int32_t) * (size_t)B + 256;
}

template <int NB>
static int launch_viterbi(const float *post, const float *stats, long ld, int T, int B, int klen, int nkmer, float skip_pen, int mode, float min_prob,
                          uint8_t *tb, int32_t *best, float *score_out, int32_t *path_out, int32_t *len_out,
                          const int *lens, hipStream_t s)
{
    int nrem1 = nkmer / NB;
    int threads = nrem1 < 64 ? 64 : (nrem1 > 1024 ? 1024 : ((nrem1 + 63) / 64) * 64);
    if (nrem1 > 1024) return SLK_ERR_UNSUPPORTED;
    size_t lds = sizeof(float) * ((size_t)nkmer + nrem1 + 16) + sizeof(int) * ((size_t)nrem1 + 16);
    float one_m = (float)(1.0 - (double)min_prob);
    bool packed_tb = false;                                  // which traceback format the forward kernel writes
    if (NB == 4 && nkmer <= 1024 && nkmer >= 64 && B >= 1536) {
        packed_tb = true;
        // one wave per chunk, four chunks per workgroup: fewer instructions per (chunk, step), but a lone wave per SIMD
        // cannot hide its own dependent-issue and memory latency -- it wins once there are ~2 chunks per SIMD (1024 SIMDs)
        const size_t ldsw = sizeof(float) * 4 * ((size_t)nkmer + 64);
        const dim3 grid((B + 3) / 4), block(256);
        if (stats)
            hipLaunchKernelGGL((viterbi_forward4_wave_kernel<true>), grid, block, ldsw, s, post,
                               reinterpret_cast<const float2 *>(stats), ld, T, B, nkmer, skip_pen, mode, min_prob, one_m, tb,
                               best, score_out, lens);
        else
            hipLaunchKernelGGL((viterbi_forward4_wave_kernel<false>), grid, block, ldsw, s, post, nullptr, ld, T, B, nkmer,
                               skip_pen, mode, min_prob, one_m, tb, best, score_out, lens);
    } else if constexpr (NB == 4) {
        packed_tb = true;
        const size_t lds4 = sizeof(float) * (2 * (size_t)nkmer + threads + 32 + (size_t)VIT_TBS * nrem1);
        if (stats)
            hipLaunchKernelGGL((viterbi_forward4_kernel<true>), dim3(B), dim3(threads), lds4, s, post,
                               reinterpret_cast<const float2 *>(stats), ld, T, B, nkmer, skip_pen, mode, min_prob, one_m, tb,
                               best, score_out, lens);
        else
            hipLaunchKernelGGL((viterbi_forward4_kernel<false>), dim3(B), dim3(threads), lds4, s, post, nullptr, ld, T, B,
                               nkmer, skip_pen, mode, min_prob, one_m, tb, best, score_out, lens);
    } else if (stats)
        hipLaunchKernelGGL((viterbi_forward_kernel<NB, true>), dim3(B), dim3(threads), lds, s, post,
                           reinterpret_cast<const float2 *>(stats), ld, T, B, nkmer, skip_pen, mode, min_prob, one_m, tb, best,
                           score_out, lens);
    else
        hipLaunchKernelGGL((viterbi_forward_kernel<NB, false>), dim3(B), dim3(threads), lds, s, post, nullptr, ld, T, B, nkmer,
                           skip_pen, mode, min_prob, one_m, tb, best, score_out, lens);
    int rc = slk_launch_status();
    if (rc != SLK_OK) return rc;
    const int rowbytes = packed_tb ? nkmer / 2 : nkmer;
    int tblk = VBT_BLOCK / rowbytes;                         // rows per staged block
    if (tblk < 1) return SLK_ERR_UNSUPPORTED;
    const int dma = (VBT_BLOCK % rowbytes == 0) && (rowbytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(tb) & 15) == 0);
    if (tblk > T) tblk = T;
    if (packed_tb)
        hipLaunchKernelGGL((viterbi_backtrace_kernel<NB, true>), dim3(B), dim3(256), 0, s, tb, best, T, nkmer, tblk, dma, path_out,
                           len_out, lens);
    else
        hipLaunchKernelGGL((viterbi_backtrace_kernel<NB, false>), dim3(B), dim3(256), 0, s, tb, best, T, nkmer, tblk, dma, path_out,
                           len_out, lens);
    return slk_launch_status();
}

int slk_backtrace_packed4(const uint8_t *tb, const int32_t *best, int T, int B, int nkmer, int32_t *path_out, int32_t *len_out,
                          const int *lens, hipStream_t s)
{
    const int rowbytes = nkmer / 2;
    int tblk = VBT_BLOCK / rowbytes;
    if (tblk < 1) return SLK_ERR_UNSUPPORTED;
    const int dma = (VBT_BLOCK % rowbytes == 0) && (rowbytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(tb) & 15) == 0);
    if (tblk > T) tblk = T;
    hipLaunchKernelGGL((viterbi_backtrace_kernel<4, true>), dim3(B), dim3(256), 0, s, tb, best, T, nkmer, tblk, dma, path_out,
                       len_out, lens);
    return slk_launch_status();
}

// the one-byte-per-four-states traceback of softmax_viterbi.hip: 4^5 k-mers (a row is one register per lane), a 4-byte aligned buffer
int slk_backtrace_packed8(const uint8_t *tb, const int32_t *best, int T, int B, int nkmer, int32_t *path_out, int32_t *len_out,
                          const int *lens, hipStream_t s)
{
    if (nkmer != 1024 || (reinterpret_cast<uintptr_t>(tb) & 3) != 0) return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(viterbi_backtrace_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, s, tb, best, T, B, nkmer, path_out, len_out, lens);
    return slk_launch_status();
}

static int viterbi_entry(const float *post, const float *stats, long ld, int T, int B, int nbase, int klen, float skip_pen,
                         int input_mode, float min_prob, void *workspace, size_t workspace_bytes, float *score_out,
                         int32_t *path_out, int32_t *len_out, const int32_t *lens, slk_stream_t stream)
{
    int nkmer;
    if (!post || !score_out || !path_out || !len_out || T < 1 || B < 1 || input_mode < 0 || input_mode > 2)
        return SLK_ERR_INVALID_ARG;
    if (!vit_dims(nbase, klen, &nkmer)) return SLK_ERR_INVALID_ARG;      // decode.py:50
    if (ld == 0) ld = nkmer + 1;
    if (ld < nkmer + 1) return SLK_ERR_INVALID_ARG;
    size_t need = slk_viterbi_kmer_workspace_bytes(T, B, nbase, klen);
    if (!workspace || workspace_bytes < need) return SLK_ERR_WORKSPACE;
    uint8_t *tb = static_cast<uint8_t *>(workspace);
    size_t tbbytes = ((size_t)B * T * nkmer + 255) & ~(size_t)255;
    int32_t *best = reinterpret_cast<int32_t *>(tb + tbbytes);
    hipStream_t s = slk_stream(stream);
    switch (nbase) {
    case 4: return launch_viterbi<4>(post, stats, ld, T, B, klen, nkmer, skip_pen, input_mode, min_prob, tb, best, score_out, path_out, len_out, lens, s);
    case 5: return launch_viterbi<5>(post, stats, ld, T, B, klen, nkmer, skip_pen, input_mode, min_prob, tb, best, score_out, path_out, len_out, lens, s);
    default: return SLK_ERR_UNSUPPORTED;
    }
}

extern "C" int slk_viterbi_kmer_f32(const float *post, int T, int B, int nbase, int klen, float skip_pen,
                                    int input_mode, float min_prob, void *workspace, size_t workspace_bytes,
                                    float *score_out, int32_t *path_out, int32_t *len_out, slk_stream_t stream)
{
    return viterbi_entry(post, nullptr, 0, T, B, nbase, klen, skip_pen, input_mode, min_prob, workspace, workspace_bytes,
                         score_out, path_out, len_out, nullptr, stream);
}

extern "C" int slk_viterbi_kmer_logits_f32(const float *logits, long ld, const float *stats, int T, int B, int nbase, int klen,
                                           float skip_pen, float min_prob, void *workspace, size_t workspace_bytes,
                                           float *score_out, int32_t *path_out, int32_t *len_out, slk_stream_t stream)
{
    if (!stats) return SLK_ERR_INVALID_ARG;
    return viterbi_entry(logits, stats, ld, T, B, nbase, klen, skip_pen, SLK_POST_RAW, min_prob, workspace, workspace_bytes,
                         score_out, path_out, len_out, nullptr, stream);
}

// Ragged batch (whole reads of different lengths padded to T steps): lens[b] in [1, T]; chunk b is decoded over its own
// first lens[b] steps, exactly as a call with T = lens[b] on that chunk alone would; path_out rows stay T long (-1 padded).
extern "C" int slk_viterbi_kmer_logits_ragged_f32(const float *logits, long ld, const float *stats, int T, int B, int nbase,
                                                  int klen, float skip_pen, float min_prob, const int32_t *lens,
                                                  void *workspace, size_t workspace_bytes, float *score_out,
                                                  int32_t *path_out, int32_t *len_out, slk_stream_t stream)
{
    if (!stats || !lens) return SLK_ERR_INVALID_ARG;
    return viterbi_entry(logits, stats, ld, T, B, nbase, klen, skip_pen, SLK_POST_RAW, min_prob, workspace, workspace_bytes,
                         score_out, path_out, len_out, lens, stream);
}

// ------------------------------------------------------------------------------------------------------
// decode.argmax (decode.py:5-18), batched: one workgroup per chunk, wave-parallel row argmax.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) argmax_decode_kernel(const float *__restrict__ post, int T, int B, int nst,
                                                            int zero_is_blank, int32_t *__restrict__ path_out,
                                                            int32_t *__restrict__ len_out)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    int32_t *path = path_out + (size_t)b * T;
    const int blank = zero_is_blank ? 0 : nst - 1;
    // pass 1: per-time argmax (first max) written to path[t]
    for (int t = wave; t < T; t += nw) {
        const float *p = post + ((size_t)t * B + b) * nst;
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int s = lane; s < nst; s += 64) {
            float c = p[s];
            if (c > bv) { bv = c; bi = s; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            float ov = __shfl_xor(bv, o);
            int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) path[t] = bi;
    }
    __threadfence_block();
    __syncthreads();
    // pass 2: compaction by one lane (T is small)
    if (tid == 0) {
        int n = 0;
        for (int t = 0; t < T; t++) {
            int s = path[t];
            if (s != blank) path[n++] = zero_is_blank ? s - 1 : s;
        }
        for (int t = n; t < T; t++) path[t] = -1;
        len_out[b] = n;
    }
}

extern "C" int slk_argmax_decode_f32(const float *post, int T, int B, int nstate, int zero_is_blank, int32_t *path_out,
                                     int32_t *len_out, slk_stream_t stream)
{
    if (!post || !path_out || !len_out || T < 1 || B < 1 || nstate < 1) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(argmax_decode_kernel, dim3(B), dim3(256), 0, slk_stream(stream), post, T, B, nstate,
                       zero_is_blank, path_out, len_out);
    return slk_launch_status();
}
