// softmax_viterbi.hip -- the Softmax layer's projection, softmax, prepare_post, log and the k-mer Viterbi forward pass in ONE
// kernel: the [T', B, 1025] logits (3.4 GB at B = 1024) never exist in HBM.
//
//   layers.Softmax.run        sloika/layers.py:309-314    x.W^T + b, exp(t - max) / sum
//   decode.prepare_post       sloika/decode.py:21-36      min_prob + (1 - min_prob) * post
//   decode.viterbi            sloika/decode.py:39-82      log(post + 1e-10), forward max-plus DP, traceback
//
// (the backtrace, decode.py:84-91, is decode.hip's viterbi_backtrace_kernel on the packed traceback this kernel writes.)
//
// Plan (nbase 4, klen 5: 1024 k-mers + blank; K = insize a multiple of 16, <= 128):
//   * one 512-thread workgroup = TWO chunks, walked through time in blocks of 16 steps.  The projection of a block is ONE
//     32-row MFMA tile row (v_mfma_f32_32x32x16_f16, A = x rows, B = weight columns, three fp16 terms per product as in
//     gemm_rows_f16x3.hip): rows 0-3, 8-11, 16-19, 24-27 are the 16 steps of the first chunk, the others those of the
//     second, so the accumulators of lane half h (lanes 32h .. 32h+31) ARE the logits of chunk h -- register i of a tile is
//     step i.  Wave w computes the four tiles whose column c is k-mer 4*(32w + c) + n, n = 0..3: lane (w, c, h) holds, for
//     chunk h, the 16 steps x 4 to-states 4j .. 4j+3 (j = 32w + c) that thread j of viterbi_forward4_kernel owns.  The logits
//     go from the matrix pipe to the dynamic programme without leaving the registers of the lane that consumes them.
//   * the block AFTER the one being decoded is produced meanwhile (same waves, same basic blocks: MFMAs, exponentials and
//     logarithms fill the issue slots the dependent max-plus chain leaves empty): step 0 splits the x rows into fp16 hi/lo
//     operand images in LDS (row scales by powers of two, f16split.h), steps 1-9 run the 12 * K/16 MFMAs per wave with the
//     weight fragments streamed from L2 three pairs ahead (pre-packed in fragment order: one 1-KiB coalesced load per
//     fragment), steps 10-13 reduce the row maximum and the row sum over the 32 lanes of a half (a halving butterfly on
//     v_permlane16_swap + DPP) and over the eight waves (through LDS, on the barriers the DP has anyway), steps 14-15 turn
//     the exponentials into log-posteriors.
//   * the blank column (state 0) is a float32 dot product on the vector unit (one row per 16-lane DPP row).
//   * the DP itself is viterbi_forward4_kernel's (decode.hip): ping-pong score vectors in LDS (bank-conflict-free padding),
//     one barrier per step, quad-DPP skip arg-max, first-maximum tie rules of np.argmax, packed 16-bit traceback staged in
//     LDS and written as 8-KB runs.  Given its log-posteriors the paths and float32 scores are bit-identical to the
//     reference's; the log-posteriors themselves can be dumped (lp_dump) so that tests decode THEM with the oracle.
#include "f16split.h"
#include <type_traits>

#include "decode_internal.h"

#define SV_THREADS 512
#define SV_BLK 16
#define SV_NK 1024
#define SV_AS 320               /* floats between the four first-base blocks of a score vector (5 x 64 dwords: ds_read2st64) */
#define SV_VP (4 * SV_AS)       /* padded score vector: element a*256 + r lives at a*SV_AS + r + 8 * (r >> 6) */
#ifndef SV_D
#define SV_D 3                  /* weight fragment pairs in flight per wave */
#endif
// diagnostic build (tools only): workgroup 0, wave 0 stamps the shader clock after every step's barrier into lp_dump (then a
// buffer of uint64 [periods][16], not a log-posterior dump)
#ifdef SV_DIAG
#define SV_STAMP(K)                                                                                      \
    do {                                                                                                 \
        if (blockIdx.x == 0 && tid == 0 && lp_dump)                                                      \
            reinterpret_cast<unsigned long long *>(lp_dump)[(cb + 1) * 16 + K] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// extra stamps inside steps: slot 0..15 of a second table behind the first (offset 64 periods... see tools/sv_variants.py)
#define SV_STAMP2(S)                                                                                     \
    do {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        if (blockIdx.x == 0 && tid == 0 && lp_dump)                                                      \
            reinterpret_cast<unsigned long long *>(lp_dump)[4096 + (cb + 1) * 16 + S] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    } while (0)
#else
#define SV_STAMP(K) do { } while (0)
#define SV_STAMP2(S) do { } while (0)
#endif
#ifndef SV_ORDER
#define SV_ORDER 0
#endif
#ifndef SV_MIX
#define SV_MIX 6                /* vector instructions the scheduler is asked to place behind every MFMA (SV_ORDER 2) */
#endif
#ifndef SV_MMA_STEPS
#define SV_MMA_STEPS 10         /* steps 1 .. SV_MMA_STEPS of a period carry the MFMAs of the next block */
#endif
#define SV_ETA 1e-10f
#define SV_LOG2E 1.4426950408889634f
#define SV_LN2 0.6931471805599453f
// timing-only builds of tools/build_sv_variants.sh (results are then garbage): 1 no MFMAs, 2 no weight loads, 4 no dynamic
// programme, 8 no exponentials / logarithms, 16 no row reductions, 32 no operand preparation
#ifndef SV_ABL
#define SV_ABL 0
#endif

template <int K> using ic = std::integral_constant<int, K>;
// two float32 values per vector instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32): a wave64 vector instruction takes
// four cycles of its SIMD whatever it does, so the element-wise passes over the logits are written on pairs
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- the packed weights: [wave 8][tile 4][K block KS][hi 1 KiB | lo 1 KiB] fragments, then per k-mer column the inverse
// ---- scale and the bias (both times log2 e), then the blank column's float32 weights and bias
__host__ __device__ static inline size_t sv_frag_bytes(int KS) { return (size_t)8 * 4 * KS * 2048; }
__host__ __device__ static inline size_t sv_pack_bytes(int KS) { return sv_frag_bytes(KS) + 2 * 4096 + (size_t)64 * KS + 16; }

// one wave per fragment (w, n, s); the last block writes the blank column
__global__ void __launch_bounds__(64) sv_pack_kernel(const float *__restrict__ W, const float *__restrict__ bias, int K, int KS,
                                                     uint8_t *__restrict__ pack)
{
    const int lane = threadIdx.x, c = lane & 31, hk = lane >> 5;
    const size_t frag = sv_frag_bytes(KS);
    float *cinv = reinterpret_cast<float *>(pack + frag), *cbias = cinv + 1024, *w0 = cbias + 1024;
    if ((int)blockIdx.x == 32 * KS) {
        for (int k = lane; k < 16 * KS; k += 64) w0[k] = k < K ? W[k] : 0.0f;
        if (lane == 0) w0[16 * KS] = bias ? bias[0] : 0.0f;
        return;
    }
    const int s = blockIdx.x % KS, n = (blockIdx.x / KS) & 3, w = blockIdx.x / (4 * KS);
    const int kmer = 4 * (32 * w + c) + n;
    const float *row = W + (size_t)(1 + kmer) * K;
    float amax = 0.0f;
    for (int k = 0; k < K; k++) amax = fmaxf(amax, fabsf(row[k]));
    float inv;
    const float sc = pow2_scale(amax, inv);
    half8 hi, lo;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        const int k = 16 * s + 8 * hk + jj;
        float v = k < K ? row[k] * sc : 0.0f;
        keepf(v);
        const _Float16 hv = (_Float16)v;
        hi[jj] = hv;
        lo[jj] = (_Float16)(v - (float)hv);
    }
    uint8_t *dst = pack + ((size_t)(w * 4 + n) * KS + s) * 2048 + lane * 16;
    *reinterpret_cast<half8 *>(dst) = hi;
    *reinterpret_cast<half8 *>(dst + 1024) = lo;
    if (s == 0 && hk == 0) {
        cinv[kmer] = inv * SV_LOG2E;                 // the kernel works on logits in units of log 2 (exp2 needs no multiply)
        cbias[kmer] = bias ? bias[1 + kmer] * SV_LOG2E : 0.0f;
    }
}

// ---- reductions over the 32 lanes of a wave half, 16 values at a time: every stage halves the number of values a lane
// ---- carries (it keeps the half its selector bit names and receives the partner's contribution to it), so 16 values cost
// ---- 8 + 4 + 2 + 1 + 1 exchanges instead of 16 x 5.  Returns the reduced value of index (lane & 31) >> 1.
template <bool SUM> __device__ __forceinline__ float sv_op(float a, float b) { return SUM ? a + b : fmaxf(a, b); }
template <int CTRL> __device__ __forceinline__ float sv_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <bool SUM> __device__ __forceinline__ float sv_half_reduce16(const float (&v)[16], int lane)
{
    float a[8], b[4], c[2];
    // lanes l and l ^ 16: v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its
    // second, so (value i, value i + 8) -> even rows hold both halves of value i, odd rows both halves of value i + 8
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
        a[i] = sv_op<SUM>(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    const bool s8 = lane & 8, s4 = lane & 4, s2 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {                       // partner l ^ 8 (row_ror:8)
        const float send = s8 ? a[i] : a[i + 4], keepv = s8 ? a[i + 4] : a[i];
        b[i] = sv_op<SUM>(keepv, sv_dpp<0x128>(send));
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {                       // partner 7 - l within eight lanes (row_half_mirror): bit 2 differs
        const float send = s4 ? b[i] : b[i + 2], keepv = s4 ? b[i + 2] : b[i];
        c[i] = sv_op<SUM>(keepv, sv_dpp<0x141>(send));
    }
    const float send = s2 ? c[0] : c[1], keepv = s2 ? c[1] : c[0];
    const float d = sv_op<SUM>(keepv, sv_dpp<0x4E>(send));          // partner l ^ 2 (quad_perm [2,3,0,1])
    return sv_op<SUM>(d, sv_dpp<0xB1>(d));                          // partner l ^ 1: both lanes end with the same value
}
// all 16 lanes of a DPP row receive the row's maximum / sum
template <bool SUM> __device__ __forceinline__ float sv_row_allreduce(float v)
{
    v = sv_op<SUM>(v, sv_dpp<0x128>(v));
    v = sv_op<SUM>(v, sv_dpp<0x124>(v));
    v = sv_op<SUM>(v, sv_dpp<0x122>(v));
    return sv_op<SUM>(v, sv_dpp<0x121>(v));
}

// (Their results must not be read by compiler-generated DPP instructions: see prepare_a.)
// v_max_f32 on values that come straight from memory: fmaxf() would first canonicalise every operand (v_max_f32 x, x, x -- the
// IEEE rule for signalling NaNs), doubling the instruction count of the score chain.  The DPP forms take the partner lane's
// value as first operand; hipcc pads nothing inside asm, so the two wait states between a vector write and a DPP read are here.
__device__ __forceinline__ float sv_max(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float sv_max3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float sv_quad_max(float a)           // maximum over the four lanes of a quad
{
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(r)
        : "v"(a));
    return r;
}

__device__ __forceinline__ float sv_log(float x) { return __builtin_amdgcn_logf(x) * SV_LN2; }
// decode.py:36 and :56 on a posterior p, float32 like numpy (no contraction)
__device__ __forceinline__ float sv_logpost(float p, float min_prob, float one_m)
{
    return sv_log(__fadd_rn(__fadd_rn(min_prob, __fmul_rn(one_m, p)), SV_ETA));
}

// Schedule of a period (the BS steps of the block being decoded, during which the next block is produced); k = step:
//   k = 0 .. MMA_LAST   the MFMAs (weight fragments three pairs ahead)         k = 1   operand images of the block after next
//   k = FIN_K           accumulators -> logits, the wave's maxima              k = EXP_K  exponentials, the wave's sums, x request
//   k = ROW_K           row statistics (needs every wave's sums: one barrier after EXP_K)
//   k = LOG_K ..        exponentials -> log-posteriors (needs the factors: one barrier after ROW_K)
template <int BS> struct SvSched;
template <> struct SvSched<16> { static constexpr int MMA_LAST = 10, FIN_K = 11, EXP_K = 12, ROW_K = 13, LOG_K = 14; };
template <> struct SvSched<8> { static constexpr int MMA_LAST = 4, FIN_K = 5, EXP_K = 5, ROW_K = 6, LOG_K = 7; };

// NCH chunks per workgroup (2 or 4): lane half h decodes chunks h and, with four, h + 2 -- two independent score chains per
// lane, twice the work between two barriers.  A block is BS = 32 / NCH steps of every chunk (the 32 rows of an MFMA tile).
template <int KS, int NCH, bool DUMP>
__global__ void __launch_bounds__(SV_THREADS) softmax_viterbi_kernel(const float *__restrict__ x, long ldx, int T, int B,
                                                                     const uint8_t *__restrict__ pack, float skip_pen,
                                                                     float min_prob, float one_m, uint8_t *__restrict__ tb,
                                                                     int32_t *__restrict__ best_out,
                                                                     float *__restrict__ score_out,
                                                                     const int *__restrict__ lens,
                                                                     float *__restrict__ lp_dump)
{
    constexpr int BS = 32 / NCH;                                // steps per block
    constexpr int NPL = NCH / 2;                                // chunks per lane
    using Sched = SvSched<BS>;
    constexpr int NP = 4 * KS;                                  // weight fragment pairs per wave and block
    constexpr int OFF_V = 0;                                    // [chunk NCH][parity 2][SV_VP] float
    constexpr int OFF_TBS = OFF_V + NCH * 2 * SV_VP * 4;        // [parity 2][chunk NCH][step BS][256] uint16
    constexpr int OFF_A = OFF_TBS + 2 * 16 * 1024;              // [parity 2][KS][hi, lo][64 lanes][16 B]
    constexpr int OFF_CINV = OFF_A + 2 * KS * 2048;             // [1024] float
    constexpr int OFF_CBIAS = OFF_CINV + 4096;                  // [1024] float
    constexpr int OFF_W0 = OFF_CBIAS + 4096;                    // [16 KS] float, then the blank bias
    constexpr int OFF_XINV = OFF_W0 + 64 * KS + 16;             // [parity 2][half 2][16] float: inverse row scales
    constexpr int OFF_L0 = OFF_XINV + 256;                      // [parity 2][2][16] blank logits
    constexpr int OFF_REDA = OFF_L0 + 256;                      // [half 2][wave 8][16] maxima over a wave's 128 columns
    constexpr int OFF_REDB = OFF_REDA + 1024;                   // [2][8][16] sums of exp(logit - wave maximum)
    constexpr int OFF_FAC = OFF_REDB + 1024;                    // [2][8][16] exp(wave maximum - row maximum) / row sum
    constexpr int OFF_LP0 = OFF_FAC + 1024;                     // [parity 2][2][16] blank log-posteriors
    constexpr int OFF_REDV = OFF_LP0 + 256;                     // [chunk NCH][8]
    constexpr int OFF_REDI = OFF_REDV + 128;                    // [chunk NCH][8]
    constexpr int SMEM = OFF_REDI + 128;
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
    // the 16 rows of a lane half: row i is step i % BS of chunk h + 2 * (i / BS)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hch = lane >> 5;
    const int j = 32 * wave + c, q = j >> 2, cc = j & 3;
    const int b0 = NCH * blockIdx.x;
    const int Tpad = T;
    int Tcs[NCH];
    int tmax = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        Tcs[ch] = b0 + ch < B ? (lens ? min(max(lens[b0 + ch], 1), T) : T) : 0;
        tmax = max(tmax, Tcs[ch]);
    }
    const int nblk = (tmax + BS - 1) / BS;
    // decode.py:36 and :56: log(min_prob + (1 - min_prob) p + 1e-10) as log(fma(e, factor (1 - min_prob), min_prob + 1e-10))
    const float mp_eta = __fadd_rn(min_prob, SV_ETA);

    uint16_t *const tbs = reinterpret_cast<uint16_t *>(smem + OFF_TBS);
    const float *const lp0b = reinterpret_cast<const float *>(smem + OFF_LP0) + hch * 16;
    const int o_step = j + 8 * (j >> 6), o_skip = cc * 72 + q;
    const int o_own = (j >> 6) * SV_AS + 4 * (j & 63) + 8 * ((j & 63) >> 4);     // states 4j .. 4j+3: block a = j >> 6
    auto vbase = [&](int p) __attribute__((always_inline)) { return reinterpret_cast<float *>(smem + OFF_V) + (hch + 2 * p) * 2 * SV_VP; };
    auto bar = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- constants of the whole kernel into LDS ----
    {
        const float *src = reinterpret_cast<const float *>(pack + sv_frag_bytes(KS));
        float *dst = reinterpret_cast<float *>(smem + OFF_CINV);
        for (int i = tid; i < 2048 + 16 * KS + 4; i += SV_THREADS) dst[i] = src[i];
    }

    // ---- production of a block: state ----
    // buffer loads: descriptor of the fragment area (uniform), the wave's and the fragment's offset in the scalar offset, the
    // lane's 16 bytes in the vector offset -- no per-fragment 64-bit pointers in vector registers
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(pack), 0, (int)sv_frag_bytes(KS), 0x00020000);
    const int wwave = wave * 4 * KS * 2048;
    const int wlane = lane * 16;
    half8 wfh[SV_D], wfl[SV_D];
    f32x16 acc[4];
    float val[4][16];                                          // logits -> exponentials of the block in the making
    // log-posteriors of the block being decoded, [row][to-state]: the four to-states of a step lie in neighbouring registers, so the
    // packed adds of the dynamic programme take them as they are (laid out [to-state][row] like the accumulators they needed two
    // register moves per pair and step)
    float lp[16][4];
    float4 xr0, xr1;                                           // this lane's eight x values of a coming block
    // operand preparation: wave w, 16-lane row r4 handles MFMA row rho = 4 w + r4 = lane half (w & 1), row i of that half
    const int kb = lane & 15, r4 = lane >> 4;
    const int pa_half = wave & 1, pa_i = r4 + 4 * (wave >> 1), rho = 4 * wave + r4;
    const int pa_chunk = pa_half + 2 * (pa_i / BS), pa_step = pa_i % BS;
    const bool pa_act = kb < 2 * KS;
    const int pa_b = min(b0 + pa_chunk, B - 1);

    auto load_x = [&](int blk) __attribute__((always_inline)) {
        const int t = min(BS * blk + pa_step, T - 1);
        const float *src = x + ((size_t)t * B + pa_b) * ldx + 8 * min(kb, 2 * KS - 1);
        xr0 = *reinterpret_cast<const float4 *>(src);
        xr1 = *reinterpret_cast<const float4 *>(src + 4);
    };
    // x rows of block blk -> fp16 hi/lo A-operand images, row scales, blank logits (all double buffered by block parity)
    auto prepare_a = [&](int blk) __attribute__((always_inline)) {
        const int par = blk & 1;
        float xv[8] = {xr0.x, xr0.y, xr0.z, xr0.w, xr1.x, xr1.y, xr1.z, xr1.w};
#pragma unroll
        for (int i = 0; i < 8; i++) xv[i] = pa_act ? xv[i] : 0.0f;
        float amax = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; i++) amax = fmaxf(amax, fabsf(xv[i]));
        // (no sv_max here: a DPP instruction that reads a register needs two wait states after the vector instruction that
        // wrote it, and the compiler pads only behind instructions it emitted itself -- never feed its DPP from inline asm)
        amax = sv_row_allreduce<false>(amax);
        float inv;
        const float sc = pow2_scale(amax, inv);
        const float *w0 = reinterpret_cast<const float *>(smem + OFF_W0);
        const float4 wa = *reinterpret_cast<const float4 *>(w0 + 8 * min(kb, 2 * KS - 1));
        const float4 wb = *reinterpret_cast<const float4 *>(w0 + 8 * min(kb, 2 * KS - 1) + 4);
        float dot = xv[0] * wa.x;
        dot = fmaf(xv[1], wa.y, dot);
        dot = fmaf(xv[2], wa.z, dot);
        dot = fmaf(xv[3], wa.w, dot);
        dot = fmaf(xv[4], wb.x, dot);
        dot = fmaf(xv[5], wb.y, dot);
        dot = fmaf(xv[6], wb.z, dot);
        dot = fmaf(xv[7], wb.w, dot);
        dot = sv_row_allreduce<true>(dot);
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float v = xv[i] * sc;
            keepf(v);
            const _Float16 hv = (_Float16)v;
            hi[i] = hv;
            lo[i] = (_Float16)(v - (float)hv);
        }
        if (pa_act) {
            uint8_t *dst = smem + OFF_A + par * KS * 2048 + (kb >> 1) * 2048 + (rho + 32 * (kb & 1)) * 16;
            *reinterpret_cast<half8 *>(dst) = hi;
            *reinterpret_cast<half8 *>(dst + 1024) = lo;
        }
        if (kb == 0) {
            reinterpret_cast<float *>(smem + OFF_XINV)[par * 32 + pa_half * 16 + pa_i] = inv;
            reinterpret_cast<float *>(smem + OFF_L0)[par * 32 + pa_half * 16 + pa_i] = (dot + w0[16 * KS]) * SV_LOG2E;
        }
    };
    auto wload = [&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NP && !(SV_ABL & 2)) {
            wfh[p % SV_D] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wwave + p * 2048, 0));
            wfl[p % SV_D] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wwave + p * 2048 + 1024, 0));
        }
    };
    // fragment pair p = (tile n, K block s): three MFMAs, small terms first (gemm_rows_f16x3.hip).
    //
    // Register hygiene around v_mfma_f32_32x32x16_f16 (hipcc of ROCm 7.2, gfx950).  The first MFMA of a tile has C = 0 and a
    // destination that is defined right there, so the allocator may place it on registers that have just died: the
    // instruction's own `alo` operand, or the operands of the MFMAs issued just before it (v_mfma v[34:49], v[34:37], ..., 0
    // two lines behind an MFMA that reads v[34:37]).  The compiler considers both safe; the hardware does not execute them
    // safely: whenever the matrix pipe was contended, row 30 of a tile came out computed from overwritten operand registers --
    // one register of sixteen lanes, in some waves, in some runs (found by repeating tests/test_gpu_fused_decode.py's K = 64
    // cases and by tools/mfma_overlap_scan.py on the ISA).  So the operands of a pair (and of the pair before it) are kept
    // alive until the NEXT pair's first MFMA has been issued: an empty asm takes that MFMA's result together with them.
    half8 pahi = {}, palo = {}, pwfh = {}, pwfl = {};           // the previous pair's operands
    auto mma_pair = [&](auto pc, int apar) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        constexpr int n = p / KS, s = p % KS;
        const uint8_t *ab = smem + OFF_A + apar * KS * 2048 + s * 2048 + lane * 16;
        const half8 ahi = *reinterpret_cast<const half8 *>(ab);
        const half8 alo = *reinterpret_cast<const half8 *>(ab + 1024);
        const half8 wh = wfh[p % SV_D], wl = wfl[p % SV_D];
        if constexpr (s == 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[n][i] = 0.0f;
        }
        if constexpr (!(SV_ABL & 1)) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, wh, acc[n], 0, 0, 0);
            asm volatile("" : "+v"(acc[n]) : "v"(alo), "v"(pahi), "v"(palo), "v"(pwfh), "v"(pwfl));
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, wl, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, wh, acc[n], 0, 0, 0);
        } else {
            acc[n][0] += (float)ahi[0] + (float)alo[1] + (float)wh[0] + (float)wl[1];
        }
        pahi = ahi;
        palo = alo;
        pwfh = wh;
        pwfl = wl;
        wload(ic<p + SV_D>{});
    };
    auto wload_first = [&](auto pc, auto &&self) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < SV_D) {
            wload(pc);
            self(ic<p + 1>{}, self);
        }
    };
    auto mma_range = [&](auto lo_c, auto hi_c, int apar, auto &&self) __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        if constexpr (lo < hi) {
            mma_pair(ic<lo>{}, apar);
            self(ic<lo + 1>{}, hi_c, apar, self);
        }
    };
    auto load16p = [&](const float *p, float (&out)[16]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float4 v = reinterpret_cast<const float4 *>(p)[i];
            out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
        }
    };
    // Softmax over the 1025 columns of a row in the online form: a wave normalises its 128 columns by ITS maximum (known to all
    // its lanes after one butterfly + a round trip through the wave's own LDS row, no barrier), the eight partial
    // (maximum, sum) pairs of a row meet once, and every wave gets back the factor exp(m_wave - m_row) / sum_row that turns its
    // exponentials into posteriors -- one exchange between waves instead of one for the maximum and one for the sum.
    // Logits are kept in units of log 2 (column scales and biases arrive multiplied by log2 e): exp is a bare v_exp_f32.
    float *const my_max = reinterpret_cast<float *>(smem + OFF_REDA) + (hch * 8 + wave) * 16;
    float *const my_sum = reinterpret_cast<float *>(smem + OFF_REDB) + (hch * 8 + wave) * 16;
    const float *const my_fac = reinterpret_cast<const float *>(smem + OFF_FAC) + (hch * 8 + wave) * 16;
    // scaled accumulators -> logits (gemm_rows_f16x3.hip's finish), the wave's maxima
    auto finish_max = [&](int nb) __attribute__((always_inline)) {
        float xinv[16];
        load16p(reinterpret_cast<const float *>(smem + OFF_XINV) + (nb & 1) * 32 + hch * 16, xinv);
        const float4 ci = reinterpret_cast<const float4 *>(smem + OFF_CINV)[j];
        const float4 cb = reinterpret_cast<const float4 *>(smem + OFF_CBIAS)[j];
        const float civ[4] = {ci.x, ci.y, ci.z, ci.w}, cbv[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 a = {acc[n][i], acc[n][i + 1]}, xi = {xinv[i], xinv[i + 1]};
                const f32x2 v = __builtin_elementwise_fma(a * xi, f32x2{civ[n], civ[n]}, f32x2{cbv[n], cbv[n]});
                val[n][i] = v.x;
                val[n][i + 1] = v.y;
            }
        float m[16];
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = fmaxf(fmaxf(val[0][i], val[1][i]), fmaxf(val[2][i], val[3][i]));
        const float r = (SV_ABL & 16) ? m[0] + m[15] : sv_half_reduce16<false>(m, lane);
        if (!(c & 1)) my_max[c >> 1] = r;
    };
    auto exp_sum = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");                         // the wave's own LDS writes above, read back in order
        float m[16];
        load16p(my_max, m);
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 d = f32x2{val[n][i], val[n][i + 1]} - f32x2{m[i], m[i + 1]};
                val[n][i] = (SV_ABL & 8) ? d.x : __builtin_amdgcn_exp2f(d.x);
                val[n][i + 1] = (SV_ABL & 8) ? d.y : __builtin_amdgcn_exp2f(d.y);
            }
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 t = ((f32x2{val[0][i], val[0][i + 1]} + f32x2{val[1][i], val[1][i + 1]}) + f32x2{val[2][i], val[2][i + 1]}) +
                            f32x2{val[3][i], val[3][i + 1]};
            s[i] = t.x;
            s[i + 1] = t.y;
        }
        const float r = (SV_ABL & 16) ? s[0] + s[15] : sv_half_reduce16<true>(s, lane);
        if (!(c & 1)) my_sum[c >> 1] = r;
    };
    auto row_stats = [&](int nb) __attribute__((always_inline)) {                             // 256 lanes: (half, row) x the eight waves' shares
        if (tid < 256) {
            const int hh = tid >> 7, ri = (tid >> 3) & 15, w = tid & 7, row = hh * 16 + ri;
            const int ch = hh + 2 * (ri / BS), st = ri % BS;
            const float l0 = reinterpret_cast<const float *>(smem + OFF_L0)[(nb & 1) * 32 + row];
            const float mw = reinterpret_cast<const float *>(smem + OFF_REDA)[(hh * 8 + w) * 16 + ri];
            const float sw = reinterpret_cast<const float *>(smem + OFF_REDB)[(hh * 8 + w) * 16 + ri];
            float m = fmaxf(mw, sv_dpp<0x141>(mw));                 // 7 - l within eight lanes
            m = fmaxf(m, sv_dpp<0x4E>(m));                          // l ^ 2
            m = fmaxf(m, sv_dpp<0xB1>(m));                          // l ^ 1
            m = fmaxf(m, l0);
            const float ew = __builtin_amdgcn_exp2f(mw - m);
            const float e0 = __builtin_amdgcn_exp2f(l0 - m);
            float ssum = sw * ew;
            ssum += sv_dpp<0x141>(ssum);
            ssum += sv_dpp<0x4E>(ssum);
            ssum += sv_dpp<0xB1>(ssum);
            ssum += e0;
            // Ragged batch: a chunk past its own end keeps its scores.  Its steps get NaN log-posteriors and a zero blank one, so
            // that "move" (a > comparison) is false and "stay" adds nothing -- no per-state guard in the dynamic programme.
            int tc = Tcs[0];
#pragma unroll
            for (int k2 = 1; k2 < NCH; k2++) tc = ch == k2 ? Tcs[k2] : tc;
            const bool dead = BS * nb + st >= tc;
            const float inv = dead ? __builtin_nanf("") : 1.0f / ssum;
            reinterpret_cast<float *>(smem + OFF_FAC)[(hh * 8 + w) * 16 + ri] = ew * (inv * one_m);
            if (w == 0) {
                const float lb = dead ? 0.0f : sv_log(fmaf(e0, inv * one_m, mp_eta));
                reinterpret_cast<float *>(smem + OFF_LP0)[(nb & 1) * 32 + row] = lb;
#ifndef SV_DIAG
                if constexpr (DUMP) {
                    const int t = BS * nb + st, bb = b0 + ch;
                    if (t < T && bb < B) lp_dump[((size_t)t * B + bb) * (SV_NK + 1)] = lb;
                }
#endif
            }
        }
    };
    // exponentials -> log-posteriors, written over the log-posteriors the dynamic programme has already consumed: row i of the
    // block being decoded is dead once step i % BS has run.  Rows with i0 <= i % BS < i1 of tiles n0 .. n1-1.
    auto to_logpost = [&](auto n0c, auto n1c, auto i0c, auto i1c) __attribute__((always_inline)) {
        constexpr int n0 = decltype(n0c)::value, n1 = decltype(n1c)::value, i0 = decltype(i0c)::value, i1 = decltype(i1c)::value;
        if constexpr (n0 < n1 && i0 < i1) {
            float fac[16];
            load16p(my_fac, fac);
            // tiles in pairs (n, n + 1): the fused multiply-adds run on row pairs (neighbouring accumulator registers), the logarithms
            // are scalar instructions that write where the dynamic programme wants them, the final scaling runs on to-state pairs
            static_assert((n1 - n0) % 2 == 0, "tiles in pairs");
#pragma unroll
            for (int n = n0; n < n1; n += 2)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const bool in0 = i % BS >= i0 && i % BS < i1, in1 = (i + 1) % BS >= i0 && (i + 1) % BS < i1;
                    if (in0 && in1) {
                        const f32x2 a = __builtin_elementwise_fma(f32x2{val[n][i], val[n][i + 1]}, f32x2{fac[i], fac[i + 1]},
                                                                  f32x2{mp_eta, mp_eta});
                        const f32x2 b = __builtin_elementwise_fma(f32x2{val[n + 1][i], val[n + 1][i + 1]}, f32x2{fac[i], fac[i + 1]},
                                                                  f32x2{mp_eta, mp_eta});
                        const f32x2 l0 = f32x2{__builtin_amdgcn_logf(a.x), __builtin_amdgcn_logf(b.x)} * SV_LN2;
                        const f32x2 l1 = f32x2{__builtin_amdgcn_logf(a.y), __builtin_amdgcn_logf(b.y)} * SV_LN2;
                        lp[i][n] = (SV_ABL & 8) ? a.x : l0.x;
                        lp[i][n + 1] = (SV_ABL & 8) ? b.x : l0.y;
                        lp[i + 1][n] = (SV_ABL & 8) ? a.y : l1.x;
                        lp[i + 1][n + 1] = (SV_ABL & 8) ? b.y : l1.y;
                    } else {
                        if (in0) {
                            lp[i][n] = sv_log(fmaf(val[n][i], fac[i], mp_eta));
                            lp[i][n + 1] = sv_log(fmaf(val[n + 1][i], fac[i], mp_eta));
                        }
                        if (in1) {
                            lp[i + 1][n] = sv_log(fmaf(val[n][i + 1], fac[i + 1], mp_eta));
                            lp[i + 1][n + 1] = sv_log(fmaf(val[n + 1][i + 1], fac[i + 1], mp_eta));
                        }
                    }
                }
            // the values are first used a period later: without this the compiler sinks the whole transform to the loop's end,
            // out of the steps whose waiting time it is meant to fill
#pragma unroll
            for (int n = n0; n < n1; n++)
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (i % BS >= i0 && i % BS < i1) keepf(lp[i][n]);
        }
    };
    auto dump_block = [&](int nb) __attribute__((always_inline)) {
#ifdef SV_DIAG
        return;
#endif
        if constexpr (DUMP) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int t = BS * nb + i % BS, bb = b0 + hch + 2 * (i / BS);
                if (t < T && bb < B) {
                    float *dst = lp_dump + ((size_t)t * B + bb) * (SV_NK + 1) + 1 + 4 * j;
                    dst[0] = lp[i][0]; dst[1] = lp[i][1]; dst[2] = lp[i][2]; dst[3] = lp[i][3];
                }
            }
        }
    };
    // side work of step k of a period: the production of block nb (and the operand images of block nb + 1)
    auto side = [&](auto kc, int nb) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int NM = Sched::MMA_LAST + 1;
        if constexpr (k == 0) wload_first(ic<0>{}, wload_first);
        if constexpr (k <= Sched::MMA_LAST) mma_range(ic<(k * NP) / NM>{}, ic<((k + 1) * NP) / NM>{}, nb & 1, mma_range);
        if constexpr (k == 1 && !(SV_ABL & 32)) prepare_a(nb + 1);
        if constexpr (k == Sched::FIN_K) finish_max(nb);
        if constexpr (k == Sched::EXP_K) {
            load_x(nb + 2);                                    // its operand images are made in step 1 of the next period
            exp_sum();
        }
        if constexpr (k == Sched::ROW_K) row_stats(nb);
        if constexpr (BS == 16) {
            if constexpr (k == 14) to_logpost(ic<0>{}, ic<2>{}, ic<0>{}, ic<14>{});
            if constexpr (k == 15) to_logpost(ic<2>{}, ic<4>{}, ic<0>{}, ic<15>{});
        } else {
            if constexpr (k == 7) to_logpost(ic<0>{}, ic<4>{}, ic<0>{}, ic<7>{});
        }
    };
    auto side_tail = [&](int nb) __attribute__((always_inline)) {                             // after the last step of the block being decoded
        if constexpr (BS == 16) {
            to_logpost(ic<0>{}, ic<2>{}, ic<14>{}, ic<16>{});
            to_logpost(ic<2>{}, ic<4>{}, ic<15>{}, ic<16>{});
        } else {
            to_logpost(ic<0>{}, ic<4>{}, ic<7>{}, ic<8>{});
        }
        dump_block(nb);
    };

    // ---- the dynamic programme: step t0 + k of block cb, log-posteriors lp[p BS + k][n] (viterbi_forward4_kernel::step) ----
    float own[NPL][4];                                         // this thread's four scores of the previous step, per chunk
#pragma unroll
    for (int p = 0; p < NPL; p++)
#pragma unroll
        for (int n = 0; n < 4; n++) own[p][n] = 0.0f;
    struct DpIn { float vs0, vs1, vs2, vs3, vk0, vk1, vk2, vk3, lp0; };
    auto dp_read = [&](auto kc, int par, int p) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        const float *vold = vbase(p) + ((k & 1) ^ 1) * SV_VP;
        DpIn d;
        d.vs0 = vold[o_step]; d.vs1 = vold[SV_AS + o_step]; d.vs2 = vold[2 * SV_AS + o_step]; d.vs3 = vold[3 * SV_AS + o_step];
        d.vk0 = vold[o_skip]; d.vk1 = vold[SV_AS + o_skip]; d.vk2 = vold[2 * SV_AS + o_skip]; d.vk3 = vold[3 * SV_AS + o_skip];
        d.lp0 = lp0b[par * 32 + p * BS + k];
        return d;
    };
    auto dp_compute = [&](auto kc, int t0, int par, int p, const DpIn &d) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        float *vnew = vbase(p) + (k & 1) * SV_VP;
        // Values first, arguments beside them: the score chain is max, max, two quad maxima, one subtraction, one maximum, one
        // addition, one compare, one select; which predecessor attained a maximum only feeds the traceback word.
        const float vs0 = d.vs0, vs1 = d.vs1, vs2 = d.vs2, vs3 = d.vs3, vk0 = d.vk0, vk1 = d.vk1, vk2 = d.vk2, vk3 = d.vk3;
        const float lp0 = d.lp0;
        // step maximum over a; the first a that attains it (np.argmax, decode.py:67-68)
        const float sstep = sv_max(sv_max3(vs0, vs1, vs2), vs3);
        int sarg = vs2 == sstep ? 2 : 3;
        sarg = vs1 == sstep ? 1 : sarg;
        sarg = vs0 == sstep ? 0 : sarg;
        // skip maximum over ab = a*4 + b (decode.py:72-73): this thread's share is b = cc, the quad holds the other three; the
        // first ab that attains the maximum = the smallest key among the lanes whose share attains it
        const float kpart = sv_max(sv_max3(vk0, vk1, vk2), vk3);
        int kl = vk2 == kpart ? 8 + cc : 12 + cc;
        kl = vk1 == kpart ? 4 + cc : kl;
        kl = vk0 == kpart ? cc : kl;
        const float kbest = sv_quad_max(kpart);
        int karg = kpart == kbest ? kl : 16;
        karg = min(karg, __builtin_amdgcn_update_dpp(0, karg, 0xB1, 0xf, 0xf, false));
        karg = min(karg, __builtin_amdgcn_update_dpp(0, karg, 0x4E, 0xf, 0xf, false));
        const float sskip = kbest - skip_pen;                       // decode.py:72
        const float mx = sv_max(sstep, sskip);
        const bool bystep = sstep > sskip;                          // decode.py:76 (tie -> skip)
        float nw[4];
        uint32_t moves = 0;                                         // bit 2n: to-state n moves
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const float nv = lp[p * BS + k][n] + mx;                // decode.py:75
            const float stay = own[p][n] + lp0;                     // decode.py:80
            const bool move = nv > stay;                            // decode.py:81 (tie -> stay)
            moves |= move ? (1u << (2 * n)) : 0u;
            float r = move ? nv : stay;
            if constexpr (k == 0) r = t0 == 0 ? lp[p * BS][n] : r;  // t = 0: v = lpost[0][1:] (decode.py:57)
            nw[n] = r;
        }
        // two bits per to-state: 0 stay, 1 step, 2 skip (viterbi_forward4_kernel's traceback word)
        const uint32_t packed = (moves << (bystep ? 0 : 1)) | ((uint32_t)sarg << 8) | ((uint32_t)karg << 10);
#pragma unroll
        for (int n = 0; n < 4; n++) own[p][n] = nw[n];
        *reinterpret_cast<float4 *>(&vnew[o_own]) = make_float4(nw[0], nw[1], nw[2], nw[3]);
        tbs[((par * NCH + hch + 2 * p) * BS + k) * 256 + j] = (uint16_t)packed;
    };
    // rows of block blk (staged with parity par) -> HBM: per chunk BS rows of 512 bytes, contiguous on both sides
    auto flush_tb = [&](int blk, int par) __attribute__((always_inline)) {
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            constexpr int PER_CHUNK = BS * 512;                     // bytes
            const int off = pass * 8192 + tid * 16;
            const int ch = off / PER_CHUNK, rest = off % PER_CHUNK, t = BS * blk + rest / 512;
            int tc = Tcs[0];
#pragma unroll
            for (int k2 = 1; k2 < NCH; k2++) tc = ch == k2 ? Tcs[k2] : tc;
            uint8_t *dst = tb + ((size_t)(b0 + ch) * Tpad + BS * blk) * (SV_NK / 2) + rest;
            if (t >= 1 && t < tc)
                *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(smem + OFF_TBS + par * 16384 + off);
        }
    };

    // MFMA steps: one MFMA, then a group of the programme's vector instructions (the wave issues in order: MFMAs back to back
    // would keep it from its vector work for 32 cycles each, vector work first would leave the matrix pipe idle)
    auto mix_hint = [&](auto kc) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr int NM = Sched::MMA_LAST + 1;
        if constexpr (k <= Sched::MMA_LAST) {
            constexpr int npairs = ((k + 1) * NP) / NM - (k * NP) / NM;
#pragma unroll
            for (int i = 0; i < npairs; i++) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the pair's A operands
#pragma unroll
                for (int m = 0; m < 3; m++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, SV_MIX, 0);   // vector instructions
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // the fragment loads three pairs ahead
            }
        }
    };
    auto period = [&](auto dpc, auto prodc, int cb) __attribute__((always_inline)) {
        constexpr bool DP = decltype(dpc)::value, PROD = decltype(prodc)::value;
        const int t0 = BS * cb, par = cb & 1, nb = cb + 1;
        // Order inside a step: the programme's LDS reads are requested first, the production work of the step runs while they
        // are on their way, then the programme's arithmetic and its writes.
#define SV_STEP(K)                                                                         \
    do {                                                                                   \
        if constexpr ((K) < BS) {                                                          \
            if constexpr (DP && !(SV_ABL & 4)) {                                           \
                DpIn d[NPL];                                                               \
                _Pragma("unroll") for (int p = 0; p < NPL; p++) d[p] = dp_read(ic<K>{}, par, p); \
                if constexpr (SV_ORDER >= 1) __builtin_amdgcn_sched_barrier(0);            \
                if constexpr (PROD) side(ic<K>{}, nb);                                     \
                if constexpr (SV_ORDER == 1) __builtin_amdgcn_sched_barrier(0);            \
                _Pragma("unroll") for (int p = 0; p < NPL; p++) dp_compute(ic<K>{}, t0, par, p, d[p]); \
                if constexpr (PROD && SV_ORDER == 2) mix_hint(ic<K>{});                    \
            } else {                                                                       \
                if constexpr (PROD) side(ic<K>{}, nb);                                     \
            }                                                                              \
            bar();                                                                         \
            SV_STAMP(K);                                                                   \
            if constexpr ((K) == Sched::FIN_K - 1) {                                       \
                if (DP && cb >= 1) flush_tb(cb - 1, par ^ 1);                              \
            }                                                                              \
        }                                                                                  \
    } while (0)
        SV_STEP(0);
        SV_STEP(1);
        SV_STEP(2);
        SV_STEP(3);
        SV_STEP(4);
        SV_STEP(5);
        SV_STEP(6);
        SV_STEP(7);
        SV_STEP(8);
        SV_STEP(9);
        SV_STEP(10);
        SV_STEP(11);
        SV_STEP(12);
        SV_STEP(13);
        SV_STEP(14);
        SV_STEP(15);
#undef SV_STEP
        if constexpr (PROD) side_tail(nb);
    };

    load_x(0);
    bar();                                                      // constants staged
    prepare_a(0);
#ifdef SV_DBG_A
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    load_x(1);
#ifdef SV_DBG_B
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    bar();
    period(std::false_type{}, std::true_type{}, -1);
    for (int cb = 0; cb + 1 < nblk; cb++) period(std::true_type{}, std::true_type{}, cb);
    period(std::true_type{}, std::false_type{}, nblk - 1);
    flush_tb(nblk - 1, (nblk - 1) & 1);

    // ---- first argmax of the final scores (np.argmax, decode.py:85); the last step of a block writes parity 1 ----
#pragma unroll
    for (int p = 0; p < NPL; p++) {
        const int ch = hch + 2 * p;
        const float4 fv = *reinterpret_cast<const float4 *>(&vbase(p)[SV_VP + o_own]);
        const float f[4] = {fv.x, fv.y, fv.z, fv.w};
        float bv = f[0];
        int bi = 4 * j;
#pragma unroll
        for (int n = 1; n < 4; n++)
            if (f[n] > bv) { bv = f[n]; bi = 4 * j + n; }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        float *redv = reinterpret_cast<float *>(smem + OFF_REDV) + ch * 8;
        int *redi = reinterpret_cast<int *>(smem + OFF_REDI) + ch * 8;
        if (c == 0) { redv[wave] = bv; redi[wave] = bi; }
    }
    bar();
    if (tid < NCH && b0 + tid < B) {
        const float *redv = reinterpret_cast<const float *>(smem + OFF_REDV) + tid * 8;
        const int *redi = reinterpret_cast<const int *>(smem + OFF_REDI) + tid * 8;
        float bv = redv[0];
        int bi = redi[0];
        for (int w = 1; w < 8; w++)
            if (redv[w] > bv || (redv[w] == bv && redi[w] < bi)) { bv = redv[w]; bi = redi[w]; }
        score_out[b0 + tid] = bv;
        best_out[b0 + tid] = bi;
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
// insize 64, 96, 112, 128: the Softmax layers of the shipped raw models (baseline_raw_gru, raw_0.98_rgrgr, pretrained.pkl /
// raw_1.00_rGr padded, bigger_raw_gru)
static bool sv_shape_ok(int K, int nbase, int klen) { return nbase == 4 && klen == 5 && (K == 64 || K == 96 || K == 112 || K == 128); }

extern "C" size_t slk_softmax_viterbi_pack_bytes(int K, int nbase, int klen)
{
    if (!sv_shape_ok(K, nbase, klen)) return 0;
    return (sv_pack_bytes(K / 16) + 255) & ~(size_t)255;
}

extern "C" int slk_softmax_viterbi_pack_f32(const float *W, const float *bias, int K, int nbase, int klen, void *pack,
                                            slk_stream_t stream)
{
    if (!W || !pack || K < 1 || nbase < 2 || klen < 1) return SLK_ERR_INVALID_ARG;
    if (!sv_shape_ok(K, nbase, klen)) return SLK_ERR_UNSUPPORTED;
    const int KS = K / 16;
    hipLaunchKernelGGL(sv_pack_kernel, dim3(32 * KS + 1), dim3(64), 0, slk_stream(stream), W, bias, K, KS,
                       static_cast<uint8_t *>(pack));
    return slk_launch_status();
}

template <int KS, int NCH>
static int sv_launch(const float *x, long ldx, int T, int B, const uint8_t *pack, float skip_pen, float min_prob, uint8_t *tb,
                     int32_t *best, float *score_out, const int *lens, float *lp_dump, hipStream_t s)
{
#ifdef SV_DIAG
    constexpr bool diag = true;
#else
    constexpr bool diag = false;
#endif
    if (lp_dump && !diag)
        hipLaunchKernelGGL((softmax_viterbi_kernel<KS, NCH, true>), dim3((B + NCH - 1) / NCH), dim3(SV_THREADS), 0, s, x, ldx, T, B,
                           pack, skip_pen, min_prob, (float)(1.0 - (double)min_prob), tb, best, score_out, lens, lp_dump);
    else
        hipLaunchKernelGGL((softmax_viterbi_kernel<KS, NCH, false>), dim3((B + NCH - 1) / NCH), dim3(SV_THREADS), 0, s, x, ldx, T,
                           B, pack, skip_pen, min_prob, (float)(1.0 - (double)min_prob), tb, best, score_out, lens, lp_dump);
    return slk_launch_status();
}

extern "C" int slk_softmax_viterbi_f32(const float *x, long ldx, const void *pack, int K, int T, int B, int nbase, int klen,
                                       float skip_pen, float min_prob, const int32_t *lens, int plan, void *workspace,
                                       size_t workspace_bytes, float *score_out, int32_t *path_out, int32_t *len_out,
                                       float *lp_dump, slk_stream_t stream)
{
    if (!x || !pack || !score_out || !path_out || !len_out || T < 1 || B < 1 || K < 1 || ldx < K || nbase < 2 || klen < 3 ||
        (plan != 0 && plan != 2 && plan != 4))
        return SLK_ERR_INVALID_ARG;
#ifndef SV_WITH_NCH4
    if (plan == 4) return SLK_ERR_UNSUPPORTED;    // four chunks per workgroup: measured equal to two (a vector instruction
                                                  // costs its SIMD four cycles either way), so production builds leave it out
#endif
    if (!sv_shape_ok(K, nbase, klen) || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;
    const size_t need = slk_viterbi_kmer_workspace_bytes(T, B, nbase, klen);
    if (!workspace || workspace_bytes < need) return SLK_ERR_WORKSPACE;
    uint8_t *tb = static_cast<uint8_t *>(workspace);
    const size_t tbbytes = ((size_t)B * T * SV_NK + 255) & ~(size_t)255;
    int32_t *best = reinterpret_cast<int32_t *>(tb + tbbytes);
    const uint8_t *pk = static_cast<const uint8_t *>(pack);
    hipStream_t s = slk_stream(stream);
    int rc;
#ifdef SV_WITH_NCH4
    const bool four = plan == 4;
#endif
    switch (K / 16) {
#ifdef SV_WITH_NCH4
#define SV_CASE(KS)                                                                                                              \
    case KS:                                                                                                                     \
        rc = four ? sv_launch<KS, 4>(x, ldx, T, B, pk, skip_pen, min_prob, tb, best, score_out, lens, lp_dump, s)                \
                  : sv_launch<KS, 2>(x, ldx, T, B, pk, skip_pen, min_prob, tb, best, score_out, lens, lp_dump, s);               \
        break;
#else
#define SV_CASE(KS)                                                                                                              \
    case KS: rc = sv_launch<KS, 2>(x, ldx, T, B, pk, skip_pen, min_prob, tb, best, score_out, lens, lp_dump, s); break;
#endif
#ifdef SV_ONLY_KS          /* development builds: one instantiation */
    SV_CASE(SV_ONLY_KS)
#else
    SV_CASE(4) SV_CASE(6) SV_CASE(7) SV_CASE(8)
#endif
#undef SV_CASE
    default: return SLK_ERR_UNSUPPORTED;
    }
    if (rc != SLK_OK) return rc;
#ifdef SV_NO_BACKTRACE
    return rc;
#else
    return slk_backtrace_packed4(tb, best, T, B, SV_NK, path_out, len_out, lens, s);
#endif
}
