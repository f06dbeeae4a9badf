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
//     one barrier per step, quad-DPP skip arg-max, first-maximum tie rules of np.argmax, the traceback packed into ONE BYTE per
//     four states (round 5: 256 bytes per step and chunk, half of round 4's) staged in LDS and written as 4-KB runs.  Given its log-posteriors the paths and float32 scores are bit-identical to the
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
#ifndef SV_AHEAD
#define SV_AHEAD 4              /* positions between the LDS request of a pair's A images and its first MFMA */
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
template <int B_, int E_, class F> __device__ __forceinline__ void static_for_sv(F &&f)
{
    if constexpr (B_ < E_) {
        f(ic<B_>{});
        static_for_sv<B_ + 1, E_>(f);
    }
}
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
// The same as ONE hand-scheduled sequence (round 5): hipcc's schedule of sv_half_reduce16 is ~90 instructions -- fmaxf canonicalises
// its operands (v_max x, x, x), every select pair costs two v_cndmask and a mask, every DPP move waits two states for its source --
// ~880 cycles per reduction in the per-step stamps.  Here the selects of the 8- and 4-lane stages are bank-masked DPP instructions
// (a lane's selector bit = its bank of four lanes: the instruction that serves lanes with the bit clear writes banks 0/1 or 0/2,
// the other the rest, in place), the stages are interleaved so that every DPP source is at least two instructions old: 35 issue
// slots.  Data movement and operand pairing are those of sv_half_reduce16 (the sums come out bit for bit the same).  The values
// must not be NaN-signalling (v_max_f32 / v_add_f32 as the hardware executes them).  m2 = 0xCCCC...: lanes with bit 1 set.
#define SV_RED_STAGES(OP)                                                                                        \
    "s_nop 1\n\t"                                                                                                \
    "v_permlane16_swap_b32 %0, %8\n\t"                                                                           \
    "v_permlane16_swap_b32 %1, %9\n\t"                                                                           \
    "v_permlane16_swap_b32 %2, %10\n\t"                                                                          \
    "v_permlane16_swap_b32 %3, %11\n\t"                                                                          \
    "v_permlane16_swap_b32 %4, %12\n\t"                                                                          \
    "v_permlane16_swap_b32 %5, %13\n\t"                                                                          \
    "v_permlane16_swap_b32 %6, %14\n\t"                                                                          \
    "v_permlane16_swap_b32 %7, %15\n\t"                                                                          \
    OP " %0, %0, %8\n\t"                                                                                         \
    OP " %1, %1, %9\n\t"                                                                                         \
    OP " %2, %2, %10\n\t"                                                                                        \
    OP " %3, %3, %11\n\t"                                                                                        \
    OP " %4, %4, %12\n\t"                                                                                        \
    OP " %5, %5, %13\n\t"                                                                                        \
    OP " %6, %6, %14\n\t"                                                                                        \
    OP " %7, %7, %15\n\t"                                                                                        \
    OP "_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                                                \
    OP "_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                                                \
    OP "_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                                                \
    OP "_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                                                \
    OP "_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                                                \
    OP "_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                                                \
    OP "_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"                                                \
    OP "_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"                                                \
    OP "_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"                                          \
    OP "_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"                                          \
    OP "_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"                                          \
    OP "_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"                                          \
    "v_cndmask_b32 %2, %1, %0, %16\n\t"                                                                          \
    "v_cndmask_b32 %0, %0, %1, %16\n\t"                                                                          \
    "s_nop 0\n\t"                                                                                                \
    OP "_dpp %0, %2, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                                      \
    "s_nop 1\n\t"                                                                                                \
    OP "_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
template <bool SUM> __device__ __forceinline__ float sv_half_reduce16_asm(float (&v)[16], unsigned long long m2)
{
    if constexpr (SUM)
        asm volatile(SV_RED_STAGES("v_add_f32")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                     : "s"(m2));
    else
        asm volatile(SV_RED_STAGES("v_max_f32")
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                     : "s"(m2));
    return v[0];
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

// Schedule of a period = the 16 steps of the block being decoded (cb), during which block nb = cb + 1 is produced.  Round 5: the
// production work is spread over ALL sixteen steps instead of sitting in five of them (the per-step stamps of round 4's kernel read
// 1040 cycles for a step that carries only MFMAs and 2500-2850 for the steps with the scaling / exponentials / logarithms of a whole
// block: both waves of a SIMD were in those at the same time, and a SIMD issues one vector instruction per four cycles whoever it
// comes from).  k = step:
//   every k             log-posteriors of row k of the block being decoded, made from its exponentials just before they are used
//                       (eight vector instructions in the shadow of the dynamic programme's LDS reads)
//   k = 0 .. MMA_LAST   the MFMAs of block nb, tile after tile (weight fragments SV_D pairs ahead)
//   k = FIN(n)          tile n's accumulators -> logits (in place), running row maxima         [the step after the tile's last MFMA]
//   k = FIN(3)          ... and the wave's maxima (butterfly, own LDS row)
//   k = E0 + n          tile n: exponentials (in place), running row sums                      [E0 = FIN(3) + 1]
//   k = E0 + 3          ... and the wave's sums
//   k = ROW_K           row statistics (waves 0-3; needs every wave's sums: the barrier of step E0 + 3) | operand images of block
//                       nb + 1 (waves 4-7)
//   k = ROW_K + 1       operand images of block nb + 1 (waves 0-3)
//   after step 15       the finished exponentials move to the registers the next period decodes from
template <int KS> struct SvSched {
    static constexpr int NP = 4 * KS;                           // weight fragment pairs per wave and block
    static constexpr int NMF = 3 * NP;                          // MFMAs per wave and block
#ifdef SV_MMA_LAST
    static constexpr int MMA_LAST = SV_MMA_LAST;
#else
    static constexpr int MMA_LAST = 8;
#endif
    static constexpr int NM = MMA_LAST + 1;
    static constexpr int MPS = (NMF + NM - 1) / NM;             // MFMAs per step
    // positions of a step that take MFMAs: two right behind the programme's LDS requests (their A images were requested at the end of the
    // step before), one in front of each of its eight chunks, two behind its writes -- where the wave would otherwise only wait for LDS and
    // for the barrier, and so that the two waves of a SIMD, which run the same program in lock step, do not ask the matrix pipe for
    // sixteen MFMAs within the 250 cycles of the chunks; position q takes MFMAs [q MPS / NPOS, (q + 1) MPS / NPOS) of the step
    static constexpr int NPOS = 12;
#ifndef SV_EDGE
#define SV_EDGE 0
#endif
    // first MFMA (of the step's MPS) at position q.  SV_EDGE: all of them in two bursts, right behind the programme's LDS requests and
    // right behind its writes -- while MFMAs execute the SIMD issues one vector instruction per ~4.75 cycles for BOTH its waves instead
    // of one per wave (tools/probes/coissue_probe.hip: a wave beside an MFMA-issuing partner runs at half rate), so the MFMAs belong
    // where both waves only wait (LDS round trip, barrier), not between the chunks of the programme
    static constexpr int slot_lo(int q)
    {
        if (SV_EDGE) return q <= 0 ? 0 : (q < NPOS - 1 ? MPS / 2 : (q == NPOS - 1 ? MPS / 2 : MPS));
        return (q * MPS) / NPOS;
    }
    static constexpr int slot_hi(int q) { return q >= NPOS - 1 ? MPS : slot_lo(q + 1); }
    static constexpr int mf_step(int m) { return m / MPS; }
    static constexpr int fin_step(int n) { return mf_step(3 * KS * (n + 1) - 1) + 1; }
    static constexpr int E0 = fin_step(3) + 1, SUM_K = E0 + 3, ROW_K = SUM_K + 1, LOADX_K = MMA_LAST + 2;
    // the traceback stores of the previous block: behind the last weight fragment (vmcnt counts in order: a wait for a fragment issued
    // after them would wait for the stores to reach memory)
    static constexpr int FLUSH_K = MMA_LAST + 1;
    static constexpr int WFIRST_K = SUM_K;                      // the first weight fragments of the NEXT block are requested here
    static_assert(ROW_K + 1 <= 15, "the production of a block must fit its period");
};

// One 512-thread workgroup decodes TWO chunks: lane half h (lanes 32h .. 32h+31) owns chunk h.
template <int KS, bool DUMP>
__global__ void __launch_bounds__(SV_THREADS) softmax_viterbi_kernel(const float *__restrict__ x, long ldx, int T, int B,
                                                                     const uint8_t *__restrict__ pack, float skip_pen,
                                                                     float min_prob, float one_m, uint8_t *__restrict__ tb,
                                                                     int32_t *__restrict__ best_out,
                                                                     float *__restrict__ score_out,
                                                                     const int *__restrict__ lens,
                                                                     float *__restrict__ lp_dump)
{
    constexpr int NCH = 2, BS = 16;                             // chunks per workgroup, steps per block
    using Sched = SvSched<KS>;
    constexpr int NP = Sched::NP;
    constexpr int OFF_V = 0;                                    // [chunk NCH][parity 2][SV_VP] float
    constexpr int OFF_TBS = OFF_V + NCH * 2 * SV_VP * 4;        // [parity 2][chunk NCH][step BS][256] uint8: the traceback of a block
    constexpr int OFF_A = OFF_TBS + 2 * 8 * 1024;               // [parity 2][KS][hi, lo][64 lanes][16 B]
    constexpr int OFF_CINV = OFF_A + 2 * KS * 2048;             // [1024] float
    constexpr int OFF_CBIAS = OFF_CINV + 4096;                  // [1024] float
    constexpr int OFF_W0 = OFF_CBIAS + 4096;                    // [16 KS] float, then the blank bias
    constexpr int OFF_XINV = OFF_W0 + 64 * KS + 16;             // [parity 2][half 2][16] float: inverse row scales
    constexpr int OFF_L0 = OFF_XINV + 256;                      // [parity 2][2][16] blank logits
    constexpr int OFF_REDA = OFF_L0 + 256;                      // [half 2][wave 8][16] maxima over a wave's 128 columns
    constexpr int OFF_REDB = OFF_REDA + 1024;                   // [2][8][16] sums of exp(logit - wave maximum)
    constexpr int OFF_FAC = OFF_REDB + 1024;                    // [parity 2][2][8][16] exp(wave maximum - row maximum) / row sum
    constexpr int OFF_LP0 = OFF_FAC + 2048;                     // [parity 2][2][16] blank log-posteriors
    constexpr int OFF_REDV = OFF_LP0 + 256;                     // [chunk NCH][8]
    constexpr int OFF_REDI = OFF_REDV + 128;                    // [chunk NCH][8]
    constexpr int SMEM = OFF_REDI + 128;
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM];
    // the 16 rows of a lane half: row i is step i of chunk h

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SV_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(SV_PRIO);
#endif
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

    uint8_t *const tbs = smem + OFF_TBS;
    const float *const lp0b = reinterpret_cast<const float *>(smem + OFF_LP0) + hch * 16;
    const int o_step = j + 8 * (j >> 6), o_skip = cc * 72 + q;
    const int o_own = (j >> 6) * SV_AS + 4 * (j & 63) + 8 * ((j & 63) >> 4);     // states 4j .. 4j+3: block a = j >> 6
    float *const vb = reinterpret_cast<float *>(smem + OFF_V) + hch * 2 * SV_VP;
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
    // acc: accumulators -> logits -> exponentials of the block in the making, in place ([tile n][row i]); ecur: the exponentials of
    // the block being decoded.  Their log-posteriors exist one row at a time (lpk), made in the step that consumes them.
    f32x16 acc[4];
    float ecur[4][16];
    float rst[16];                                             // running row maxima, then running row sums, of the block in the making
    float4 xr0, xr1;                                           // this lane's eight x values of a coming block
    // operand preparation: wave w, 16-lane row r4 handles MFMA row rho = 4 w + r4 = lane half (w & 1), row i of that half
    const int kb = lane & 15, r4 = lane >> 4;
    const int pa_half = wave & 1, pa_i = r4 + 4 * (wave >> 1), rho = 4 * wave + r4;
    const int pa_chunk = pa_half, pa_step = pa_i;
    const bool pa_act = kb < 2 * KS;
    const int pa_b = min(b0 + pa_chunk, B - 1);

    auto load_x = [&](int blk) __attribute__((always_inline)) {
        const int t = min(BS * blk + pa_step, T - 1);
        const float *src = x + ((size_t)t * B + pa_b) * ldx + 8 * min(kb, 2 * KS - 1);
        xr0 = *reinterpret_cast<const float4 *>(src);
        xr1 = *reinterpret_cast<const float4 *>(src + 4);
    };
    // x rows of block blk -> fp16 hi/lo A-operand images, row scales, blank logits (all double buffered by block parity)
    // (Round 5: one hand-scheduled sequence.  hipcc's version of the same arithmetic -- two 16-lane row reductions through DPP moves with
    //  canonicalised maxima and wait states, element-wise conversions -- kept a wave ~1030 cycles, alone on its SIMD; the row maximum
    //  and the blank column's dot product are independent chains and interleave here, the split is f16split.h's four instructions per
    //  pair.  Same values bit for bit: the dot product is the same single fused chain, the row sum takes the partners in the same order.)
    const float pa_actf = pa_act ? 1.0f : 0.0f;
    auto prepare_a = [&](int blk) __attribute__((always_inline)) {
        const int par = blk & 1;
        const float *w0 = reinterpret_cast<const float *>(smem + OFF_W0);
        const float4 wa = *reinterpret_cast<const float4 *>(w0 + 8 * min(kb, 2 * KS - 1));
        const float4 wb = *reinterpret_cast<const float4 *>(w0 + 8 * min(kb, 2 * KS - 1) + 4);
        const float b0 = w0[16 * KS];                           // the blank column's bias (requested with the weights: one LDS round trip)
        float x0 = xr0.x, x1 = xr0.y, x2 = xr0.z, x3 = xr0.w, x4 = xr1.x, x5 = xr1.y, x6 = xr1.z, x7 = xr1.w;
        float a0, a1, a2, a3, am, dot, inv, sc;
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        asm volatile("v_max_f32 %[a0], |%[x0]|, |%[x1]|\n\t"
                     "v_mul_f32 %[dot], %[x0], %[w0]\n\t"
                     "v_max_f32 %[a1], |%[x2]|, |%[x3]|\n\t"
                     "v_fma_f32 %[dot], %[x1], %[w1], %[dot]\n\t"
                     "v_max_f32 %[a2], |%[x4]|, |%[x5]|\n\t"
                     "v_fma_f32 %[dot], %[x2], %[w2], %[dot]\n\t"
                     "v_max_f32 %[a3], |%[x6]|, |%[x7]|\n\t"
                     "v_fma_f32 %[dot], %[x3], %[w3], %[dot]\n\t"
                     "v_max3_f32 %[am], %[a0], %[a1], %[a2]\n\t"
                     "v_fma_f32 %[dot], %[x4], %[w4], %[dot]\n\t"
                     "v_max_f32 %[am], %[am], %[a3]\n\t"
                     "v_fma_f32 %[dot], %[x5], %[w5], %[dot]\n\t"
                     "v_mul_f32 %[am], %[am], %[act]\n\t"                                     // lanes beyond K: 0
                     "v_fma_f32 %[dot], %[x6], %[w6], %[dot]\n\t"
                     "s_nop 0\n\t"
                     "v_fma_f32 %[dot], %[x7], %[w7], %[dot]\n\t"
                     "v_max_f32_dpp %[am], %[am], %[am] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mul_f32 %[dot], %[dot], %[act]\n\t"
                     "s_nop 0\n\t"
                     "v_max_f32_dpp %[am], %[am], %[am] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %[dot], %[dot], %[dot] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 0\n\t"
                     "v_max_f32_dpp %[am], %[am], %[am] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %[dot], %[dot], %[dot] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 0\n\t"
                     "v_max_f32_dpp %[am], %[am], %[am] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"  // the row's largest magnitude
                     "v_add_f32_dpp %[dot], %[dot], %[dot] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                     "v_bfe_u32 %[inv], %[am], 23, 8\n\t"                                     // pow2_scale (f16split.h)
                     "s_nop 0\n\t"
                     "v_add_f32_dpp %[dot], %[dot], %[dot] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"  // the row's blank-column product
                     "v_med3_u32 %[inv], %[inv], 27, %[c227]\n\t"
                     "v_lshlrev_b32 %[inv], 23, %[inv]\n\t"
                     "v_sub_u32 %[sc], 0x7f000000, %[inv]\n\t"
                     "v_mul_f32 %[sc], %[sc], %[act]\n\t"
                     "v_mul_f32 %[x0], %[x0], %[sc]\n\t"
                     "v_mul_f32 %[x1], %[x1], %[sc]\n\t"
                     "v_mul_f32 %[x2], %[x2], %[sc]\n\t"
                     "v_mul_f32 %[x3], %[x3], %[sc]\n\t"
                     "v_mul_f32 %[x4], %[x4], %[sc]\n\t"
                     "v_mul_f32 %[x5], %[x5], %[sc]\n\t"
                     "v_mul_f32 %[x6], %[x6], %[sc]\n\t"
                     "v_mul_f32 %[x7], %[x7], %[sc]\n\t"
                     "v_cvt_pk_f16_f32 %[h0], %[x0], %[x1]\n\t"
                     "v_cvt_pk_f16_f32 %[h1], %[x2], %[x3]\n\t"
                     "v_cvt_pk_f16_f32 %[h2], %[x4], %[x5]\n\t"
                     "v_cvt_pk_f16_f32 %[h3], %[x6], %[x7]\n\t"
                     "v_fma_mix_f32 %[x0], %[h0], -1.0, %[x0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x1], %[h0], -1.0, %[x1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x2], %[h1], -1.0, %[x2] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x3], %[h1], -1.0, %[x3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x4], %[h2], -1.0, %[x4] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x5], %[h2], -1.0, %[x5] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x6], %[h3], -1.0, %[x6] op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %[x7], %[h3], -1.0, %[x7] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                     "v_cvt_pk_f16_f32 %[l0], %[x0], %[x1]\n\t"
                     "v_cvt_pk_f16_f32 %[l1], %[x2], %[x3]\n\t"
                     "v_cvt_pk_f16_f32 %[l2], %[x4], %[x5]\n\t"
                     "v_cvt_pk_f16_f32 %[l3], %[x6], %[x7]"
                     : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6),
                       [x7] "+v"(x7), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [am] "=&v"(am), [dot] "=&v"(dot),
                       [inv] "=&v"(inv), [sc] "=&v"(sc), [h0] "=&v"(h0), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [l0] "=&v"(l0),
                       [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3)
                     : [w0] "v"(wa.x), [w1] "v"(wa.y), [w2] "v"(wa.z), [w3] "v"(wa.w), [w4] "v"(wb.x), [w5] "v"(wb.y), [w6] "v"(wb.z),
                       [w7] "v"(wb.w), [act] "v"(pa_actf), [c227] "s"(227));
        if (pa_act) {
            uint8_t *dst = smem + OFF_A + par * KS * 2048 + (kb >> 1) * 2048 + (rho + 32 * (kb & 1)) * 16;
            *reinterpret_cast<uint4 *>(dst) = make_uint4(h0, h1, h2, h3);
            *reinterpret_cast<uint4 *>(dst + 1024) = make_uint4(l0, l1, l2, l3);
        }
        if (kb == 0) {
            reinterpret_cast<float *>(smem + OFF_XINV)[par * 32 + pa_half * 16 + pa_i] = inv;
            reinterpret_cast<float *>(smem + OFF_L0)[par * 32 + pa_half * 16 + pa_i] = (dot + b0) * SV_LOG2E;
        }
    };
    auto wload = [&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NP && !(SV_ABL & 2)) {
            wfh[p % SV_D] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wwave + p * 2048, 0));
            wfl[p % SV_D] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, wwave + p * 2048 + 1024, 0));
        }
    };
    // MFMA m of a block = term m % 3 of fragment pair p = m / 3 = (tile n = p / KS, K block s = p % KS): three fp16 terms per product,
    // small terms first (gemm_rows_f16x3.hip): a_lo.w_hi, a_hi.w_lo, a_hi.w_hi.  The MFMAs are asm statements of their own, placed
    // BETWEEN the chunks of the hand-scheduled dynamic programme (asm volatile statements keep their order): round 4's kernel issued a
    // step's MFMAs and its programme one after the other, and the ablation builds showed their times adding up -- a wave issues in order,
    // the matrix pipe idled while the programme's fifty vector instructions went by and the vector unit while the wave waited for the pipe.
    // The accumulators are read-write operands even where C = 0: a freshly defined destination may be given registers that operands of
    // MFMAs still in the pipe have just vacated (round 3: rows computed from overwritten operands whenever the pipe was contended).
    half8 ah[NP], al[NP];                                      // A operand images of a pair: live from their LDS read to the pair's last MFMA
    auto aload = [&](auto pc, int apar) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < NP) {
            const uint8_t *ab = smem + OFF_A + apar * KS * 2048 + (p % KS) * 2048 + lane * 16;
            ah[p] = *reinterpret_cast<const half8 *>(ab);
            al[p] = *reinterpret_cast<const half8 *>(ab + 1024);
        }
    };
    auto mfma_m = [&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value;
        if constexpr (m < Sched::NMF && !(SV_ABL & 1)) {
            if constexpr (SV_ABL & 128) { if (wave >= 4) return; }          // (timing only: the MFMAs of one wave half alone)
            constexpr int p = m / 3, term = m % 3, n = p / KS;
            constexpr bool zero = (p % KS == 0) && term == 0;
            const half8 a = term == 0 ? al[p] : ah[p];
            const half8 w = term == 1 ? wfl[p % SV_D] : wfh[p % SV_D];
            // ("+&v": the accumulator is written while later passes still read A and B -- no operand may share its registers, which
            //  hipcc would otherwise allow where the accumulator's incoming value is undefined: tools/mfma_overlap_scan.py)
            if constexpr (zero) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "+&v"(acc[n]) : "v"(a), "v"(w));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+&v"(acc[n]) : "v"(a), "v"(w));
            if constexpr (term == 2) wload(ic<p + SV_D>{});
        }
    };
    // the MFMAs of step k at position c8: a step's MPS MFMAs in their order
    auto mfma_run = [&](auto lo_c, auto hi_c, auto &&self) __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        if constexpr (lo < hi) {
            mfma_m(ic<lo>{});
            self(ic<lo + 1>{}, hi_c, self);
        }
    };
    auto mfma_slot = [&](auto kc, auto cc_, int apar) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, c8 = decltype(cc_)::value;
        if constexpr (k <= Sched::MMA_LAST) {
            constexpr int base = k * Sched::MPS;
            mfma_run(ic<base + Sched::slot_lo(c8)>{}, ic<base + Sched::slot_hi(c8)>{}, mfma_run);
        }
    };
    // the pairs whose first MFMA lies in chunk c8 of step k: their A images are requested a chunk earlier (at the top of the step for
    // chunks 0 and 1; a pair that straddles the top of a step was requested in the step before)
    auto aload_run = [&](auto lo_c, auto hi_c, int apar, auto &&self) __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        if constexpr (lo < hi) {
            if constexpr (lo < Sched::NMF && lo % 3 == 0) aload(ic<lo / 3>{}, apar);
            self(ic<lo + 1>{}, hi_c, apar, self);
        }
    };
    auto aload_for = [&](auto kc, auto cc_, int apar) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value, c8 = decltype(cc_)::value;
        if constexpr (k <= Sched::MMA_LAST && c8 < Sched::NPOS) {
            constexpr int base = k * Sched::MPS;
            aload_run(ic<base + Sched::slot_lo(c8)>{}, ic<base + Sched::slot_hi(c8)>{}, apar, aload_run);
        }
    };
    // the A images for position P of the block's linear position count (step P / NPOS, position P % NPOS): requested SV_AHEAD positions
    // before the MFMA that opens the pair -- an LDS read takes 64 - 130 cycles, a position ~35, and the MFMA behind a late read waits
    auto aload_pos = [&](auto Pc, int apar) __attribute__((always_inline)) {
        constexpr int P = decltype(Pc)::value;
        if constexpr (P >= 0 && P / Sched::NPOS <= Sched::MMA_LAST) aload_for(ic<P / Sched::NPOS>{}, ic<P % Sched::NPOS>{}, apar);
    };
    auto wload_first = [&](auto pc, auto &&self) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < SV_D) {
            wload(pc);
            self(ic<p + 1>{}, self);
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
    const float *const my_fac = reinterpret_cast<const float *>(smem + OFF_FAC) + (hch * 8 + wave) * 16;      // + 256 * parity
    // scaled accumulators of tile n -> logits in place (gemm_rows_f16x3.hip's finish), running row maxima
    auto fin_tile = [&](auto nc, int nb) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        float xinv[16];
        load16p(reinterpret_cast<const float *>(smem + OFF_XINV) + (nb & 1) * 32 + hch * 16, xinv);
        const float civ = reinterpret_cast<const float *>(smem + OFF_CINV)[4 * j + n];
        const float cbv = reinterpret_cast<const float *>(smem + OFF_CBIAS)[4 * j + n];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 a = {acc[n][i], acc[n][i + 1]}, xi = {xinv[i], xinv[i + 1]};
            const f32x2 v = __builtin_elementwise_fma(a * xi, f32x2{civ, civ}, f32x2{cbv, cbv});
            acc[n][i] = v.x;
            acc[n][i + 1] = v.y;
        }
        // (sv_max: fmaxf would canonicalise both operands first -- three instructions per maximum; its only reader is reduce_max's asm)
#pragma unroll
        for (int i = 0; i < 16; i++) rst[i] = n == 0 ? acc[0][i] : sv_max(rst[i], acc[n][i]);
        // pinned to this step: the values are first needed steps later, and the compiler would sink the whole tile there
        asm volatile("" : "+v"(acc[n]));
#pragma unroll
        for (int i = 0; i < 16; i++) keepf(rst[i]);
    };
    auto reduce_max = [&]() __attribute__((always_inline)) {
        const float r = (SV_ABL & 16) ? rst[0] + rst[15] : sv_half_reduce16_asm<false>(rst, 0xCCCCCCCCCCCCCCCCull);
        if (!(c & 1)) my_max[c >> 1] = r;
    };
    // tile n: exponentials in place, running row sums (the order of round 4's sum: ((t0 + t1) + t2) + t3)
    auto exp_tile = [&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value;
        asm volatile("" ::: "memory");                         // the wave's own LDS writes (reduce_max), read back in order
        float m[16];
        load16p(my_max, m);
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 d = f32x2{acc[n][i], acc[n][i + 1]} - f32x2{m[i], m[i + 1]};
            acc[n][i] = (SV_ABL & 8) ? d.x : __builtin_amdgcn_exp2f(d.x);
            acc[n][i + 1] = (SV_ABL & 8) ? d.y : __builtin_amdgcn_exp2f(d.y);
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            if constexpr (n == 0) {
                rst[i] = acc[0][i];
                rst[i + 1] = acc[0][i + 1];
            } else {
                const f32x2 t = f32x2{rst[i], rst[i + 1]} + f32x2{acc[n][i], acc[n][i + 1]};
                rst[i] = t.x;
                rst[i + 1] = t.y;
            }
        }
        asm volatile("" : "+v"(acc[n]));                       // pinned to this step (see fin_tile)
#pragma unroll
        for (int i = 0; i < 16; i++) keepf(rst[i]);
    };
    auto reduce_sum = [&]() __attribute__((always_inline)) {
        const float r = (SV_ABL & 16) ? rst[0] + rst[15] : sv_half_reduce16_asm<true>(rst, 0xCCCCCCCCCCCCCCCCull);
        if (!(c & 1)) my_sum[c >> 1] = r;
    };
    auto row_stats = [&](int nb) __attribute__((always_inline)) {                             // 256 lanes: (half, row) x the eight waves' shares
        if (tid < 256) {
            const int hh = tid >> 7, ri = (tid >> 3) & 15, w = tid & 7, row = hh * 16 + ri;
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
            const int tc = hh ? Tcs[1] : Tcs[0];
            const bool dead = BS * nb + ri >= tc;
            const float inv = dead ? __builtin_nanf("") : 1.0f / ssum;
            reinterpret_cast<float *>(smem + OFF_FAC)[(nb & 1) * 256 + (hh * 8 + w) * 16 + ri] = ew * (inv * one_m);
            if (w == 0) {
                const float lb = dead ? 0.0f : sv_log(fmaf(e0, inv * one_m, mp_eta));
                reinterpret_cast<float *>(smem + OFF_LP0)[(nb & 1) * 32 + row] = lb;
#ifndef SV_DIAG
                if constexpr (DUMP) {
                    const int t = BS * nb + ri, bb = b0 + hh;
                    if (t < T && bb < B) lp_dump[((size_t)t * B + bb) * (SV_NK + 1)] = lb;
                }
#endif
            }
        }
    };
    // vector side work of step k of a period: the production of block nb (and the operand images of block nb + 1)
    auto side = [&](auto kc, int nb) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k == Sched::LOADX_K) load_x(nb + 1);
        if constexpr (k == Sched::fin_step(0)) fin_tile(ic<0>{}, nb);
        if constexpr (k == Sched::fin_step(1)) fin_tile(ic<1>{}, nb);
        if constexpr (k == Sched::fin_step(2)) fin_tile(ic<2>{}, nb);
        if constexpr (k == Sched::fin_step(3)) {
            fin_tile(ic<3>{}, nb);
            reduce_max();
        }
        if constexpr (k == Sched::E0) exp_tile(ic<0>{});
        if constexpr (k == Sched::E0 + 1) exp_tile(ic<1>{});
        if constexpr (k == Sched::E0 + 2) exp_tile(ic<2>{});
        if constexpr (k == Sched::E0 + 3) {
            exp_tile(ic<3>{});
            reduce_sum();
        }
        if constexpr (k == Sched::WFIRST_K) wload_first(ic<0>{}, wload_first);
        if constexpr (!(SV_ABL & 32)) {
            if constexpr (k == Sched::ROW_K) {
                if (wave < 4) row_stats(nb);
                else prepare_a(nb + 1);
            }
            if constexpr (k == Sched::ROW_K + 1) {
                if (wave < 4) prepare_a(nb + 1);
            }
        } else {
            if constexpr (k == Sched::ROW_K) row_stats(nb);
        }
    };

    // ---- the dynamic programme: step t0 + k of block cb (viterbi_forward4_kernel::step, decode.hip) ----
    // The arithmetic of one step for this lane's four to-states, written out as hand-scheduled instruction sequences (chunks J, D1 .. D8;
    // the step's MFMAs go between them): hipcc's own schedule of the same instructions carried ~20 wait states per step (a v_cndmask two
    // instructions behind the v_cmp that made its mask, a DPP instruction behind the write of its source: 21 cycles per compare-select
    // pair in tools/probes/coissue_probe.hip against 9.5 for two independent instructions) plus the waits of a dependent chain.  Here every
    // consumer sits at least two instructions behind its producer (three behind a mask or a DPP source), across chunk boundaries too.
    //   step maximum over a and the first a that attains it (np.argmax, decode.py:67-68); skip maximum over ab = a*4 + b
    //   (decode.py:72-73: this lane's share is b = cc, the quad holds the other three; the first ab that attains the maximum is the
    //   smallest key among the lanes whose share attains it); score = lp + max(step, skip - pen) against stay = own + lp0, a tie
    //   stays (decode.py:75-81); "move ? nv : stay" is v_max_f32(nv, stay): the same value whether nv > stay or not, and a NaN nv (a
    //   chunk past its end) returns stay like the compare does.  Traceback word: two bits per to-state (0 stay, 1 step, 2 skip), the
    //   step argument at bit 8, the skip argument at bit 10.
    float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;          // this thread's four scores of the previous step
    const float mp_eta_s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(mp_eta)));
    const float ln2_s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(SV_LN2)));
    // J: exponentials of a row -> log-posteriors (decode.py:36 and :56), made at the END of the step before the one that consumes them,
    // behind that step's LDS writes: the wave would otherwise only wait for them and for the barrier.  t = the row's time step.
    float ln0 = 0.0f, ln1 = 0.0f, ln2 = 0.0f, ln3 = 0.0f;
    auto jit_log = [&](float e0, float e1, float e2, float e3, float fac, int t) __attribute__((always_inline)) {
        asm volatile("v_fma_f32 %[l0], %[e0], %[fac], %[mpe]\n\t"
                     "v_fma_f32 %[l1], %[e1], %[fac], %[mpe]\n\t"
                     "v_fma_f32 %[l2], %[e2], %[fac], %[mpe]\n\t"
                     "v_fma_f32 %[l3], %[e3], %[fac], %[mpe]\n\t"
                     "v_log_f32 %[l0], %[l0]\n\t"
                     "v_log_f32 %[l1], %[l1]\n\t"
                     "v_log_f32 %[l2], %[l2]\n\t"
                     "v_log_f32 %[l3], %[l3]\n\t"
                     "v_mul_f32 %[l0], %[ln2], %[l0]\n\t"
                     "v_mul_f32 %[l1], %[ln2], %[l1]\n\t"
                     "v_mul_f32 %[l2], %[ln2], %[l2]\n\t"
                     "v_mul_f32 %[l3], %[ln2], %[l3]"
                     : [l0] "=&v"(ln0), [l1] "=&v"(ln1), [l2] "=&v"(ln2), [l3] "=&v"(ln3)
                     : [e0] "v"(e0), [e1] "v"(e1), [e2] "v"(e2), [e3] "v"(e3), [fac] "v"(fac), [mpe] "s"(mp_eta_s), [ln2] "s"(ln2_s));
#ifndef SV_DIAG
        if constexpr (DUMP) {
            const int bb = b0 + hch;
            if (t < T && bb < B) {
                float *dst = lp_dump + ((size_t)t * B + bb) * (SV_NK + 1) + 1 + 4 * j;
                dst[0] = ln0; dst[1] = ln1; dst[2] = ln2; dst[3] = ln3;
            }
        }
#endif
    };
    // one whole step: reads, the step's vector side work, the chunks with the MFMAs between them, writes
    auto step = [&](auto kc, auto dpc, auto prodc, int t0, int par, int nb) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        constexpr bool DP = decltype(dpc)::value && !(SV_ABL & 4), PROD = decltype(prodc)::value;
        const int apar = nb & 1;
        [[maybe_unused]] float vs0, vs1, vs2, vs3, vk0, vk1, vk2, vk3, lp0, l0, l1, l2, l3, t0r, t1r, t2r, fac;
        [[maybe_unused]] float ls0, ls1, ls2, ls3;
        [[maybe_unused]] unsigned long long sA, sB, sC, sD, sE, sF, sG;
        if constexpr (DP) {
            const float *vold = vb + ((k & 1) ^ 1) * SV_VP;
            vs0 = vold[o_step]; vs1 = vold[SV_AS + o_step]; vs2 = vold[2 * SV_AS + o_step]; vs3 = vold[3 * SV_AS + o_step];
            vk0 = vold[o_skip]; vk1 = vold[SV_AS + o_skip]; vk2 = vold[2 * SV_AS + o_skip]; vk3 = vold[3 * SV_AS + o_skip];
            lp0 = lp0b[par * 32 + k];
            // the factor of the next row (for J at the end of this step): of this block, or (last step) of the block the next period decodes
            fac = k < 15 ? my_fac[par * 256 + k + 1] : my_fac[(par ^ 1) * 256];
        }
        if constexpr (PROD) {
            if constexpr (k == 0) {                             // (the images of a block are made in the last steps of the period before)
                static_for_sv<0, SV_AHEAD>([&](auto qc) { aload_pos(ic<decltype(qc)::value>{}, apar); });
            }
            mfma_slot(kc, ic<0>{}, apar);
            aload_pos(ic<k * Sched::NPOS + 0 + SV_AHEAD>{}, apar);
            mfma_slot(kc, ic<1>{}, apar);
            aload_pos(ic<k * Sched::NPOS + 1 + SV_AHEAD>{}, apar);
        }
        if constexpr (DP) {
            l0 = ln0; l1 = ln1; l2 = ln2; l3 = ln3;             // made at the end of the step before (J)
            if constexpr (k == 0) { ls0 = l0; ls1 = l1; ls2 = l2; ls3 = l3; }
        }
        [[maybe_unused]] const int cb = nb - 1;                 // (SV_STAMP2 names it)
        if constexpr (k == 14) SV_STAMP2(0);
        if constexpr (k == 15) SV_STAMP2(4);
        if constexpr (PROD) {
            if constexpr (SV_ABL & 64) { if (wave < 4) side(kc, nb); }       // (timing only: what the side work costs one wave half alone)
            else side(kc, nb);
        }
        if constexpr (k == 14) SV_STAMP2(1);
        if constexpr (k == 15) SV_STAMP2(5);
        // ---- D1 ----
        if constexpr (PROD) { mfma_slot(kc, ic<2>{}, apar); aload_pos(ic<k * Sched::NPOS + 2 + SV_AHEAD>{}, apar); }
        if constexpr (DP)
            asm volatile("v_max3_f32 %[t0], %[vs0], %[vs1], %[vs2]\n\t"
                         "v_max3_f32 %[t1], %[vk0], %[vk1], %[vk2]\n\t"
                         "v_add_f32 %[o0], %[o0], %[lp0]\n\t"
                         "v_max_f32 %[t0], %[t0], %[vs3]\n\t"                                                   // sstep
                         "v_max_f32 %[t1], %[t1], %[vk3]\n\t"                                                   // kpart
                         "v_add_f32 %[o1], %[o1], %[lp0]\n\t"
                         "v_add_f32 %[o2], %[o2], %[lp0]"
                         : [t0] "=&v"(t0r), [t1] "=&v"(t1r), [o0] "+v"(o0), [o1] "+v"(o1), [o2] "+v"(o2)
                         : [vs0] "v"(vs0), [vs1] "v"(vs1), [vs2] "v"(vs2), [vs3] "v"(vs3), [vk0] "v"(vk0), [vk1] "v"(vk1), [vk2] "v"(vk2),
                           [vk3] "v"(vk3), [lp0] "v"(lp0));
        // ---- D2 ----
        if constexpr (PROD) { mfma_slot(kc, ic<3>{}, apar); aload_pos(ic<k * Sched::NPOS + 3 + SV_AHEAD>{}, apar); }
        if constexpr (DP)
            asm volatile("v_cmp_eq_f32 %[sA], %[vs2], %[t0]\n\t"
                         "v_max_f32_dpp %[t2], %[t1], %[t1] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_cmp_eq_f32 %[sB], %[vs1], %[t0]\n\t"
                         "v_cmp_eq_f32 %[sC], %[vs0], %[t0]\n\t"
                         "v_add_f32 %[o3], %[o3], %[lp0]\n\t"
                         "v_max_f32_dpp %[t2], %[t2], %[t2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"       // kbest
                         : [sA] "=&s"(sA), [sB] "=&s"(sB), [sC] "=&s"(sC), [t2] "=&v"(t2r), [o3] "+v"(o3)
                         : [vs0] "v"(vs0), [vs1] "v"(vs1), [vs2] "v"(vs2), [t0] "v"(t0r), [t1] "v"(t1r), [lp0] "v"(lp0));
        // ---- D3 ----  (vs2 becomes the step argument, vs3 the skip score)
        if constexpr (PROD) { mfma_slot(kc, ic<4>{}, apar); aload_pos(ic<k * Sched::NPOS + 4 + SV_AHEAD>{}, apar); }
        if constexpr (DP)
            asm volatile("v_cmp_eq_f32 %[sD], %[vk2], %[t1]\n\t"
                         "v_cmp_eq_f32 %[sE], %[vk1], %[t1]\n\t"
                         "v_cmp_eq_f32 %[sF], %[vk0], %[t1]\n\t"
                         "v_cndmask_b32 %[vs2], 3, 2, %[sA]\n\t"
                         "v_subrev_f32 %[vs3], %[pen], %[t2]\n\t"                                               // sskip = kbest - skip_pen
                         "v_cmp_eq_f32 %[sG], %[t1], %[t2]\n\t"                                                 // my share attains the quad's maximum
                         "v_cndmask_b32 %[vs2], %[vs2], 1, %[sB]"
                         : [sD] "=&s"(sD), [sE] "=&s"(sE), [sF] "=&s"(sF), [sG] "=&s"(sG), [vs2] "=&v"(vs2), [vs3] "=&v"(vs3)
                         : [vk0] "v"(vk0), [vk1] "v"(vk1), [vk2] "v"(vk2), [t1] "v"(t1r), [t2] "v"(t2r), [sA] "s"(sA), [sB] "s"(sB),
                           [pen] "s"(skip_pen));
        // ---- D4 ----  (t2 becomes mx, vk2 my first skip maximum's a*4, sA "by step")
        if constexpr (PROD) { mfma_slot(kc, ic<5>{}, apar); aload_pos(ic<k * Sched::NPOS + 5 + SV_AHEAD>{}, apar); }
        if constexpr (DP)
            asm volatile("v_max_f32 %[t2], %[t0], %[vs3]\n\t"                                                   // mx
                         "v_cndmask_b32 %[vk2], 12, 8, %[sD]\n\t"
                         "v_cmp_gt_f32 %[sA], %[t0], %[vs3]\n\t"                                                // by step (a tie skips)
                         "v_cndmask_b32 %[vs2], %[vs2], 0, %[sC]\n\t"                                           // step argument
                         "v_cndmask_b32 %[vk2], %[vk2], 4, %[sE]\n\t"
                         "v_add_f32 %[l0], %[l0], %[t2]\n\t"
                         "v_cndmask_b32 %[vk2], %[vk2], 0, %[sF]"
                         : [t2] "=&v"(t2r), [vk2] "=&v"(vk2), [sA] "=&s"(sA), [vs2] "+v"(vs2), [l0] "+v"(l0)
                         : [t0] "v"(t0r), [vs3] "v"(vs3), [sC] "s"(sC), [sD] "s"(sD), [sE] "s"(sE), [sF] "s"(sF));
        // ---- D5 ----
        if constexpr (PROD) { mfma_slot(kc, ic<6>{}, apar); aload_pos(ic<k * Sched::NPOS + 6 + SV_AHEAD>{}, apar); }
        if constexpr (DP)
            asm volatile("v_add_f32 %[l1], %[l1], %[t2]\n\t"
                         "v_or_b32 %[vk2], %[vk2], %[cc]\n\t"                                                   // a*4 + b of my first maximum
                         "v_add_f32 %[l2], %[l2], %[t2]\n\t"
                         "v_cndmask_b32 %[vk2], 16, %[vk2], %[sG]\n\t"
                         "v_add_f32 %[l3], %[l3], %[t2]\n\t"
                         "v_cmp_gt_f32 %[sB], %[l0], %[o0]\n\t"                                                 // move (a tie stays)
                         "v_cmp_gt_f32 %[sC], %[l1], %[o1]"
                         : [l1] "+v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [vk2] "+v"(vk2), [sB] "=&s"(sB), [sC] "=&s"(sC)
                         : [t2] "v"(t2r), [cc] "v"(cc), [sG] "s"(sG), [l0] "v"(l0), [o0] "v"(o0), [o1] "v"(o1));
        // ---- D6 ----
        if constexpr (PROD) { mfma_slot(kc, ic<7>{}, apar); aload_pos(ic<k * Sched::NPOS + 7 + SV_AHEAD>{}, apar); }
        if constexpr (DP)
            asm volatile("v_min_i32_dpp %[vk2], %[vk2], %[vk2] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_cmp_gt_f32 %[sD], %[l2], %[o2]\n\t"
                         "v_cmp_gt_f32 %[sE], %[l3], %[o3]\n\t"
                         "v_max_f32 %[o0], %[l0], %[o0]\n\t"
                         "v_min_i32_dpp %[vk2], %[vk2], %[vk2] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"   // skip argument
                         "v_max_f32 %[o1], %[l1], %[o1]"
                         : [vk2] "+v"(vk2), [sD] "=&s"(sD), [sE] "=&s"(sE), [o0] "+v"(o0), [o1] "+v"(o1)
                         : [l0] "v"(l0), [l1] "v"(l1), [l2] "v"(l2), [l3] "v"(l3), [o2] "v"(o2), [o3] "v"(o3));
        // ---- D7 ----  (l0 .. l3 become the move bits)
        if constexpr (PROD) { mfma_slot(kc, ic<8>{}, apar); aload_pos(ic<k * Sched::NPOS + 8 + SV_AHEAD>{}, apar); }
        if constexpr (DP) {
            asm volatile("v_cndmask_b32 %[l0], 0, 1, %[sB]\n\t"
                         "v_cndmask_b32 %[l1], 0, 2, %[sC]\n\t"
                         "v_max_f32 %[o2], %[l2], %[o2]\n\t"
                         "v_max_f32 %[o3], %[l3], %[o3]\n\t"
                         "v_cndmask_b32 %[l2], 0, 4, %[sD]\n\t"
                         "v_cndmask_b32 %[l3], 0, 8, %[sE]\n\t"
                         "v_cndmask_b32 %[t0], 0, 16, %[sA]"                                                    // bit 4: by step (else by skip)
                         : [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "+v"(l2), [l3] "+v"(l3), [o2] "+v"(o2), [o3] "+v"(o3), [t0] "=&v"(t0r)
                         : [sA] "s"(sA), [sB] "s"(sB), [sC] "s"(sC), [sD] "s"(sD), [sE] "s"(sE));
            if constexpr (k == 0) {                                 // t = 0: v = lpost[0][1:] (decode.py:57)
                if (t0 == 0) { o0 = ls0; o1 = ls1; o2 = ls2; o3 = ls3; }
            }
            float *vnew = vb + (k & 1) * SV_VP + o_own;
            vnew[0] = o0; vnew[1] = o1; vnew[2] = o2; vnew[3] = o3;
        }
        // ---- D8 ----
        if constexpr (PROD) { mfma_slot(kc, ic<9>{}, apar); aload_pos(ic<k * Sched::NPOS + 9 + SV_AHEAD>{}, apar); }
        if constexpr (DP) {
            // the traceback byte of this lane's four to-states (decode.hip: viterbi_backtrace_kernel FMT 2): bits 0-3 moves, bit 4 by step,
            // bits 5-6 the step argument, bit 7 = bit cc of the skip argument (the quad's four lanes hold the same one: a bit each)
            asm volatile("v_or3_b32 %[l0], %[l0], %[l1], %[l2]\n\t"
                         "v_lshl_or_b32 %[l3], %[vs2], 5, %[l3]\n\t"
                         "v_bfe_u32 %[vk2], %[vk2], %[cc], 1\n\t"
                         "v_or3_b32 %[l0], %[l0], %[l3], %[t0]\n\t"
                         "v_lshl_or_b32 %[l0], %[vk2], 7, %[l0]"
                         : [l0] "+v"(l0), [vk2] "+v"(vk2), [l3] "+v"(l3)
                         : [l1] "v"(l1), [l2] "v"(l2), [vs2] "v"(vs2), [t0] "v"(t0r), [cc] "v"(cc)
                         : "memory");
            tbs[((par * NCH + hch) * BS + k) * 256 + j] = (uint8_t)__float_as_uint(l0);
        }
        if constexpr (k == 14) SV_STAMP2(2);
        if constexpr (k == 15) SV_STAMP2(6);
        if constexpr (DP) {                                     // J for the next step's row
            if constexpr (k < 15) jit_log(ecur[0][k + 1], ecur[1][k + 1], ecur[2][k + 1], ecur[3][k + 1], fac, t0 + k + 1);
            else if constexpr (PROD) jit_log(acc[0][0], acc[1][0], acc[2][0], acc[3][0], fac, t0 + 16);    // row 0 of the block just finished
        }
        if constexpr (PROD) {                                   // behind the writes: the wave would only wait for them and the barrier
            mfma_slot(kc, ic<10>{}, apar);
            aload_pos(ic<k * Sched::NPOS + 10 + SV_AHEAD>{}, apar);
            mfma_slot(kc, ic<11>{}, apar);
            aload_pos(ic<k * Sched::NPOS + 11 + SV_AHEAD>{}, apar);
        }
        if constexpr (k == 14) SV_STAMP2(3);
        if constexpr (k == 15) SV_STAMP2(7);
    };
    // rows of block blk (staged with parity par) -> HBM: per chunk BS rows of 256 bytes, contiguous on both sides (8 KB: 16 bytes per thread)
    auto flush_tb = [&](int blk, int par) __attribute__((always_inline)) {
        constexpr int PER_CHUNK = BS * 256;                         // bytes
        const int off = tid * 16;
        const int ch = off / PER_CHUNK, rest = off % PER_CHUNK, t = BS * blk + rest / 256;
        const int tc = ch ? Tcs[1] : Tcs[0];
        uint8_t *dst = tb + ((size_t)(b0 + ch) * Tpad + BS * blk) * (SV_NK / 4) + rest;
        if (t >= 1 && t < tc)
            *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(smem + OFF_TBS + par * 8192 + off);
    };

    auto period = [&](auto dpc, auto prodc, int cb) __attribute__((always_inline)) {
        constexpr bool DP = decltype(dpc)::value, PROD = decltype(prodc)::value;
        const int t0 = BS * cb, par = cb & 1, nb = cb + 1;
#define SV_STEP(K)                                                                         \
    do {                                                                                   \
        step(ic<K>{}, dpc, prodc, t0, par, nb);                                            \
        bar();                                                                             \
        SV_STAMP(K);                                                                       \
        if constexpr ((K) == Sched::FLUSH_K) {                                             \
            if (DP && cb >= 1) flush_tb(cb - 1, par ^ 1);                                  \
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
        if constexpr (PROD) {                                   // the block just finished becomes the block being decoded
#pragma unroll
            for (int n = 0; n < 4; n++)
#pragma unroll
                for (int i = 0; i < 16; i++) ecur[n][i] = acc[n][i];
            if constexpr (!DP) jit_log(acc[0][0], acc[1][0], acc[2][0], acc[3][0], my_fac[0], 0);    // (prologue) row 0 of block 0
        }
    };

    // the accumulators exist (as defined values) from here on: an accumulator that first comes into being at its first MFMA may be given
    // registers that operands of the MFMAs just issued have vacated (tools/mfma_overlap_scan.py, the hazard of round 3)
#pragma unroll
    for (int n = 0; n < 4; n++) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[n][i] = 0.0f;
        asm volatile("" : "+v"(acc[n]));
    }
    load_x(0);
    bar();                                                      // constants staged
    prepare_a(0);
    wload_first(ic<0>{}, wload_first);
    bar();
    period(std::false_type{}, std::true_type{}, -1);
    for (int cb = 0; cb + 1 < nblk; cb++) period(std::true_type{}, std::true_type{}, cb);
    period(std::true_type{}, std::false_type{}, nblk - 1);
    flush_tb(nblk - 1, (nblk - 1) & 1);

    // ---- first argmax of the final scores (np.argmax, decode.py:85); the last step of a block writes parity 1 ----
    {
        const int ch = hch;
        const float4 fv = *reinterpret_cast<const float4 *>(&vb[SV_VP + o_own]);
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

template <int KS>
static int sv_launch(const float *x, long ldx, int T, int B, const uint8_t *pack, float skip_pen, float min_prob, uint8_t *tb,
                     int32_t *best, float *score_out, const int *lens, float *lp_dump, hipStream_t s)
{
#ifdef SV_DIAG
    constexpr bool diag = true;
#else
    constexpr bool diag = false;
#endif
    if (lp_dump && !diag)
        hipLaunchKernelGGL((softmax_viterbi_kernel<KS, true>), dim3((B + 1) / 2), dim3(SV_THREADS), 0, s, x, ldx, T, B, pack,
                           skip_pen, min_prob, (float)(1.0 - (double)min_prob), tb, best, score_out, lens, lp_dump);
    else
        hipLaunchKernelGGL((softmax_viterbi_kernel<KS, false>), dim3((B + 1) / 2), dim3(SV_THREADS), 0, s, x, ldx, T, B, pack,
                           skip_pen, min_prob, (float)(1.0 - (double)min_prob), tb, best, score_out, lens, lp_dump);
    return slk_launch_status();
}

// the fused path's own workspace: one traceback byte per FOUR k-mers (256 B per step and chunk), then best[B]
extern "C" size_t slk_softmax_viterbi_workspace_bytes(int T, int B, int nbase, int klen)
{
    if (T < 1 || B < 1 || nbase != 4 || klen != 5) return 0;
    const size_t tb = ((size_t)B * T * (SV_NK / 4) + 255) & ~(size_t)255;
    return tb + sizeof(int32_t) * (size_t)B + 256;
}

extern "C" int slk_softmax_viterbi_f32(const float *x, long ldx, const void *pack, int K, int T, int B, int nbase, int klen,
                                       float skip_pen, float min_prob, const int32_t *lens, int plan, void *workspace,
                                       size_t workspace_bytes, float *score_out, int32_t *path_out, int32_t *len_out,
                                       float *lp_dump, slk_stream_t stream)
{
    if (!x || !pack || !score_out || !path_out || !len_out || T < 1 || B < 1 || K < 1 || ldx < K || nbase < 2 || klen < 3 ||
        (plan != 0 && plan != 2 && plan != 4))
        return SLK_ERR_INVALID_ARG;
    // four chunks per workgroup (rounds 3-4, -DSV_WITH_NCH4) measured equal to two -- a SIMD issues one vector instruction per four
    // cycles whichever of its waves it comes from, and that plan only moved instructions from one wave to the other -- and was removed
    if (plan == 4) return SLK_ERR_UNSUPPORTED;
    if (!sv_shape_ok(K, nbase, klen) || (ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return SLK_ERR_UNSUPPORTED;
    const size_t need = slk_softmax_viterbi_workspace_bytes(T, B, nbase, klen);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return SLK_ERR_WORKSPACE;
    uint8_t *tb = static_cast<uint8_t *>(workspace);
    const size_t tbbytes = ((size_t)B * T * (SV_NK / 4) + 255) & ~(size_t)255;
    int32_t *best = reinterpret_cast<int32_t *>(tb + tbbytes);
    const uint8_t *pk = static_cast<const uint8_t *>(pack);
    hipStream_t s = slk_stream(stream);
    int rc;
    switch (K / 16) {
#define SV_CASE(KS)                                                                                                              \
    case KS: rc = sv_launch<KS>(x, ldx, T, B, pk, skip_pen, min_prob, tb, best, score_out, lens, lp_dump, s); break;
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
    return slk_backtrace_packed8(tb, best, T, B, SV_NK, path_out, len_out, lens, s);
#endif
}
