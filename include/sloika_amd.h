/*
 * sloika_amd.h -- C ABI of the MI355X (gfx950) basecalling hot path.
 *
 * Drop-in boundary for ONE path of nanoporetech/sloika:
 *     chunkify/normalise -> conv front end -> stacked GRU/LSTM -> softmax -> k-mer Viterbi decode
 *     (+ the transducer remap DP).
 * The reference implements this path in Python/numpy on top of Theano-generated code plus one Cython
 * function; each entry point below names the reference interface (file:line, relative to the sloika
 * checkout) it replaces.  INTEGRATION.md shows the binding a sloika maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types.
 *   - ALL pointers are DEVICE pointers (HIP) unless the parameter is documented as host.
 *     The library never allocates, frees or synchronises; scratch space is passed in by the caller
 *     (size from the matching *_workspace_bytes function).
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*; NULL = the null stream)
 *     and returns immediately.  Safe to capture into a hipGraph.
 *   - return value: SLK_OK (0) or a negative SLK_ERR_* code; nothing throws.
 *   - tensors are float32, C order [time][batch][feature] (sloika/layers.py:13-14).
 *   - thread-safe and re-entrant: no global state.
 */
#ifndef SLOIKA_AMD_H
#define SLOIKA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLK_ABI_VERSION 1
/* Every entry point below is exported; nothing else in libsloika_amd.so is (the library is built with
 * -fvisibility=hidden, tests/test_cabi.py compares the export table with this header both ways).          */
#define SLK_API __attribute__((visibility("default")))

#define SLK_OK 0
#define SLK_ERR_INVALID_ARG (-1) /* bad shape / null pointer: the reference raises AssertionError here      */
#define SLK_ERR_UNSUPPORTED (-2) /* valid request this build has no kernel for                               */
#define SLK_ERR_LAUNCH (-3)      /* HIP reported a launch error                                              */
#define SLK_ERR_WORKSPACE (-4)   /* workspace pointer null or too small                                      */
#define SLK_ERR_NO_DEVICE (-5)   /* no HIP device visible to this process                                    */

/* Activation ids: one per function of sloika/activation.py:8-115 (same order as the file).                  */
enum slk_activation {
    SLK_ACT_LINEAR = 0,      /* activation.py:8   */
    SLK_ACT_TANH = 1,        /* activation.py:52  */
    SLK_ACT_SIGMOID = 2,     /* activation.py:56  */
    SLK_ACT_ELU = 3,         /* activation.py:38  */
    SLK_ACT_RELU = 4,        /* activation.py:12  */
    SLK_ACT_RELU_SMOOTH = 5, /* activation.py:16  */
    SLK_ACT_SOFTPLUS = 6,    /* activation.py:21  */
    SLK_ACT_EXP = 7,         /* activation.py:45  */
    SLK_ACT_ERF = 8,         /* activation.py:60  */
    SLK_ACT_L1ML2 = 9,       /* activation.py:64  */
    SLK_ACT_FAIR = 10,       /* activation.py:68  */
    SLK_ACT_RETU = 11,       /* activation.py:72  */
    SLK_ACT_TANH_PM = 12,    /* activation.py:81  */
    SLK_ACT_SIGMOID_PM = 13, /* activation.py:88  */
    SLK_ACT_BOUNDED_LINEAR = 14, /* activation.py:95 */
    SLK_ACT_SIN = 15,        /* activation.py:102 */
    SLK_ACT_CAUCHY = 16,     /* activation.py:106 */
    SLK_ACT_GEMAN_MCCLURE = 17, /* activation.py:110 */
    SLK_ACT_WELSH = 18,      /* activation.py:114 */
    SLK_ACT_COUNT = 19
};

typedef void *slk_stream_t; /* hipStream_t */

SLK_API int slk_abi_version(void);
SLK_API const char *slk_error_string(int code);
/* Number of HIP devices visible (host call; does not create a context when 0).                               */
SLK_API int slk_device_count(void);
/* Device self-test: one v_mfma_f32_4x4x1_16b_f32 on 64 lanes, a[lane], b[lane] -> d[4][64], with the CBSZ / ABID broadcast
 * modifiers the exact-fp32 recurrent kernels rely on (cbsz, abid in {(0,0), (4,0), (4,3), (4,15), (3,0), (3,5), (2,1), (2,3)}):
 * lets an integrator confirm on the installed device the operand layout those kernels assume.                              */
/* Measurement aid: the shader clock the device holds at the moment the probe runs.  One wave reads the shader-cycle counter and
 * the constant 100 MHz counter around `spins` x 127 sleep quanta; out2[0] / out2[1] x 100 = MHz (device memory, two uint64).
 * Launched on a stream of its own beside a running workload it reports the clock under that load (bench.py `sustained`).     */
SLK_API int slk_clock_probe(unsigned long long *out2, int spins, slk_stream_t stream);
SLK_API int slk_selftest_mfma4_f32(const float *a, const float *b, float *d, int cbsz, int abid, slk_stream_t stream);

/* Elementwise activation y[i] = act(x[i]) (sloika/activation.py); in place allowed (y == x).                 */
SLK_API int slk_activation_f32(const float *x, float *y, size_t count, int act, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a1. Chunk front end: per-chunk median/MAD normalisation.
 * Replaces: sloika/tools/chunkify_raw.py:172-181 ("per-chunk" branch of raw_chunkify), the same maths as
 * sloika/basecall.py:117-118 and sloika/maths.py:4-27 (factor 1.4826, numpy median semantics).
 *   signal : [nchunk][chunk_len] float32 (chunk-major, as raw_chunkify's reshape produces it)
 *   out    : element (c, i) is written at out[c*out_chunk_stride + i*out_sample_stride]
 *            (out_chunk_stride=chunk_len, out_sample_stride=1 keeps chunk-major;
 *             out_chunk_stride=1, out_sample_stride=nchunk writes the [T][B][1] network layout of
 *             bin/train_network.py:304 directly)
 *   med_out, mad_out : optional [nchunk] (NULL to skip); mad includes the 1.4826 factor.
 * Results are bit-identical to numpy's float32 evaluation.  Any chunk_len (LDS sort up to 32768 samples, exact radix
 * selection beyond: whole reads).  -0.0 and +0.0 are distinct keys in the selection; NaNs are not supported.
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_med_mad_normalise_f32(const float *signal, int nchunk, int chunk_len, float *out,
                              long out_chunk_stride, long out_sample_stride, float *med_out, float *mad_out,
                              slk_stream_t stream);
/* The same per READ for a batch of whole reads of different lengths (sloika/basecall.py:117-118 normalises a read over its own
 * length): read r is lens[r] samples at signal + r*in_stride; element (r, i), i < lens[r], goes to
 * out[r*out_chunk_stride + i*out_sample_stride]; nothing else is written (the caller zero-fills a padded batch).  Exact order
 * statistics by radix selection, any length >= 1; bit-identical to slk_med_mad_normalise_f32 on each read alone.            */
SLK_API int slk_med_mad_normalise_ragged_f32(const float *signal, int nread, long in_stride, const int32_t *lens, float *out,
                                             long out_chunk_stride, long out_sample_stride, float *med_out, float *mad_out,
                                             slk_stream_t stream);
/* Whole-read mode (sloika/basecall.py:88-121 calls reads one at a time; the build batches reads of similar length): the read set lies
 * in one device buffer, read r in src[start[r] .. start[r] + len[r]).
 *   slk_pack_reads_f32       dst:[nread][ld] (ld >= max len) <- the reads, zero-padded to ld samples each
 *   slk_reads_nonfinite_f32  flags[r] |= 1 when read r holds a NaN or an infinity (flags zeroed by the caller; max_len >= max len):
 *                            such a read is skipped and reported, as raw_worker skips a read that fails (basecall.py:103-115),
 *                            instead of poisoning the batch it would have shared.                                               */
SLK_API int slk_pack_reads_f32(const float *src, const int64_t *start, const int32_t *len, int nread, float *dst, long ld,
                       slk_stream_t stream);
SLK_API int slk_reads_nonfinite_f32(const float *src, const int64_t *start, const int32_t *len, int nread, int max_len, int32_t *flags,
                            slk_stream_t stream);
/* batch.trim_open_pore(signal, max_op_fraction=0) + util.trim_array for every read of an uploaded set, without a round trip to the host
 * (sloika/batch.py:194-220 with the CLI's default fraction, bin/basecall_network.py:71: np.percentile(., 0) is the minimum; then
 * sloika/basecall.py:111-112).  spread: the per-window spreads of ALL reads (MAD or std of every `window` samples, as
 * slk_med_mad_normalise_f32 / slk_window_std_f32 leave them); read r owns windows first_win[r] .. first_win[r] + nwin[r] (whole windows
 * only, as the reference's reshape) and starts at sample first_sample[r] of the set.  Out: start[r] (sample index in the set) and len[r]
 * of the trimmed read.  flags[r]: bit 0 set by the caller = the read holds a sample that is not finite; on return bit 1 = the reference's
 * function fails on the read (no whole window, or no window livelier than the minimum), bit 2 = nothing left after trimming; a read with
 * any bit set gets len 0 (the reference's worker reports it and goes on, sloika/basecall.py:103-115).                              */
SLK_API int slk_open_pore_trim_f32(const float *spread, const int64_t *first_win, const int32_t *nwin, const int64_t *first_sample,
                           int nread, int window, int trim0, int trim1, int64_t *start, int32_t *len, int32_t *flags,
                           slk_stream_t stream);
/* Standard deviation of each of nwin consecutive windows of `win` samples (population form, numpy's .std()):
 * batch.trim_open_pore(var_method='std'), sloika/batch.py:210-211.  out:[nwin].                                    */
SLK_API int slk_window_std_f32(const float *signal, int nwin, int win, float *out, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a3. Convolution.run  (sloika/layers.py:417-419 -> sloika/conv.py:66-77,90-111)
 *   y[to][b][o] = act( sum_{c,k} xpad[to*stride + k][b][c] * W[o][c][k] + bias[o] ), zero padding
 *   (pad_l, pad_r) on the time axis, cross-correlation (filter_flip=False).
 *   x : element (t, b, c) at x[t*x_t_stride + b*x_b_stride + c]   (x_t_stride=B*Cin, x_b_stride=Cin for the
 *       reference layout; x_t_stride=1, x_b_stride=T reads chunk-major signal when Cin == 1)
 *   W : [Cout][Cin][winlen]   bias: [Cout] or NULL   y : [Tout][B][Cout], Tout = slk_conv1d_out_len(...)
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_conv1d_out_len(int T, int winlen, int stride, int pad_l, int pad_r);
SLK_API int slk_conv1d_f32(const float *x, long x_t_stride, long x_b_stride, const float *W, const float *bias, float *y,
                   int T, int B, int Cin, int Cout, int winlen, int stride, int pad_l, int pad_r, int act,
                   slk_stream_t stream);

/* a3b. Window.run (sloika/layers.py:346-351): zero pad w/2 both ends, y[t][b][k*F+f] = xpad[t+k][b][f].      */
SLK_API int slk_window_f32(const float *x, float *y, int T, int B, int F, int w, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a6. FeedForward.run (sloika/layers.py:157-158) and every other `tensordot(x, W, axes=(2,1)) + b`:
 *   y[r][j] = act( sum_k x[r][k] * W[j][k] + bias[j] ),  r < M (= T*B rows), j < N, k < K.
 *   x rows are ldx floats apart, y rows ldy floats apart (lets Parallel/birnn outputs be written straight
 *   into their slice of the concatenated tensor, sloika/layers.py:1486-1487).  W:[N][K] dense.  fp32 MFMA.
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_gemm_bias_act_f32(const float *x, long ldx, const float *W, const float *bias, float *y, long ldy,
                          long M, int K, int N, int act, slk_stream_t stream);

/* Softmax.run (sloika/layers.py:309-314): logits = x.W^T + b; p = exp(l - max) / sum.  y:[M][N] dense.      */
SLK_API int slk_linear_softmax_f32(const float *x, long ldx, const float *W, const float *bias, float *y, long M, int K,
                           int N, slk_stream_t stream);
/* x-stationary variant for wide outputs (csrc/gemm_rows.hip): logits y[r][j] = x[r].W[j] + b[j] with row stride ldy,
 * plus (when stats != NULL) stats[r] = (max_j, 1 / sum_j exp(y[r][j] - max_j)) accumulated on the fly.  K <= 128, else
 * SLK_ERR_UNSUPPORTED.  slk_softmax_from_stats_f32 turns (logits, stats) into the posterior exp(l - max) * inv_sum.    */
SLK_API int slk_linear_rowstats_f32(const float *x, long ldx, const float *W, const float *bias, float *y, long ldy, long M, int K,
                            int N, float *stats /* [M][2] or NULL */, slk_stream_t stream);
SLK_API int slk_softmax_from_stats_f32(const float *logits, long ld_in, const float *stats, float *post, long ld_out, long M,
                               int N, slk_stream_t stream);
/* Same contraction on the FP16 matrix pipe with float32-grade accuracy (csrc/gemm_rows_f16x3.hip): operands split
 * v = hi + lo in fp16, x.w ~= x_hi.w_hi + x_hi.w_lo + x_lo.w_hi accumulated in float32 -- ~5x the fp32-MFMA
 * throughput, error a few float32 ulps.  Weights are split once with slk_split_f16x2_f32 into two fp16 matrices
 * [N][KP], KP = K rounded up to 16 (2*N*KP bytes each), every row scaled by a power of two that brings its largest
 * magnitude into [1, 2) (inv_scale[N] receives the inverse scales); the kernels scale every row of x the same way and undo
 * both on the float32 accumulators, so operands of ANY finite float32 magnitude are handled (fp16 alone overflows at 65504
 * and loses its lo half below 6e-5).                                                                                 */
SLK_API int slk_split_f16x2_f32(const float *w, int rows, int K, void *hi, void *lo, float *inv_scale, slk_stream_t stream);
SLK_API int slk_linear_rowstats_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *W_inv_scale,
                              const float *bias, float *y, long ldy, long M, int K, int N,
                              float *stats /* [M][2] or NULL */, slk_stream_t stream);
/* FeedForward.run (sloika/layers.py:157-158) on the same kernel: y = act(x.W^T + b); act one of linear / tanh / sigmoid /
 * relu / elu, otherwise SLK_ERR_UNSUPPORTED (use slk_gemm_bias_act_f32).  Both: K <= 192, N <= 2048.                  */
SLK_API int slk_gemm_bias_act_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *W_inv_scale,
                            const float *bias, float *y, long ldy, long M, int K, int N, int act, slk_stream_t stream);
/* The same product for rows longer than that kernel takes (csrc/gemm_bf16x6.hip: any K that is a multiple of 4; the dL/dx
 * products of the training step -- th.grad through T.dot in layers.py:157-158, :310-313, :1010-1021 -- whose K = 3n / 4n / nstate
 * exceeds 192): every float32 operand as three bf16 pieces, six bf16 MFMA terms per product in float32 accumulators (float32-grade,
 * float32's exponent range: gradients need no scaling).  W:[N][K] is cut once by slk_pack_bf16x3_f32 into
 * slk_pack_bf16x3_bytes(N, K) bytes.  act: linear / tanh / sigmoid; bias may be NULL.  SLK_ERR_UNSUPPORTED (->
 * slk_gemm_bias_act_f32) for other activations, K or ldx not multiples of 4, x or packed not 16-byte aligned. */
SLK_API size_t slk_pack_bf16x3_bytes(int N, int K);
SLK_API int slk_pack_bf16x3_f32(const float *W, int N, int K, void *packed, slk_stream_t stream);
SLK_API int slk_gemm_bias_act_bf16x6(const float *x, long ldx, const void *packed, const float *bias, float *y, long ldy, long M,
                             int K, int N, int act, slk_stream_t stream);
/* out = (x . W^T) * fun'(.), fun' in terms of the OUTPUT yref:[M][N] (rows ldyref apart) of the activation `dact` of the layer below
 * (tanh, sigmoid, relu, elu, linear): the dL/dx product of a layer and slk_act_backward_f32 of the layer below it in one pass.
 * Same shapes and packed weights as slk_gemm_bias_act_bf16x6; SLK_ERR_UNSUPPORTED likewise (-> the two calls). */
SLK_API int slk_gemm_dact_bf16x6(const float *x, long ldx, const void *packed, const float *yref, long ldyref, int dact, float *out,
                         long ldo, long M, int K, int N, slk_stream_t stream);
/* In-place row softmax of y:[M][N] (the second half of the above, exposed for testing).                     */
SLK_API int slk_softmax_rows_f32(float *y, long M, int N, slk_stream_t stream);
/* Row statistics only: stats[r] = (max_j logits[r][j], 1 / sum_j exp(logits[r][j] - max)); the posterior
 * exp(l - max) * inv_sum rebuilt from them is bit-identical to what slk_softmax_rows_f32 writes.  Lets the decoder
 * consume logits directly so the normalised posterior is never written (slk_viterbi_kmer_logits_f32).            */
SLK_API int slk_softmax_rowstats_f32(const float *logits, long M, int N, float *stats /* [M][2] */, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a4. Gru  (sloika/layers.py:952-1021; step :1010-1021; zero initial state :85-88; Reverse :1449-1450)
 *   slk_gru_recurrent_f32 consumes the time-parallel input projection
 *       vI[t][b][g*n + j] = x_t . iW[g*n+j] + b[g*n+j]   (g = 0 z, 1 r, 2 candidate; gate-block layout)
 *   computed by slk_gemm_bias_act_f32, and runs the sequential part
 *       z = gate(vI_z + h.sW_z^T)  r = gate(vI_r + h.sW_r^T)
 *       hbar = act(vI_c + (r*h).sW2^T)   h = z*h + (1-z)*hbar
 *   over t = 0..T-1 (reverse=0) or t = T-1..0 (reverse=1, implements Reverse(Gru) without copies).
 *   sW:[2n][n]  sW2:[n][n]  (reference shapes, [out][in]);  h_out element (t,b,j) at
 *   h_out[(t*B + b)*ldh + j]  (ldh >= n).
 * slk_gru_f32 = projection + recurrence; workspace >= slk_gru_workspace_bytes(T,B,n).
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_gru_recurrent_f32(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T, int B,
                          int n, int reverse, int act, int gate_act, slk_stream_t stream);
SLK_API size_t slk_gru_workspace_bytes(int T, int B, int n);
SLK_API int slk_gru_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias,
                float *y, long ldy, int T, int B, int insize, int n, int reverse, int act, int gate_act,
                void *workspace, size_t workspace_bytes, slk_stream_t stream);
/* The whole layer with the RECURRENCE on the fp16 matrix pipe as well (csrc/gru_bar16.hip): every float32 operand of
 * h.sW^T and (r*h).sW2^T is split v = hi + lo into two fp16 halves and each product evaluated in float32 accumulators
 * (v_mfma_f32_16x16x32_f16) -- 22 significand bits per operand.  |h| <= 1 by construction; every row of x and of the
 * three weight matrices is scaled by a power of two to a maximum in [1, 2) before its split and the accumulators are
 * scaled back, so inputs and weights of any finite float32 magnitude are safe.  lens: NULL, or ragged lengths as
 * slk_gru_recurrent_ragged_f32; zr_out: NULL, or [T*B][2n] = [z | r] of every step (the training forward pass).
 * Execution plan: four waves per workgroup, one per SIMD with 512 registers each, stepping in lock step through two
 * s_barrier per time step; projection weights live in accumulation registers.  This is the plan sloika_amd.layers.Gru runs;
 * SLK_ERR_UNSUPPORTED for shapes without an instantiation ((insize, n) in {(96,96), (64,64), (32,96), (128,96), (64,96),
 * (48,32), (16,64)}, tanh / sigmoid).
 * A workgroup takes four chunks; a batch with more such workgroups than the device has CUs runs the eight-chunk plan of
 * csrc/gru_bar16d.hip instead (two four-chunk tiles through the same MFMAs), one with more eight-chunk workgroups than CUs
 * the sixteen-chunk plan of csrc/gru_bar16q.hip (four and eight chunks: bit-identical results; sixteen: to float32 rounding).
 * Bits 8-9 of `reverse` force a plan: 0 = by batch size, 1 / 2 / 3 = four / eight / sixteen chunks per workgroup.  Bit 10: the
 * four-chunk workgroups of a layer up to 64 wide may share a CU (no exclusive LDS request): for callers that run two such
 * launches side by side -- the directions of a birnn -- with more workgroups in all than the device has CUs.            */
SLK_API int slk_gru_bar16_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2, const float *bias,
                      float *y, long ldy, int T, int B, int insize, int n, int reverse, int act, int gate_act,
                      const int32_t *lens, float *zr_out, slk_stream_t stream);
/* Ragged batches (whole reads of different lengths, zero-padded to T steps; the reference calls reads one at a time,
 * sloika/basecall.py:88-121): lens[b] in [1, T] (int32, device) is the number of valid steps of chunk b.  Steps
 * t >= lens[b] of y / h_out are left untouched, and with reverse = 1 the scan of chunk b starts at ITS last step,
 * i.e. each chunk gets exactly what a call on the unpadded chunk alone would produce.                                */
SLK_API int slk_gru_recurrent_ragged_f32(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T, int B,
                                 int n, int reverse, int act, int gate_act, const int32_t *lens, slk_stream_t stream);
/* The same scan (Gru.step over a projection vI [T*B][ldv >= 3n] = x.iW^T + b in HBM; sloika/layers.py:1010-1021) for layers
 * too wide for the fused kernels, on the execution plan of slk_gru_bar16_f32: recurrent products as 3-term fp16 splits, four
 * waves stepping through two barriers per step (csrc/gru_scan16.hip).  n = 112 or 128 (models/pretrained.pkl,
 * models/raw_1.00_rGr.py zero-padded), tanh / sigmoid; anything else SLK_ERR_UNSUPPORTED (-> slk_gru_recurrent_f32).
 * lens (int32 [B], device) or NULL as for slk_gru_recurrent_ragged_f32.                                                  */
SLK_API int slk_gru_scan16_f32(const float *vI, long ldv, const float *sW, const float *sW2, float *y, long ldy, int T, int B, int n,
                       int reverse, int act, int gate_act, const int32_t *lens, slk_stream_t stream);
/* Force the portable (non-MFMA) recurrence kernel: 0 = auto, 1 = force generic.  Testing aid; passed per call
 * through the `_ex` form so that there is still no global state.                                            */
SLK_API int slk_gru_recurrent_f32_ex(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T,
                             int B, int n, int reverse, int act, int gate_act, int force_generic,
                             slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a6b. Lstm (sloika/layers.py:599-697; step :677-691 is authoritative: INTERLEAVED gate layout,
 *   row = j*4 + g with g = 0 update, 1 input gate, 2 forget gate, 3 output gate; peepholes p:[3][n]).
 *   vW[t][b][j*4+g] = x_t . iW[j*4+g] + b[j*4+g]  (from slk_gemm_bias_act_f32);  sW:[4n][n];
 *   p may be NULL (has_peep=False).  Output = the `out` half of the state (layers.py:697).
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_lstm_recurrent_f32(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B,
                           int n, int reverse, int act, int gate_act, slk_stream_t stream);
SLK_API int slk_lstm_recurrent_ragged_f32(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B,
                                  int n, int reverse, int act, int gate_act, const int32_t *lens /* see slk_gru_recurrent_ragged_f32 */,
                                  slk_stream_t stream);
/* The same scan with the recurrent product as a 3-term fp16 split on the barrier-stepped plan (csrc/lstm_scan16.hip; n a multiple
 * of 16 up to 128, tanh / sigmoid, vW 16-byte aligned and < 4 GiB; SLK_ERR_UNSUPPORTED otherwise -> slk_lstm_recurrent_f32).
 * lens may be NULL (all chunks T steps long). */
SLK_API int slk_lstm_scan16_f32(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B, int n,
                        int reverse, int act, int gate_act, const int32_t *lens, slk_stream_t stream);
/* A whole Lstm layer of up to 64 units and up to 64 inputs in one kernel (csrc/lstm_fused16.hip): the scan of slk_lstm_scan16_f32 with
 * the projection vW = x.iW^T + b computed inside, four steps at a time, from x:[T][B][insize] (rows ldx floats apart) -- vW is never
 * written.  insize a multiple of 4 up to 64, n a multiple of 16 up to 64, tanh / sigmoid, x 16-byte aligned, ldx a multiple of 4;
 * SLK_ERR_UNSUPPORTED otherwise (-> projection GEMM + slk_lstm_scan16_f32).  bias, p and lens may be NULL. */
SLK_API int slk_lstm_fused16_f32(const float *x, long ldx, const float *iW, const float *sW, const float *bias, const float *p, float *y,
                         long ldy, int T, int B, int insize, int n, int reverse, int act, int gate_act, const int32_t *lens,
                         slk_stream_t stream);
SLK_API size_t slk_lstm_workspace_bytes(int T, int B, int n);
SLK_API int slk_lstm_f32(const float *x, long ldx, const float *iW, const float *sW, const float *bias, const float *p,
                 float *y, long ldy, int T, int B, int insize, int n, int reverse, int act, int gate_act,
                 void *workspace, size_t workspace_bytes, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a7 + a8. decode.prepare_post (sloika/decode.py:21-36) and decode.viterbi (sloika/decode.py:39-93),
 * batched over chunks (the reference decodes one [T,1,S] posterior per call, sloika/basecall.py:26-51).
 *   post : [T][B][nstate] float32, nstate = nbase^klen + 1, state 0 = blank/stay
 *   input_mode: SLK_POST_RAW   post is a network posterior: lp = log(min_prob + (1-min_prob)*post + 1e-10)
 *               SLK_POST_PLAIN lp = log(post + 1e-10)                        (decode.viterbi(log=False))
 *               SLK_POST_LOG   lp = post                                      (decode.viterbi(log=True))
 *   skip_pen  : decode.viterbi's skip_pen (float32 arithmetic as numpy does for float32 input)
 *   score_out : [B] best path score (float32)
 *   path_out  : [B][T] int32 k-mer states in time order, left aligned; entries >= len are -1
 *   len_out   : [B] path lengths (1 <= len <= T)
 * Tie-breaking is exactly the reference's: first maximum for step/skip/argmax (np.argmax), step vs skip tie
 * -> skip (decode.py:76), move vs stay tie -> stay (decode.py:81).  Integer outputs are bit-exact given the
 * same log-posteriors.  klen >= 3 (decode.py:50), nbase in {4,5} (any nbase with nbase^2 <= 64 works).
 * slk_log_post_f32 exposes the exact log-posterior transform the decoder applies (for tests).
 * ------------------------------------------------------------------------------------------------------- */
#define SLK_POST_RAW 0
#define SLK_POST_PLAIN 1
#define SLK_POST_LOG 2
#define SLK_POST_LN 3 /* slk_log_post_f32 only: lp = log(post), no eta (np.log(trans), sloika/transducer.py:30) */
SLK_API size_t slk_viterbi_kmer_workspace_bytes(int T, int B, int nbase, int klen);
SLK_API int slk_viterbi_kmer_f32(const float *post, int T, int B, int nbase, int klen, float skip_pen, int input_mode,
                         float min_prob, void *workspace, size_t workspace_bytes, float *score_out,
                         int32_t *path_out, int32_t *len_out, slk_stream_t stream);
SLK_API int slk_log_post_f32(const float *post, float *lpost, size_t count, int input_mode, float min_prob,
                     slk_stream_t stream);
/* Softmax.run + decode_post in one pass over the LOGITS (sloika/layers.py:309-314 + sloika/basecall.py:26-51):
 * logits: rows (t,b) of nstate floats, `ld` floats apart, = x.W^T + b;  stats:[T*B][2] from slk_linear_rowstats_f32
 * (or slk_softmax_rowstats_f32 for dense logits).  Same outputs as
 * slk_viterbi_kmer_f32(SLK_POST_RAW) on the normalised posterior, bit for bit.  slk_log_post_logits_f32 exposes the
 * log-posterior it decodes (for tests).                                                                          */
SLK_API int slk_viterbi_kmer_logits_f32(const float *logits, long ld /* floats between rows, >= nstate */, const float *stats,
                                int T, int B, int nbase, int klen, float skip_pen, float min_prob, void *workspace,
                                size_t workspace_bytes, float *score_out, int32_t *path_out, int32_t *len_out,
                                slk_stream_t stream);
/* Ragged form (see slk_gru_recurrent_ragged_f32): chunk b is decoded over its first lens[b] steps only; path_out rows keep
 * the padded length T.                                                                                               */
SLK_API int slk_viterbi_kmer_logits_ragged_f32(const float *logits, long ld, const float *stats, int T, int B, int nbase, int klen,
                                       float skip_pen, float min_prob, const int32_t *lens, void *workspace,
                                       size_t workspace_bytes, float *score_out, int32_t *path_out, int32_t *len_out,
                                       slk_stream_t stream);
SLK_API int slk_log_post_logits_f32(const float *logits, long ld, const float *stats, float *lpost /* dense [rows][nstate] */,
                            size_t rows, int nstate, float min_prob, slk_stream_t stream);
/* The same chain from the Softmax layer's INPUT: x.W^T + b, softmax, prepare_post, log and the Viterbi forward pass in one
 * kernel, so that the [T][B][nstate] logits never exist in memory (sloika/layers.py:309-314 -> sloika/decode.py:21-36 ->
 * :39-93; what basecall.decode_post(calc_post(x)) computes from the last hidden layer on, sloika/basecall.py:26-51, 119).
 *   x    : rows (t, b) of K floats, ldx floats apart (ldx % 4 == 0, 16-byte aligned), row index t*B + b
 *   pack : the layer's weights W[nstate][K], b[nstate] (or NULL) re-laid-out once by slk_softmax_viterbi_pack_f32 into a
 *          buffer of slk_softmax_viterbi_pack_bytes(K, nbase, klen) bytes (fp16 hi/lo MFMA fragments, column scales)
 *   lens : per-chunk step counts for a ragged batch, or NULL
 *   plan : chunks per workgroup: 0 (the build's default) or 2; 4 (two score chains per lane, half as many workgroups) is
 *          compiled only with -DSV_WITH_NCH4 (measured no faster) and returns SLK_ERR_UNSUPPORTED otherwise; results do
 *          not depend on it
 *   lp_dump : NULL, or [T][B][nstate] floats that receive the log-posteriors the dynamic programme consumed (tests decode
 *          THESE with the oracle: paths and scores are bit-exact functions of them)
 *   workspace: slk_softmax_viterbi_workspace_bytes(T, B, nbase, klen) bytes, 16-byte aligned (one traceback byte per four
 *          k-mers and step: a quarter of slk_viterbi_kmer_workspace_bytes, which is also accepted)
 * Products are 3-term fp16 splits with float32 accumulation like slk_linear_rowstats_f16x3.  This build has the kernel for
 * nbase 4, klen 5 and K in {64, 96, 112, 128}; anything else returns SLK_ERR_UNSUPPORTED (pack_bytes returns 0) and the
 * caller uses slk_linear_rowstats_* + slk_viterbi_kmer_logits_f32.                                                        */
SLK_API size_t slk_softmax_viterbi_pack_bytes(int K, int nbase, int klen);
SLK_API size_t slk_softmax_viterbi_workspace_bytes(int T, int B, int nbase, int klen);   /* 0: shape outside the fused kernel */
SLK_API int slk_softmax_viterbi_pack_f32(const float *W, const float *bias, int K, int nbase, int klen, void *pack,
                                 slk_stream_t stream);
SLK_API int slk_softmax_viterbi_f32(const float *x, long ldx, const void *pack, int K, int T, int B, int nbase, int klen,
                            float skip_pen, float min_prob, const int32_t *lens, int plan, void *workspace,
                            size_t workspace_bytes, float *score_out, int32_t *path_out, int32_t *len_out, float *lp_dump,
                            slk_stream_t stream);
/* decode.prepare_post on its own: out = min_prob + (1-min_prob)*post (decode.py:36).                        */
SLK_API int slk_prepare_post_f32(const float *post, float *out, size_t count, float min_prob, slk_stream_t stream);
/* decode.argmax (decode.py:5-18), batched: per (b) the states with argmax != blank, minus 1 if
 * zero_is_blank; same output convention as slk_viterbi_kmer_f32.                                            */
SLK_API int slk_argmax_decode_f32(const float *post, int T, int B, int nstate, int zero_is_blank, int32_t *path_out,
                          int32_t *len_out, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * f1. States -> bases (sloika/bio.py:160-179 max_overlap, :206-225 reduce_kmers, :228-237 kmers_to_sequence; what
 *   basecall.SeqPrinter.write does with a call, basecall.py:157-163), for a whole batch of decoded paths on the device.
 *   A state is the base-`nbase` number of its k-mer (first letter most significant, bio.py:12-24), so the overlap test
 *   k1[i:] == k2[:-i] is  s1 mod nbase^(k-i) == s2 div nbase^i; the smallest such i is the move (k if none, 0 for a
 *   repeated state unless always_move).  paths:[B][ld] int32, lens:[B]; alphabet: the letters packed little-endian in 8
 *   bytes; out:[B][cap] bytes with cap >= klen * max(lens); nbases[b] = letters written for read b.
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_paths_to_bases(const int32_t *paths, long ld, const int32_t *lens, int B, int klen, int nbase, int always_move,
                       unsigned long long alphabet, uint8_t *out, long cap, int32_t *nbases, slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * a9. The reference's only true FFI: viterbi_helpers.slip_update (sloika/viterbi_helpers.pyx:12-35), and
 * its caller transducer.map_to_sequence (sloika/transducer.py:14-73).
 *   slk_slip_update_f32: x:[n] -> from_score:[n] float32, from_pos:[n] int64; n >= 3.
 *   slk_map_to_sequence_f32: ltrans:[nev][nst] LOG-space float32, seq:[npos] int32 state indices,
 *     prior_initial / prior_final: [npos] float64 or NULL (the reference adds float64 priors into the
 *     float32 score vector, transducer.py:39-41,63-64); score_out:[1]; path_out:[nev] int32.
 *     workspace >= slk_map_to_sequence_workspace_bytes(nev, npos).  npos >= 3.
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_slip_update_f32(const float *x, int n, float slip, float *from_score, int64_t *from_pos,
                        slk_stream_t stream);
SLK_API size_t slk_map_to_sequence_workspace_bytes(int nev, int npos);
SLK_API int slk_map_to_sequence_f32(const float *ltrans, int nev, int nst, const int32_t *seq, int npos, float slip,
                            const double *prior_initial, const double *prior_final, void *workspace,
                            size_t workspace_bytes, float *score_out, int32_t *path_out, slk_stream_t stream);
/* Batched form of the same call (the reference remaps reads one by one: bin/chunkify.py `remap` ->
 * transducer.map_to_sequence, sloika/transducer.py:14-73): `nread` reads of different lengths, concatenated.
 *   ev_off:[nread+1] int64 -- read b owns rows ev_off[b]..ev_off[b+1] of ltrans ([sum nev][nst]) and of path_out;
 *   pos_off:[nread+1] int64 -- ... and positions pos_off[b]..pos_off[b+1] of seq / prior_initial / prior_final;
 *   ws_off:[nread] int64 -- offset (in int32 elements) of read b's nev_b*npos_b traceback inside `workspace`;
 *   max_npos = the longest sequence (sizes the LDS request: 28 bytes per position of the 160 KB, npos <= 5846); score_out:[nread].
 * A read with fewer than 3 positions or no events gets score -inf and its path is left untouched. */
SLK_API int slk_map_to_sequence_batch_f32(const float *ltrans, int nst, const int64_t *ev_off, const int32_t *seq,
                                  const int64_t *pos_off, int nread, int max_npos, float slip,
                                  const double *prior_initial, const double *prior_final, void *workspace,
                                  const int64_t *ws_off, float *score_out, int32_t *path_out, slk_stream_t stream);

/* f3, second half: the labels of raw_chunkify (sloika/tools/chunkify_raw.py:164-210), from the mapping table raw_remap
 * (chunkify_raw.py:260-296) produces.  Integer work, bit-exact.  `alphabet` is a HOST string of `nbase` <= 8 letters
 * (batch.init_chunk_identity_worker's alphabet, sloika/batch.py:17-27); every other pointer is device memory.
 * `status` (device int, zeroed by the caller) gets bit 0 set for a letter outside the alphabet and bit 1 for a reference
 * position outside the reference string -- the cases in which the reference's `batch.kmer_to_state[...]` raises KeyError.
 *
 * slk_kmer_labels_i32 = labels_from_mapping_table (chunkify_raw.py:117-136): kmers:[n][old_klen] bytes (the table's 'kmer'
 *   column), labels_out[i] = state of the middle `klen` letters + index_from, or -1 for a foreign letter. */
SLK_API int slk_kmer_labels_i32(const uint8_t *kmers, int64_t n, int old_klen, int klen, const char *alphabet, int nbase,
                                int index_from, int32_t *labels_out, int *status, slk_stream_t stream);
/* slk_raw_chunk_labels_i32 = the non-interpolated branch (chunkify_raw.py:194-204) for `nread` reads in one launch (the
 * reference runs one read per worker call, chunkify_raw.py:299-337).  The tables are the TRIMMED ones (after
 * trim_signal_and_mapping(signal, table, 0, nchunk*chunk_len), chunkify_raw.py:174), concatenated:
 *   start, move:[sum nevent] int64 columns; event_label:[sum nevent] = slk_kmer_labels_i32 of the 'kmer' column;
 *   ev_off:[nread+1]; nchunk:[nread] chunks of read b; lab_off:[nread] offset (in elements) of read b's
 *   [nchunk_b][ceil(chunk_len / downsample)] int32 labels inside labels_out; max_nchunk = max(nchunk).
 * workspace: slk_raw_chunk_labels_workspace_bytes(sum nevent) bytes. */
SLK_API size_t slk_raw_chunk_labels_workspace_bytes(int64_t nevent);
SLK_API int slk_raw_chunk_labels_i32(const int64_t *start, const int64_t *move, const int32_t *event_label,
                                     const int64_t *ev_off, int nread, const int64_t *nchunk, const int64_t *lab_off,
                                     int64_t max_nchunk, int chunk_len, int downsample, void *workspace,
                                     size_t workspace_bytes, int32_t *labels_out, slk_stream_t stream);
/* slk_raw_chunk_labels_interp_i32 = the interpolated branch (chunkify_raw.py:187-193 with interpolate_pos /
 * interpolate_labels :86-114) for one read: label o belongs to sample o * downsample; position = np.interp over the block
 * mid-times in float64, evaluated like numpy's C loop, then np.around(. - 0.5 * klen + 1e-10).
 *   forward != 0: direction '+', ref_anchor = ref_start;  forward == 0: direction '-', ref_anchor = ref_stop.
 *   map_klen = length of the table's k-mers; reference:[ref_len] bytes; pos_out (optional):[nlabel] int64 positions.
 *   times (optional):[nlabel] float64 -- the caller's own times (the closures interpolate_pos / interpolate_labels return
 *   take any), instead of o * downsample;  zero_repeats != 0 applies line :191 (raw_chunkify), 0 leaves every label.
 * labels_out[o] = state + 1 of reference[pos:pos+klen], 0 where pos repeats the previous label's, -1 on a status error;
 * labels_out may be NULL when only positions are wanted (then reference / alphabet are not read). */
SLK_API int slk_raw_chunk_labels_interp_i32(const int64_t *start, const int64_t *length, const int64_t *seq_pos,
                                            int64_t nevent, int map_klen, int forward, int64_t ref_anchor,
                                            const uint8_t *reference, int64_t ref_len, int klen, const char *alphabet,
                                            int nbase, int64_t nlabel, int downsample, const double *times,
                                            int zero_repeats, int32_t *labels_out, int64_t *pos_out, int *status,
                                            slk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * f2. The training step: bin/train_network.py:124-142 (`wrap_network`: loss, accuracy, th.grad, updates.adam) and
 * sloika/updates.py:9-103.  The reference crosses into Theano once per batch (`fg(indata, labels, weights, rate)`,
 * train_network.py:308); its gradient comes from automatic differentiation, here the reverse pass of Convolution
 * (layers.py:417-419), Gru.step (layers.py:1010-1021) and Softmax (layers.py:309-314) is explicit (csrc/train.hip).
 * Rows are m = t*B + b throughout.
 *
 * Gate recompute (forward values the reverse scan needs, recomputed time-parallel from the finished forward pass):
 *   slk_train_pack_xh_f32:  xh[m] = [x[m] (I) | h_prev[m] (N)], h_prev = h at the previous scan step, 0 at the scan start
 *   slk_train_pack_xrh_f32: xrh[m] = [x[m] | r[m] * h_prev[m]] with zr[m] = [z | r] ([M][2N]) (kept for callers that
 *     want the candidate as a GEMM too: c = tanh(xrh . [iW[2n:] | sW2]^T + b[2n:]))
 *   then zr = sigmoid(xh . [iW[:2n] | sW]^T + b[:2n]) is a plain slk_gemm_bias_act_f16x3 / slk_gemm_bias_act_f32 call.
 * slk_gru_backward_f32: the reverse scan.  dy:[T][B] rows lddy apart = dL/dh from the layer above; hprev: h at the
 *   previous scan step per row (rows ldhp apart: the h half of the packed xh rows, or a view of the layer output shifted
 *   by one step over a zero row); zr:[M][2n] the activated gates (recomputed as above, or saved by
 *   slk_gru_bar16_f32 with zr_out); h: the layer's own forward output (rows ldh apart), from which the candidate of every step is recovered as (h_t - z h_prev) / (1 - z);
 *   writes da:[M][3n] = dL/dvI = [daz | dar | dac] and rh:[M][n] = r * h_prev.  n in {16,32,48,64,96,112,128,144},
 *   tanh / sigmoid, else SLK_ERR_UNSUPPORTED.
 *   Weight gradients follow as contractions over m (slk_gemm_tn_f32): diW = da^T x, dsW = da[:, :2n]^T h_prev,
 *   dsW2 = da[:, 2n:]^T rh, db = da^T 1 (its colsum output); and dL/dx = da . iW (slk_gemm_bias_act_f32 with iW^T).
 * slk_softmax_xent_grad_f32: loss terms and dL/dlogits of train_network.py:128-136, in place over the logits written by
 *   slk_linear_rowstats_* (stats:[M][2] = max, 1/sum).  loss_rows[m] and correct_rows[m] are already divided by the
 *   number of counted positions (T - 2 drop) * B, so their sums are the data term of the loss and the accuracy.
 *   Columns nstate..ld-1 of every row are set to zero.  labels must lie in [0, nstate).
 * slk_reduce_sum_f32: out[0] = sum x (square = 0) or sum x^2 (square = 1: updates.param_sqr), float64, fixed order.
 * slk_gemm_tn_f32: C[N1][N2] (rows ldc apart) = A^T B, A:[M][N1] rows lda apart, B:[M][N2] rows ldb apart; when colsum is
 *   given it also receives the column sums of A (A^T 1: the bias gradient) from the same pass.
 * slk_act_backward_f32: out = dy * fun'(pre-activation) written through the OUTPUT y; linear/tanh/sigmoid/relu/elu.
 * slk_train_im2col_cin1_f32: window rows of a one-feature Convolution, cols:[Tout*B][winlen] (dW = dpre^T cols).
 * slk_adamski_update_f32: updates.py:77-87 over flat buffers with this step's lr_t / momentum_decay (updates.py:73-76);
 *   g = clip(grad * gscale + 2 l2 param).   slk_sgd_update_f32: updates.py:9-33.
 * ------------------------------------------------------------------------------------------------------- */
SLK_API int slk_train_pack_xh_f32(const float *x, long ldx, const float *h, long ldh, float *xh, int T, int B, int insize, int n,
                          int reverse, slk_stream_t stream);
SLK_API int slk_train_pack_xrh_f32(const float *xh, const float *zr, float *xrh, long M, int insize, int n, slk_stream_t stream);
SLK_API int slk_gru_backward_f32(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h, long ldh,
                         const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n, int reverse,
                         int act, int gate_act, slk_stream_t stream);
/* The same reverse scan with its two products as fp16 splits on the barrier-stepped plan of the forward kernels (csrc/gru_bwd16.hip:
 * one wave per 16 units, two MFMAs per product, operand images scaled per chunk and step by a power of two from a bound -- gradients
 * have no natural range).  Same arguments; n a multiple of 16 up to 128, tanh / sigmoid, fewer than 4 GiB per operand;
 * SLK_ERR_UNSUPPORTED otherwise (-> slk_gru_backward_f32).  Results agree with it to float32 rounding of the products. */
SLK_API int slk_gru_backward16_f32(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h, long ldh,
                         const float *sW, const float *sW2, float *da, float *rh, int T, int B, int n, int reverse,
                         int act, int gate_act, slk_stream_t stream);
/* ... and with the layer's dL/dx = da . iW (updates.py:67: th.grad through the projection of layers.py:1011, iW:[3n][insize]) formed in
 * the same pass from the operand images of each step (they ARE da of that step as fp16 pairs): dx:[T][B] rows of insize floats, lddx
 * apart, float32-grade like the pass's own products.  n a multiple of 16 up to 96, insize <= 64 (n <= 64) or 96; SLK_ERR_UNSUPPORTED
 * otherwise (-> slk_gru_backward16_f32 + slk_gemm_bf16x6_f32 on da).  yref (or NULL): the OUTPUT of the element-wise activation `dact`
 * whose result is this layer's input, rows ldyref apart -- dx then leaves multiplied by fun'(.), i.e. as dL/d(pre-activation) of the
 * layer below (what slk_gemm_dact_bf16x6 does for the separate product). */
SLK_API int slk_gru_backward16_dx_f32(const float *dy, long lddy, const float *hprev, long ldhp, const float *zr, const float *h,
                              long ldh, const float *sW, const float *sW2, const float *iW, float *da, float *rh, float *dx,
                              long lddx, int T, int B, int n, int insize, int reverse, int act, int gate_act, const float *yref,
                              long ldyref, int dact, slk_stream_t stream);
/* Lstm (layers.py:677-697) in the reverse pass.  sum:[M][4n] = [x_t | out_{t-1}] . [iW | sW]^T + b (a GEMM over
 * slk_train_pack_xh_f32 rows), gate rows interleaved j*4 + gate as the reference stores them.
 *   slk_lstm_gates_f32: the element-wise cell recursion -> gates:[M][4n] = (candidate, input, forget, output) activated,
 *     cell:[M][n] = c_t.   peep:[3][n] or NULL.
 *   slk_lstm_backward_f32: the reverse scan -> dsum:[M][4n] = dL/dsum, dpeep:[B][3][n] per-chunk peephole gradients.
 *     n in {16,32,48,64,96,128}, tanh / sigmoid.  Then diW = dsum^T x, dsW = dsum^T out_prev, db = dsum^T 1, dx = dsum . iW. */
SLK_API int slk_lstm_gates_f32(const float *sum, const float *peep, float *gates, float *cell, int T, int B, int n, int reverse,
                       slk_stream_t stream);
SLK_API int slk_lstm_backward_f32(const float *dy, long lddy, const float *gates, const float *cell, const float *sW, const float *peep,
                          float *dsum, float *dpeep, int T, int B, int n, int reverse, int act, int gate_act,
                          slk_stream_t stream);
/* The same reverse scan on the barrier-stepped fp16-split plan of csrc/lstm_scan16.hip (csrc/lstm_bwd16.hip: n a multiple of 16 up to
 * 64, tanh / sigmoid, gates / dsum 16-byte aligned, operands < 4 GiB; SLK_ERR_UNSUPPORTED otherwise -> slk_lstm_backward_f32). */
SLK_API int slk_lstm_backward16_f32(const float *dy, long lddy, const float *gates, const float *cell, const float *sW, const float *peep,
                          float *dsum, float *dpeep, int T, int B, int n, int reverse, int act, int gate_act,
                          slk_stream_t stream);
SLK_API int slk_softmax_xent_grad_f32(float *logits, long ld, const float *stats, const int32_t *labels, const float *weights, int T,
                              int B, int nstate, int drop, float min_prob, float *loss_rows, float *correct_rows,
                              slk_stream_t stream);
/* The softmax layer of a training step WITHOUT a logits tensor: the layer's products (pre-split weights, as for
 * slk_linear_rowstats_f16x3) are computed twice -- a statistics pass that stores nothing but, per row, the loss term, the accuracy
 * term and four floats in xrow:[M][4], and a gradient pass that stores grad:[M][ld] = dL/dlogits (columns N..ld-1 zero) -- instead
 * of written, read, overwritten and read again.  Same values bit for bit as slk_linear_rowstats_f16x3 followed by
 * slk_softmax_xent_grad_f32.  M = T * B; K in 49..64 or 81..128, N <= 2048, ld <= N rounded up to 64, xrow 16-byte aligned;
 * SLK_ERR_UNSUPPORTED otherwise (-> the two calls above). */
SLK_API int slk_linear_xent_grad_f16x3(const float *x, long ldx, const void *W_hi, const void *W_lo, const float *W_inv_scale,
                               const float *bias, float *grad, long ld, int K, int N, const int32_t *labels, const float *weights,
                               int T, int B, int drop, float min_prob, float *loss_rows, float *correct_rows, float *xrow,
                               slk_stream_t stream);
SLK_API int slk_reduce_sum_f32(const float *x, size_t n, int square, double *out, slk_stream_t stream);
/* out[r] = sum of x[r][0..n) for r < nrow (<= 16) contiguous arrays, float64, fixed order, 256 workgroups per array (the loss and
 * accuracy terms of a step: two launches of ~5 us where two slk_reduce_sum_f32 take 2 x 74 us).  scratch: nrow * 256 doubles. */
SLK_API int slk_reduce_rows_sum_f32(const float *x, int nrow, size_t n, double *out, double *scratch, slk_stream_t stream);
SLK_API size_t slk_gemm_tn_workspace_bytes(long M, int N1, int N2);
/* 1..4 contractions over the same M rows in ONE launch (the weight gradients of a recurrent layer all contract the same dL/d(pre-
 * activation) matrix: launched together its rows cross the memory bus once): C[q] = A[q]^T B[q], colsum[q] (the array or an entry may be
 * NULL) = A[q]^T 1.  The arrays have nprob entries and live in host memory.  Six bf16 terms per product as slk_gemm_tn_bf16x6_f32.
 * SLK_ERR_UNSUPPORTED: rows longer than 4 M floats or slices beyond 32-bit byte offsets (-> one slk_gemm_tn_f32 per problem). */
SLK_API size_t slk_gemm_tn_multi_workspace_bytes(long M, int nprob, const int *N1, const int *N2);
SLK_API int slk_gemm_tn_multi_bf16x6_f32(int nprob, const float *const *A, const long *lda, const float *const *B, const long *ldb,
                                 float *const *C, const long *ldc, long M, const int *N1, const int *N2, float *const *colsum,
                                 void *workspace, size_t workspace_bytes, slk_stream_t stream);
SLK_API int slk_gemm_tn_f32(const float *A, long lda, const float *B, long ldb, float *C, long ldc, long M, int N1, int N2,
                    float *colsum /* [N1] or NULL */, void *workspace, size_t workspace_bytes, slk_stream_t stream);
/* The same contraction with every float32 operand cut into three bf16 pieces and each product evaluated as six bf16 MFMA terms
 * in float32 accumulators (v_mfma_f32_32x32x16_bf16): float32-grade results (terms below 2^-24 of a product are dropped), no
 * scaling needed (bf16 has float32's exponent range), ~2x the speed of the fp32 MFMA form.  Same workspace.               */
SLK_API int slk_gemm_tn_bf16x6_f32(const float *A, long lda, const float *B, long ldb, float *C, long ldc, long M, int N1, int N2,
                           float *colsum, void *workspace, size_t workspace_bytes, slk_stream_t stream);
SLK_API int slk_act_backward_f32(const float *dy, const float *y, float *out, size_t n, int act, slk_stream_t stream);
SLK_API int slk_add_inplace_f32(float *y, const float *x, size_t n, slk_stream_t stream);   /* y += x: dL/dx of Parallel branches */
SLK_API int slk_train_im2col_cin1_f32(const float *x, long x_t_stride, long x_b_stride, int T, int B, int winlen, int stride,
                              int pad_lo, int pad_hi, float *cols, slk_stream_t stream);
/* Any number of input features (a Convolution that is not the first layer, or multi-feature input): x:[T][B][Cin], rows ldx
 * floats apart; cols[(t*B + b)][c*winlen + k] = x(t*stride + k - pad_lo, b, c), zero outside the signal (the flattened column
 * order of Convolution.W:[Cout][Cin][winlen], conv.py:66-111), so dL/dW = dpre^T cols and dL/dcols = dpre . W; col2im is the
 * adjoint (dx(t', b, c) = sum of the dcols entries whose window covers t'), a gather in a fixed order.                     */
SLK_API int slk_train_im2col_f32(const float *x, long ldx, int T, int B, int Cin, int winlen, int stride, int pad_lo, int pad_hi,
                         float *cols, slk_stream_t stream);
SLK_API int slk_train_col2im_f32(const float *dcols, int T, int B, int Cin, int winlen, int stride, int pad_lo, int pad_hi, float *dx,
                         long lddx, slk_stream_t stream);
SLK_API int slk_adamski_update_f32(float *param, const float *grad, float *momentum, float *variance, size_t n, float lr_t,
                           float momentum_decay, float decay1, float decay2, float epsilon, float clip, float l2,
                           float gscale, slk_stream_t stream);
SLK_API int slk_sgd_update_f32(float *param, const float *grad, float *vel, size_t n, float rate, float momentum, float clip,
                       float l2, float gscale, slk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SLOIKA_AMD_H */
