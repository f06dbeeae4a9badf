/*
 * sloika_oracle.c -- CPU restatement of the sloika basecalling hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the `cpu_baseline` leg of bench.py may load it.  The product
 * (sloika_amd/) never imports, links or calls anything under oracle/.
 *
 * Every function restates, in plain C, the algorithm of the reference file:line it cites
 * (paths relative to the reference checkout of nanoporetech/sloika).  Array layout follows the
 * reference convention "row major (C ordering) as (time, batch, state)"  (sloika/layers.py:13-14).
 *
 * Pinning status (see tests/test_oracle_*.py and DESIGN.md "Oracle"):
 *   - orc_viterbi_kmer_*, orc_slip_update_f32, orc_map_to_sequence_f32, orc_med_mad_normalise_f32,
 *     orc_prepare_post_f32: pinned against the reference's own known-answer tests
 *     (test/unit/test_decode.py:233-256, test_viterbi.py:14-33, test_maths.py) and against golden
 *     vectors produced by importing the reference Python in the build container
 *     (tests/golden/make_goldens.py).
 *   - orc_feedforward/softmax/window: pinned by restating the reference's numpy known-answer tests
 *     (test/unit/test_layers.py:58-69, 118-125, 246-266).
 *   - orc_conv1d, orc_gru, orc_lstm (and the layer graph as a whole): pinned to the outputs of the
 *     reference's OWN layer code -- sloika/layers.py, conv.py, activation.py, models/*.py and
 *     models/pretrained.pkl imported unmodified and executed, in the build container, under an eager
 *     stand-in for the ~40 Theano primitives they call (tests/golden/theano_standin, validated by the
 *     reference's own test/unit/test_layers.py) -- fixtures tests/golden/layers.npz, checked in
 *     tests/test_oracle_reference_layers.py (this C port <= 2e-5, the float64 numpy restatement
 *     <= 1e-10).  That pins gate order, reshapes, padding, scan order and initial state.  NOT pinned:
 *     Theano's own float32 kernels (BLAS summation order, its clipped C sigmoid, <= 3.1e-7), which
 *     were never run: Theano 0.8.2 is not installable here.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp, no -ffast-math: IEEE semantics are part of the
 * contract for the integer/float DP code).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* Activations: sloika/activation.py:8-115.  Ids are shared with include/sloika_amd.h.        */
/* ------------------------------------------------------------------------------------------ */
enum {
    ORC_ACT_LINEAR = 0, ORC_ACT_TANH, ORC_ACT_SIGMOID, ORC_ACT_ELU, ORC_ACT_RELU, ORC_ACT_RELU_SMOOTH,
    ORC_ACT_SOFTPLUS, ORC_ACT_EXP, ORC_ACT_ERF, ORC_ACT_L1ML2, ORC_ACT_FAIR, ORC_ACT_RETU, ORC_ACT_TANH_PM,
    ORC_ACT_SIGMOID_PM, ORC_ACT_BOUNDED_LINEAR, ORC_ACT_SIN, ORC_ACT_CAUCHY, ORC_ACT_GEMAN_MCCLURE,
    ORC_ACT_WELSH, ORC_ACT_COUNT
};

static inline float clipf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

static inline float orc_act(int act, float x)
{
    switch (act) {
    case ORC_ACT_LINEAR: return x;                                         /* activation.py:8-9   */
    case ORC_ACT_TANH: return tanhf(x);                                    /* activation.py:52-53 */
    case ORC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));                 /* activation.py:56-57 */
    case ORC_ACT_ELU: return x > 0.0f ? x : expm1f(x);                     /* activation.py:38-42 */
    case ORC_ACT_RELU: return x > 0.0f ? x : 0.0f;                         /* activation.py:12-13 */
    case ORC_ACT_RELU_SMOOTH: {                                            /* activation.py:16-18 */
        float y = clipf(x, 0.0f, 1.0f);
        return y * y - 2.0f * y + x + fabsf(x);
    }
    case ORC_ACT_SOFTPLUS:                                                 /* activation.py:21-35 */
        return (x > 0.0f ? x : 0.0f) + log1pf(expf(-fabsf(x)));
    case ORC_ACT_EXP: return expf(x);                                      /* activation.py:45-46 */
    case ORC_ACT_ERF: return erff(x);                                      /* activation.py:60-61 */
    case ORC_ACT_L1ML2: return x / sqrtf(1.0f + 0.5f * x * x);             /* activation.py:64-65 */
    case ORC_ACT_FAIR: return x / (1.0f + fabsf(x) / 1.3998f);             /* activation.py:68-69 */
    case ORC_ACT_RETU: return tanhf(x > 0.0f ? x : 0.0f);                  /* activation.py:72-78 */
    case ORC_ACT_TANH_PM: return clipf(x, -1.0f, 1.0f);                    /* activation.py:81-85 */
    case ORC_ACT_SIGMOID_PM: return clipf(0.5f + 0.25f * x, 0.0f, 1.0f);   /* activation.py:88-92 */
    case ORC_ACT_BOUNDED_LINEAR: return clipf(x, -1.0f, 1.0f);             /* activation.py:95-98 */
    case ORC_ACT_SIN: return sinf(x);                                      /* activation.py:102-103 */
    case ORC_ACT_CAUCHY: { float u = x / 2.3849f; return x / (1.0f + u * u); }  /* :106-107 */
    case ORC_ACT_GEMAN_MCCLURE: { float u = 1.0f + x * x; return x / (u * u); } /* :110-111 */
    case ORC_ACT_WELSH: { float u = x / 2.9846f; return x * expf(-(u * u)); }   /* :114-115 */
    default: return NAN;
    }
}

float orc_activation_f32(int act, float x) { return orc_act(act, x); }

/* dot product with 8 independent partial sums so that gcc can vectorise without -ffast-math */
static inline float dotf(const float *restrict a, const float *restrict b, int n)
{
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= n; k += 8)
        for (int l = 0; l < 8; l++) acc[l] += a[k + l] * b[k + l];
    float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    for (; k < n; k++) s += a[k] * b[k];
    return s;
}

/* ------------------------------------------------------------------------------------------ */
/* Convolution.run: layers.py:417-419 -> conv.conv_1d conv.py:90-111 (zero pad_first :66-77,   */
/* cross-correlation filter_flip=False, subsample=(1,stride), 'valid' border on the padded     */
/* signal).  x:[T][B][Cin]  W:[Cout][Cin][winlen]  y:[Tout][B][Cout],                          */
/* Tout = (T + pad_l + pad_r - winlen) / stride + 1.   b may be NULL (has_bias=False).         */
/* ------------------------------------------------------------------------------------------ */
int orc_conv1d_out_len(int T, int winlen, int stride, int pad_l, int pad_r)
{
    int tp = T + pad_l + pad_r;
    if (tp < winlen) return 0;
    return (tp - winlen) / stride + 1;
}

void orc_conv1d_f32(const float *x, int T, int B, int Cin, const float *W, const float *b, int Cout,
                    int winlen, int stride, int pad_l, int pad_r, int act, float *y)
{
    int Tout = orc_conv1d_out_len(T, winlen, stride, pad_l, pad_r);
#pragma omp parallel for collapse(2) schedule(static)
    for (int to = 0; to < Tout; to++) {
        for (int bb = 0; bb < B; bb++) {
            float *yo = y + ((size_t)to * B + bb) * Cout;
            for (int o = 0; o < Cout; o++) {
                float s = 0.0f;
                for (int c = 0; c < Cin; c++) {
                    const float *w = W + ((size_t)o * Cin + c) * winlen;
                    for (int k = 0; k < winlen; k++) {
                        int ti = to * stride + k - pad_l;
                        if (ti < 0 || ti >= T) continue; /* zero padding */
                        s += x[((size_t)ti * B + bb) * Cin + c] * w[k];
                    }
                }
                if (b) s += b[o];
                yo[o] = orc_act(act, s);
            }
        }
    }
}

/* Window.run: layers.py:346-351.  zero pad w//2 both ends, out[t] = concat_k x_pad[t+k];      */
/* feature index of the output is k*F + f.   x:[T][B][F] -> y:[T][B][w*F]                      */
void orc_window_f32(const float *x, int T, int B, int F, int w, float *y)
{
    int half = w / 2;
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; t++)
        for (int bb = 0; bb < B; bb++)
            for (int k = 0; k < w; k++) {
                int ti = t + k - half;
                float *yo = y + (((size_t)t * B + bb) * w + k) * F;
                if (ti < 0 || ti >= T) memset(yo, 0, sizeof(float) * F);
                else memcpy(yo, x + ((size_t)ti * B + bb) * F, sizeof(float) * F);
            }
}

/* FeedForward.run: layers.py:157-158   y = fun(x . W^T + b);  x:[rows][I]  W:[N][I]  y:[rows][N] */
void orc_feedforward_f32(const float *x, size_t rows, int I, const float *W, const float *b, int N, int act,
                         float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) {
        const float *xr = x + r * I;
        float *yr = y + r * N;
        for (int j = 0; j < N; j++) {
            float s = dotf(xr, W + (size_t)j * I, I);
            if (b) s += b[j];
            yr[j] = orc_act(act, s);
        }
    }
}

/* Softmax.run: layers.py:309-314   tmp = x.W^T + b; m = max; out = exp(tmp-m); out / sum(out) */
void orc_softmax_f32(const float *x, size_t rows, int I, const float *W, const float *b, int N, float *y)
{
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) {
        const float *xr = x + r * I;
        float *yr = y + r * N;
        float m = -INFINITY;
        for (int j = 0; j < N; j++) {
            float s = dotf(xr, W + (size_t)j * I, I);
            if (b) s += b[j];
            yr[j] = s;
            if (s > m) m = s;
        }
        float sum = 0.0f;
        for (int j = 0; j < N; j++) {
            yr[j] = expf(yr[j] - m);
            sum += yr[j];
        }
        for (int j = 0; j < N; j++) yr[j] = yr[j] / sum;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Gru: layers.py:952-1021, step :1010-1021; scan with zero initial state layers.py:85-88;     */
/* Reverse: layers.py:1449-1450 (run on time-flipped input, flip output back) == iterate t     */
/* from T-1 down to 0 writing y[t].                                                            */
/*   vI = x_t.iW^T + b        iW:[3n][I] rows [0,n)=z [n,2n)=r [2n,3n)=candidate               */
/*   vS = h.sW^T              sW:[2n][n]                                                       */
/*   z = gate(vI_z+vS_z)  r = gate(vI_r+vS_r)                                                  */
/*   y = (r*h).sW2^T          sW2:[n][n]                                                       */
/*   hbar = fun(vI_h + y);  h = z*h + (1-z)*hbar                                               */
/* ------------------------------------------------------------------------------------------ */
void orc_gru_f32(const float *x, int T, int B, int I, const float *iW, const float *sW, const float *sW2,
                 const float *b, int n, int reverse, int act, int gate_act, float *y)
{
#pragma omp parallel
    {
        float *h = (float *)malloc(sizeof(float) * n * 4);
        float *rh = h + n, *z = h + 2 * n, *hn = h + 3 * n;
#pragma omp for schedule(static)
        for (int bb = 0; bb < B; bb++) {
            for (int j = 0; j < n; j++) h[j] = 0.0f;
            for (int s = 0; s < T; s++) {
                int t = reverse ? T - 1 - s : s;
                const float *xt = x + ((size_t)t * B + bb) * I;
                for (int j = 0; j < n; j++) {
                    float vz = dotf(xt, iW + (size_t)j * I, I) + (b ? b[j] : 0.0f);
                    float vr = dotf(xt, iW + (size_t)(n + j) * I, I) + (b ? b[n + j] : 0.0f);
                    vz = vz + dotf(h, sW + (size_t)j * n, n);
                    vr = vr + dotf(h, sW + (size_t)(n + j) * n, n);
                    z[j] = orc_act(gate_act, vz);
                    rh[j] = orc_act(gate_act, vr) * h[j];
                }
                for (int j = 0; j < n; j++) {
                    float vh = dotf(xt, iW + (size_t)(2 * n + j) * I, I) + (b ? b[2 * n + j] : 0.0f);
                    float yy = dotf(rh, sW2 + (size_t)j * n, n);
                    float hbar = orc_act(act, vh + yy);
                    hn[j] = z[j] * h[j] + (1.0f - z[j]) * hbar;
                }
                float *yt = y + ((size_t)t * B + bb) * n;
                for (int j = 0; j < n; j++) { h[j] = hn[j]; yt[j] = hn[j]; }
            }
        }
        free(h);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Lstm: layers.py:599-697, step :677-691 (authoritative INTERLEAVED gate layout row = j*4+g,   */
/* g: 0 update, 1 input gate, 2 forget gate, 3 output gate), run :693-697 (state = [out, cell],*/
/* zero init; output = out part).   iW:[4n][I]  sW:[4n][n]  b:[4n] or NULL  p:[3][n] or NULL   */
/* ------------------------------------------------------------------------------------------ */
void orc_lstm_f32(const float *x, int T, int B, int I, const float *iW, const float *sW, const float *b,
                  const float *p, int n, int reverse, int act, int gate_act, float *y)
{
#pragma omp parallel
    {
        float *out = (float *)malloc(sizeof(float) * n * 6);
        float *cell = out + n, *sum = out + 2 * n; /* sum: [n][4] */
#pragma omp for schedule(static)
        for (int bb = 0; bb < B; bb++) {
            for (int j = 0; j < n; j++) { out[j] = 0.0f; cell[j] = 0.0f; }
            for (int s = 0; s < T; s++) {
                int t = reverse ? T - 1 - s : s;
                const float *xt = x + ((size_t)t * B + bb) * I;
                for (int r = 0; r < 4 * n; r++) {
                    float v = dotf(xt, iW + (size_t)r * I, I);
                    v = v + dotf(out, sW + (size_t)r * n, n);
                    if (b) v = v + b[r];
                    sum[r] = v;
                }
                float *yt = y + ((size_t)t * B + bb) * n;
                for (int j = 0; j < n; j++) {
                    float st = cell[j];
                    float p0 = p ? p[j] : 0.0f, p1 = p ? p[n + j] : 0.0f, p2 = p ? p[2 * n + j] : 0.0f;
                    float os = st * orc_act(gate_act, sum[j * 4 + 2] + st * p1);
                    os += orc_act(act, sum[j * 4 + 0]) * orc_act(gate_act, sum[j * 4 + 1] + st * p0);
                    float o = orc_act(act, os) * orc_act(gate_act, sum[j * 4 + 3] + os * p2);
                    cell[j] = os;
                    yt[j] = o;
                }
                for (int j = 0; j < n; j++) out[j] = yt[j];
            }
        }
        free(out);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Median / MAD normalisation of chunks: tools/chunkify_raw.py:178-181 ("per-chunk"),          */
/* maths.py:4-27 (factor 1.4826), numpy median of an even count = float32 mean of the two      */
/* middle order statistics.   signal:[nchunk][chunk_len] -> out same shape.                    */
/* Also returns med/mad per chunk when the pointers are non-NULL.                              */
/* ------------------------------------------------------------------------------------------ */
static int cmp_float(const void *a, const void *b)
{
    float fa = *(const float *)a, fb = *(const float *)b;
    return (fa > fb) - (fa < fb);
}

static float median_inplace(float *v, int n)
{
    qsort(v, n, sizeof(float), cmp_float);
    if (n & 1) return v[n / 2];
    return (v[n / 2 - 1] + v[n / 2]) / 2.0f;
}

void orc_med_mad_normalise_f32(const float *signal, int nchunk, int chunk_len, float *out, float *med_out,
                               float *mad_out)
{
#pragma omp parallel
    {
        float *tmp = (float *)malloc(sizeof(float) * chunk_len);
#pragma omp for schedule(static)
        for (int c = 0; c < nchunk; c++) {
            const float *s = signal + (size_t)c * chunk_len;
            memcpy(tmp, s, sizeof(float) * chunk_len);
            float med = median_inplace(tmp, chunk_len);
            for (int i = 0; i < chunk_len; i++) tmp[i] = fabsf(s[i] - med);
            float mad = 1.4826f * median_inplace(tmp, chunk_len);
            for (int i = 0; i < chunk_len; i++) out[(size_t)c * chunk_len + i] = (s[i] - med) / mad;
            if (med_out) med_out[c] = med;
            if (mad_out) mad_out[c] = mad;
        }
        free(tmp);
    }
}

/* decode.prepare_post: decode.py:21-36 (drop_bad=False branch): min_prob + (1-min_prob)*post.  */
/* numpy evaluates (1.0 - min_prob) in double, then multiplies the float32 array by that value  */
/* cast to float32, then adds float32(min_prob).                                                */
void orc_prepare_post_f32(const float *post, size_t n, double min_prob, float *out)
{
    float one_m = (float)(1.0 - min_prob), mp = (float)min_prob;
    for (size_t i = 0; i < n; i++) out[i] = mp + one_m * post[i];
}

/* ------------------------------------------------------------------------------------------ */
/* decode.viterbi: decode.py:39-93.  lpost:[T][nst] already in log space (the caller applies    */
/* np.log(post + 1e-10), decode.py:56, exactly as the reference does).                          */
/*   step: max over a of v[a*nkmer/nbase + s/nbase]              (first max wins, np.argmax)     */
/*   skip: max over ab of v[ab*nkmer/nbase^2 + s/nbase^2] - pen  (first max wins)                */
/*   tb = step > skip ? from_step : from_skip  (tie -> skip)    decode.py:76                     */
/*   stay = v[s] + lpost[t][0]; tb = new > stay ? tb : -1 (tie -> stay)  decode.py:79-82         */
/* Backtrace decode.py:84-91: start at first argmax of v, prepend tb when >= 0.                  */
/* Returns path in time order (length *len_out <= T).  tb_work: int32 [T][nkmer] scratch.        */
/* ------------------------------------------------------------------------------------------ */
#define DEFINE_VITERBI(NAME, real)                                                                       \
    int NAME(const real *lpost, int T, int nbase, int klen, double skip_pen, real *score_out,            \
             int32_t *path_out, int32_t *len_out, int32_t *tb_work)                                      \
    {                                                                                                    \
        if (klen < 3 || T < 1) return -1; /* decode.py:50 */                                             \
        int nkmer = 1;                                                                                   \
        for (int i = 0; i < klen; i++) nkmer *= nbase;                                                   \
        int nst = nkmer + 1;                                                                             \
        int nrem1 = nkmer / nbase, nrem2 = nkmer / (nbase * nbase);                                      \
        real pen = (real)skip_pen;                                                                       \
        real *vbase = (real *)malloc(sizeof(real) * nkmer * 2);                                          \
        real *v = vbase, *pv = vbase + nkmer;                                                            \
        int32_t *tb = tb_work;                                                                           \
        int own_tb = 0;                                                                                  \
        if (!tb) { tb = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * nkmer); own_tb = 1; }            \
        for (int s = 0; s < nkmer; s++) v[s] = lpost[1 + s];                                             \
        for (int t = 1; t < T; t++) {                                                                    \
            const real *lp = lpost + (size_t)t * nst;                                                    \
            real *tmp = pv; pv = v; v = tmp;                                                             \
            int32_t *tbt = tb + (size_t)t * nkmer;                                                       \
            for (int s = 0; s < nkmer; s++) {                                                            \
                int j1 = s / nbase, j2 = s / (nbase * nbase);                                            \
                real best1 = pv[j1]; int a1 = 0;                                                         \
                for (int a = 1; a < nbase; a++)                                                          \
                    if (pv[a * nrem1 + j1] > best1) { best1 = pv[a * nrem1 + j1]; a1 = a; }              \
                real best2 = pv[j2]; int a2 = 0;                                                         \
                for (int a = 1; a < nbase * nbase; a++)                                                  \
                    if (pv[a * nrem2 + j2] > best2) { best2 = pv[a * nrem2 + j2]; a2 = a; }              \
                real sstep = best1, sskip = best2 - pen;                                                 \
                real mx = sstep > sskip ? sstep : sskip;                                                 \
                real nv = lp[1 + s] + mx;                                                                \
                int32_t from = sstep > sskip ? a1 * nrem1 + j1 : a2 * nrem2 + j2;                        \
                real stay = pv[s] + lp[0];                                                               \
                tbt[s] = nv > stay ? from : -1;                                                          \
                v[s] = nv > stay ? nv : stay;                                                            \
            }                                                                                            \
        }                                                                                                \
        int best = 0;                                                                                    \
        for (int s = 1; s < nkmer; s++) if (v[s] > v[best]) best = s;                                    \
        *score_out = v[best];                                                                            \
        /* backtrace: fill from the right, then shift left */                                            \
        int pos = T;                                                                                     \
        int32_t cur = best;                                                                              \
        path_out[--pos] = cur;                                                                           \
        for (int t = T - 1; t >= 1; t--) {                                                               \
            int32_t ts = tb[(size_t)t * nkmer + cur];                                                    \
            if (ts >= 0) { cur = ts; path_out[--pos] = cur; }                                            \
        }                                                                                                \
        int len = T - pos;                                                                               \
        memmove(path_out, path_out + pos, sizeof(int32_t) * len);                                        \
        *len_out = len;                                                                                  \
        if (own_tb) free(tb);                                                                            \
        free(vbase);                                                                                     \
        return 0;                                                                                        \
    }

DEFINE_VITERBI(orc_viterbi_kmer_f32, float)
DEFINE_VITERBI(orc_viterbi_kmer_f64, double)

/* Batched wrapper over chunks for the CPU baseline: lpost:[T][B][nst] (network layout).         */
int orc_viterbi_kmer_batch_f32(const float *lpost, int T, int B, int nbase, int klen, double skip_pen,
                               float *score_out, int32_t *path_out /*[B][T]*/, int32_t *len_out)
{
    int nkmer = 1;
    for (int i = 0; i < klen; i++) nkmer *= nbase;
    int nst = nkmer + 1, rc = 0;
#pragma omp parallel
    {
        float *lp = (float *)malloc(sizeof(float) * (size_t)T * nst);
#pragma omp for schedule(dynamic)
        for (int bb = 0; bb < B; bb++) {
            for (int t = 0; t < T; t++)
                memcpy(lp + (size_t)t * nst, lpost + ((size_t)t * B + bb) * nst, sizeof(float) * nst);
            int r = orc_viterbi_kmer_f32(lp, T, nbase, klen, skip_pen, score_out + bb, path_out + (size_t)bb * T,
                                         len_out + bb, NULL);
            if (r) rc = r;
        }
        free(lp);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* viterbi_helpers.slip_update: viterbi_helpers.pyx:12-35.  Requires n >= 3 (the reference      */
/* writes index 2 unconditionally).  from_pos is int64 (the Cython `np.int_t` = C long).        */
/* ------------------------------------------------------------------------------------------ */
int orc_slip_update_f32(const float *x, int n, float slip, float *from_score, int64_t *from_pos)
{
    if (n < 3) return -1;
    for (int j = 0; j < n; j++) { from_score[j] = 0.0f; from_pos[j] = 0; }
    from_score[0] = from_score[1] = -1e38f;
    from_score[2] = x[0] - slip;
    from_pos[2] = 0;
    for (int j = 3; j < n; j++) {
        if (from_score[j - 1] >= x[j - 2]) {
            from_pos[j] = from_pos[j - 1];
            from_score[j] = from_score[j - 1];
        } else {
            from_pos[j] = j - 2;
            from_score[j] = x[j - 2];
        }
        from_score[j] -= slip;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* transducer.map_to_sequence: transducer.py:14-73.  ltrans:[nev][nst] log space, seq:[npos]    */
/* state indices (>=1), slip >= 0, prior_initial/prior_final: float64 [npos] or NULL (the       */
/* reference adds float64 priors into the float32 score vector in place: :39-41, :63-64).       */
/* path_out:[nev] int32.  vmat_work: int32 [nev][npos] scratch or NULL.                          */
/* ------------------------------------------------------------------------------------------ */
int orc_map_to_sequence_f32(const float *ltrans, int nev, int nst, const int32_t *seq, int npos, float slip,
                            const double *prior_initial, const double *prior_final, float *score_out,
                            int32_t *path_out, int32_t *vmat_work)
{
    if (npos < 3 || nev < 1) return -1;
    int32_t *vmat = vmat_work;
    int own = 0;
    if (!vmat) { vmat = (int32_t *)calloc((size_t)nev * npos, sizeof(int32_t)); own = 1; }
    else memset(vmat, 0, sizeof(int32_t) * (size_t)nev * npos);
    float *sbase = (float *)calloc((size_t)npos * 3, sizeof(float));
    float *pscore = sbase, *cscore = sbase + npos, *fs = sbase + 2 * npos;
    int64_t *fp = (int64_t *)malloc(sizeof(int64_t) * npos);
    if (prior_initial)
        for (int j = 0; j < npos; j++) pscore[j] = (float)((double)pscore[j] + prior_initial[j]);
    for (int j = 0; j < npos; j++) pscore[j] += fmaxf(ltrans[seq[j]], ltrans[0]); /* :41 */
    for (int i = 1; i < nev; i++) {
        const float *ct = ltrans + (size_t)i * nst;
        int32_t *vm = vmat + (size_t)i * npos;
        for (int j = 0; j < npos; j++) { vm[j] = j; cscore[j] = pscore[j] + ct[0]; } /* stay :46-47 */
        for (int j = 0; j + 1 < npos; j++) {                                          /* step :49-52 */
            float ss = pscore[j] + ct[seq[j + 1]];
            if (ss > cscore[j + 1]) { cscore[j + 1] = ss; vm[j + 1] = j; }
        }
        orc_slip_update_f32(pscore, npos, slip, fs, fp);                              /* slip :55-59 */
        for (int j = 0; j < npos; j++) {
            float f = fs[j] + ct[seq[j]];
            if (!(f <= cscore[j])) { vm[j] = (int32_t)fp[j]; cscore[j] = f; }
        }
        float *tmp = pscore; pscore = cscore; cscore = tmp;
    }
    if (prior_final)
        for (int j = 0; j < npos; j++) pscore[j] = (float)((double)pscore[j] + prior_final[j]);
    int best = 0;
    for (int j = 1; j < npos; j++) if (pscore[j] > pscore[best]) best = j;
    *score_out = pscore[best];
    /* traceback :66-71 */
    int32_t cur = best;
    path_out[nev - 1] = cur;
    for (int i = 1; i < nev; i++) {
        cur = vmat[(size_t)(nev - i) * npos + cur];
        path_out[nev - 1 - i] = cur;
    }
    free(fp);
    free(sbase);
    if (own) free(vmat);
    return 0;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
