// frontend.hip -- chunk front end of the basecalling path on gfx950:
//   * per-chunk median/MAD normalisation   (sloika/tools/chunkify_raw.py:172-181, sloika/maths.py:4-27)
//   * Convolution.run                       (sloika/layers.py:417-419, sloika/conv.py:66-111)
//   * Window.run                            (sloika/layers.py:346-351)
// All three are HBM/latency-bound byte movers: coalesced loads/stores, filter taps and sort keys staged in LDS.
#include "common.h"

// ------------------------------------------------------------------------------------------------------
// median / MAD.  One workgroup per chunk; the chunk is sorted in LDS with a bitonic network (padding
// with +inf up to a power of two), twice: once for the median, once for the median absolute deviation.
// numpy semantics: even count -> (a + b) / 2 in float32; mad = float32(1.4826) * median(|x - med|);
// out = (x - med) / mad with IEEE division (hipcc's default correctly rounded fp32 divide).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bitonic_sort_lds(float *s, int npow2, int tid, int nthreads)
{
    for (int k = 2; k <= npow2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < npow2; i += nthreads) {
                int ixj = i ^ j;
                if (ixj > i) {
                    float a = s[i], b = s[ixj];
                    bool up = (i & k) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ float median_sorted(const float *s, int n)
{
    if (n & 1) return s[n >> 1];
    return (s[(n >> 1) - 1] + s[n >> 1]) / 2.0f;
}

__global__ void __launch_bounds__(512) med_mad_kernel(const float *__restrict__ signal, int chunk_len, int npow2,
                                                      float *__restrict__ out, long out_chunk_stride,
                                                      long out_sample_stride, float *__restrict__ med_out,
                                                      float *__restrict__ mad_out)
{
    extern __shared__ float srt[];
    const int c = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const float *sig = signal + (size_t)c * chunk_len;
    for (int i = tid; i < npow2; i += nt) srt[i] = i < chunk_len ? sig[i] : INFINITY;
    __syncthreads();
    bitonic_sort_lds(srt, npow2, tid, nt);
    const float med = median_sorted(srt, chunk_len);
    __syncthreads();
    for (int i = tid; i < npow2; i += nt) srt[i] = i < chunk_len ? fabsf(sig[i] - med) : INFINITY;
    __syncthreads();
    bitonic_sort_lds(srt, npow2, tid, nt);
    const float mad = 1.4826f * median_sorted(srt, chunk_len);
    float *o = out + (size_t)c * out_chunk_stride;
    for (int i = tid; i < chunk_len; i += nt) o[(size_t)i * out_sample_stride] = (sig[i] - med) / mad;
    if (tid == 0) {
        if (med_out) med_out[c] = med;
        if (mad_out) mad_out[c] = mad;
    }
}

extern "C" int slk_med_mad_normalise_f32(const float *signal, int nchunk, int chunk_len, float *out,
                                         long out_chunk_stride, long out_sample_stride, float *med_out,
                                         float *mad_out, slk_stream_t stream)
{
    if (!signal || !out || nchunk < 0 || chunk_len < 1) return SLK_ERR_INVALID_ARG;
    if (chunk_len > 32768) return SLK_ERR_UNSUPPORTED;   // one chunk must fit the CU's LDS (128 KiB of sort keys)
    if (nchunk == 0) return SLK_OK;
    int npow2 = 1;
    while (npow2 < chunk_len) npow2 <<= 1;
    int threads = npow2 / 2 < 64 ? 64 : (npow2 / 2 > 512 ? 512 : npow2 / 2);
    if (npow2 * sizeof(float) > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(med_mad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(npow2 * sizeof(float))) != hipSuccess)
        return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(med_mad_kernel, dim3(nchunk), dim3(threads), npow2 * sizeof(float), slk_stream(stream),
                       signal, chunk_len, npow2, out, out_chunk_stride, out_sample_stride, med_out, mad_out);
    return slk_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// conv1d.  Thread <-> one output element, Cout fastest so that stores are fully coalesced; the filter is
// staged transposed in LDS (Wt[c][k][o]) so that lanes read consecutive words; the Cin*winlen input taps
// of one (to, b) are shared by the Cout lanes computing it and come from L1/L2.
// ------------------------------------------------------------------------------------------------------
template <bool W_IN_LDS>
__global__ void __launch_bounds__(256) conv1d_kernel(const float *__restrict__ x, long xs_t, long xs_b,
                                                     const float *__restrict__ W, const float *__restrict__ bias,
                                                     float *__restrict__ y, int T, int B, int Cin, int Cout,
                                                     int winlen, int stride, int pad_l, int Tout, int act)
{
    extern __shared__ float wt[];
    const int ckn = Cin * winlen;
    if (W_IN_LDS) {
        for (int i = threadIdx.x; i < ckn * Cout; i += blockDim.x) {
            int o = i / ckn, ck = i - o * ckn;
            wt[ck * Cout + o] = W[i];
        }
        __syncthreads();
    }
    const size_t total = (size_t)Tout * B * Cout;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int o = (int)(idx % Cout);
        size_t tb = idx / Cout;
        int b = (int)(tb % B);
        int to = (int)(tb / B);
        const float *xb = x + (size_t)b * xs_b;
        float s = 0.0f;
        for (int c = 0; c < Cin; c++) {
            for (int k = 0; k < winlen; k++) {
                int ti = to * stride + k - pad_l;
                if (ti < 0 || ti >= T) continue;
                float xv = xb[(size_t)ti * xs_t + c];
                float wv = W_IN_LDS ? wt[(c * winlen + k) * Cout + o] : W[((size_t)o * Cin + c) * winlen + k];
                s = fmaf(xv, wv, s);
            }
        }
        if (bias) s += bias[o];
        y[idx] = slk_act(act, s);
    }
}

extern "C" int slk_conv1d_out_len(int T, int winlen, int stride, int pad_l, int pad_r)
{
    if (winlen < 1 || stride < 1) return 0;
    int tp = T + pad_l + pad_r;
    if (tp < winlen) return 0;
    return (tp - winlen) / stride + 1;
}

extern "C" int slk_conv1d_f32(const float *x, long x_t_stride, long x_b_stride, const float *W, const float *bias,
                              float *y, int T, int B, int Cin, int Cout, int winlen, int stride, int pad_l, int pad_r,
                              int act, slk_stream_t stream)
{
    if (!x || !W || !y || T < 1 || B < 1 || Cin < 1 || Cout < 1 || winlen < 1 || stride < 1 || pad_l < 0 || pad_r < 0 ||
        !slk_act_valid(act))
        return SLK_ERR_INVALID_ARG;
    int Tout = slk_conv1d_out_len(T, winlen, stride, pad_l, pad_r);
    if (Tout <= 0) return SLK_ERR_INVALID_ARG;
    size_t total = (size_t)Tout * B * Cout;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    size_t wbytes = (size_t)Cin * winlen * Cout * sizeof(float);
    if (wbytes <= 64 * 1024)
        hipLaunchKernelGGL(conv1d_kernel<true>, dim3((unsigned)blocks), dim3(256), wbytes, slk_stream(stream), x,
                           x_t_stride, x_b_stride, W, bias, y, T, B, Cin, Cout, winlen, stride, pad_l, Tout, act);
    else
        hipLaunchKernelGGL(conv1d_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), x,
                           x_t_stride, x_b_stride, W, bias, y, T, B, Cin, Cout, winlen, stride, pad_l, Tout, act);
    return slk_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// Window: pure gather, one thread per output element.
// ------------------------------------------------------------------------------------------------------
__global__ void window_kernel(const float *__restrict__ x, float *__restrict__ y, int T, int B, int F, int w)
{
    const size_t total = (size_t)T * B * w * F;
    const int half = w / 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        int f = (int)(idx % F);
        size_t r = idx / F;
        int k = (int)(r % w);
        r /= w;
        int b = (int)(r % B);
        int t = (int)(r / B);
        int ti = t + k - half;
        y[idx] = (ti < 0 || ti >= T) ? 0.0f : x[((size_t)ti * B + b) * F + f];
    }
}

extern "C" int slk_window_f32(const float *x, float *y, int T, int B, int F, int w, slk_stream_t stream)
{
    if (!x || !y || T < 1 || B < 1 || F < 1 || w < 1 || (w & 1) == 0) return SLK_ERR_INVALID_ARG; // layers.py:328-329
    size_t total = (size_t)T * B * w * F;
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(window_kernel, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), x, y, T, B, F, w);
    return slk_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// stand-alone activation (the layers apply activations inside their own kernels; this is the API-level twin)
// ------------------------------------------------------------------------------------------------------
__global__ void activation_kernel(const float *__restrict__ x, float *__restrict__ y, size_t count, int act)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        y[i] = slk_act(act, x[i]);
}

extern "C" int slk_activation_f32(const float *x, float *y, size_t count, int act, slk_stream_t stream)
{
    if (!x || !y || !slk_act_valid(act)) return SLK_ERR_INVALID_ARG;
    if (!count) return SLK_OK;
    size_t blocks = (count + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(activation_kernel, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), x, y, count, act);
    return slk_launch_status();
}
