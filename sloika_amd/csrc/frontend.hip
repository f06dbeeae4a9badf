// frontend.hip -- chunk front end of the basecalling path on gfx950:
//   * per-chunk median/MAD normalisation   (sloika/tools/chunkify_raw.py:172-181, sloika/maths.py:4-27)
//   * Convolution.run                       (sloika/layers.py:417-419, sloika/conv.py:66-111)
//   * Window.run                            (sloika/layers.py:346-351)
// All three are HBM/latency-bound byte movers: coalesced loads/stores, filter taps and sort keys staged in LDS.
#include <limits.h>

#include "common.h"

// ------------------------------------------------------------------------------------------------------
// median / MAD.  One workgroup per chunk; the chunk is sorted ONCE in LDS with a bitonic network (padding
// with +inf up to a power of two); the median absolute deviation comes from the same sorted array.
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
    // No second sort: |x - med| over the sorted samples is two monotone runs (the samples below the median, walked
    // downwards, and those above it, walked upwards; float32 rounding is monotone), so the order statistics of the
    // deviations are order statistics of the merge of two sorted sequences -- a binary search.
    __shared__ float mad_sh;
    if (tid == 0) {
        const int n = chunk_len, p = n >> 1, nA = p, nB = n - p;        // A[j] = |s[p-1-j] - med|, B[j] = |s[p+j] - med|
        auto A = [&](int j) { return fabsf(srt[p - 1 - j] - med); };
        auto Bv = [&](int j) { return fabsf(srt[p + j] - med); };
        auto kth = [&](int r) {                                          // value of rank r (0-based) in the merge
            int lo = max(0, r + 1 - nB), hi = min(r + 1, nA);
            while (lo < hi) {
                const int a = (lo + hi) >> 1, b = r + 1 - a;             // a < hi <= nA, b >= 1
                if (A(a) < Bv(b - 1)) lo = a + 1;
                else hi = a;
            }
            const int a = lo, b = r + 1 - a;
            float v = -INFINITY;
            if (a > 0) v = A(a - 1);
            if (b > 0) v = fmaxf(v, Bv(b - 1));
            return v;
        };
        const float dm = (n & 1) ? kth(n >> 1) : (kth((n >> 1) - 1) + kth(n >> 1)) / 2.0f;
        mad_sh = 1.4826f * dm;
    }
    __syncthreads();
    const float mad = mad_sh;
    float *o = out + (size_t)c * out_chunk_stride;
    for (int i = tid; i < chunk_len; i += nt) o[(size_t)i * out_sample_stride] = (sig[i] - med) / mad;
    if (tid == 0) {
        if (med_out) med_out[c] = med;
        if (mad_out) mad_out[c] = mad;
    }
}

// ------------------------------------------------------------------------------------------------------
// Long chunks / whole reads (chunk_len > 32768: does not fit the LDS sort): exact order statistics by radix
// selection.  Float keys are mapped to unsigned integers that sort the same way; four passes of 8 bits each narrow
// the bucket that contains the wanted rank (histogram in LDS, one workgroup per chunk).  The median of an even count
// needs ranks n/2-1 and n/2; MAD repeats the selection on |x - med| recomputed on the fly (no scratch memory).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2key(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// value of rank `rank` (0-based) among f(sig[i]), f = identity or |x - med|
template <bool ABSDEV>
__device__ float radix_select(const float *__restrict__ sig, int n, float med, unsigned rank, unsigned *hist,
                              unsigned *sh)
{
    const int tid = threadIdx.x, nt = blockDim.x;
    unsigned prefix = 0, mask = 0;
    for (int pass = 3; pass >= 0; pass--) {
        const int shift = 8 * pass;
        for (int i = tid; i < 256; i += nt) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += nt) {
            float v = sig[i];
            if (ABSDEV) v = fabsf(v - med);
            const unsigned k = f2key(v);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned acc = 0, bin = 0;
            for (; bin < 256; bin++) {
                if (acc + hist[bin] > rank) break;
                acc += hist[bin];
            }
            sh[0] = bin;
            sh[1] = rank - acc;
        }
        __syncthreads();
        prefix |= sh[0] << shift;
        mask |= 0xffu << shift;
        rank = sh[1];
        __syncthreads();
    }
    return key2f(prefix);
}

template <bool ABSDEV>
__device__ float radix_median(const float *__restrict__ sig, int n, float med, unsigned *hist, unsigned *sh)
{
    if (n & 1) return radix_select<ABSDEV>(sig, n, med, (unsigned)(n >> 1), hist, sh);
    const float a = radix_select<ABSDEV>(sig, n, med, (unsigned)(n >> 1) - 1, hist, sh);
    const float b = radix_select<ABSDEV>(sig, n, med, (unsigned)(n >> 1), hist, sh);
    return (a + b) / 2.0f;
}

__global__ void __launch_bounds__(1024) med_mad_radix_kernel(const float *__restrict__ signal, int chunk_len,
                                                             float *__restrict__ out, long out_chunk_stride,
                                                             long out_sample_stride, float *__restrict__ med_out,
                                                             float *__restrict__ mad_out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[2];
    const int c = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const float *sig = signal + (size_t)c * chunk_len;
    const float med = radix_median<false>(sig, chunk_len, 0.0f, hist, sh);
    const float mad = 1.4826f * radix_median<true>(sig, chunk_len, med, hist, sh);
    float *o = out + (size_t)c * out_chunk_stride;
    for (int i = tid; i < chunk_len; i += nt) o[(size_t)i * out_sample_stride] = (sig[i] - med) / mad;
    if (tid == 0) {
        if (med_out) med_out[c] = med;
        if (mad_out) mad_out[c] = mad;
    }
}

// Whole reads of DIFFERENT lengths, one workgroup each: read r is lens[r] samples at signal + r * in_stride; only its own samples
// are written (a padded batch keeps the zeros the caller put behind them).
__global__ void __launch_bounds__(1024) med_mad_radix_ragged_kernel(const float *__restrict__ signal, long in_stride,
                                                                    const int *__restrict__ lens, float *__restrict__ out,
                                                                    long out_chunk_stride, long out_sample_stride,
                                                                    float *__restrict__ med_out, float *__restrict__ mad_out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[2];
    const int c = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int n = lens[c];
    if (n < 1) return;
    const float *sig = signal + (size_t)c * in_stride;
    const float med = radix_median<false>(sig, n, 0.0f, hist, sh);
    const float mad = 1.4826f * radix_median<true>(sig, n, med, hist, sh);
    float *o = out + (size_t)c * out_chunk_stride;
    for (int i = tid; i < n; i += nt) o[(size_t)i * out_sample_stride] = (sig[i] - med) / mad;
    if (tid == 0) {
        if (med_out) med_out[c] = med;
        if (mad_out) mad_out[c] = mad;
    }
}

extern "C" int slk_med_mad_normalise_ragged_f32(const float *signal, int nread, long in_stride, const int32_t *lens, float *out,
                                                long out_chunk_stride, long out_sample_stride, float *med_out, float *mad_out,
                                                slk_stream_t stream)
{
    if (!signal || !out || !lens || nread < 0 || in_stride < 1) return SLK_ERR_INVALID_ARG;
    if (nread == 0) return SLK_OK;
    hipLaunchKernelGGL(med_mad_radix_ragged_kernel, dim3(nread), dim3(1024), 0, slk_stream(stream), signal, in_stride, lens, out,
                       out_chunk_stride, out_sample_stride, med_out, mad_out);
    return slk_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// Chunks of up to 4096 samples (the basecaller's 4000-sample chunks): no sort at all.  Each of 256 threads keeps 16 samples
// in registers as order-preserving integer keys; the wanted order statistic is built from its most significant bit down,
// two bits per round: count the keys below three candidate prefixes (register compares, a wave reduction, one exchange
// through LDS) and keep the largest prefix that at most `rank` keys lie below.  16 rounds give the exact key of rank n/2-1;
// its upper neighbour (rank n/2) is either the same key (duplicates) or the smallest larger key -- one more count and a
// minimum.  The MAD repeats this on the keys of |x - med|.  33 us for 1024 chunks of 4000 samples (7 of them the loads and stores;
// round 5: counts on the vector unit 45 -> 40, the lower 16 bits on the collected keys -> 33; tools/mm_ab.py) against 140 us
// for the LDS bitonic sort (which moves 32 KB through LDS in each of its 78 stages).
// ------------------------------------------------------------------------------------------------------
template <int EPT>
struct KeySet {
    unsigned k[EPT];
};

// number of keys < each of the three candidates, summed over the workgroup (every thread gets the totals)
template <int EPT>
__device__ __forceinline__ void count_below3(const KeySet<EPT> &ks, unsigned c1, unsigned c2, unsigned c3, unsigned *xch, int round,
                                             unsigned &n1, unsigned &n2, unsigned &n3)
{
    unsigned a, b;                               // a = n1 | n2 << 16 (each <= 4096 over the workgroup), b = n3
    const int wave = threadIdx.x >> 6;
    bool writer;
    if constexpr (EPT >= 16) {
        // Per-lane counts on the vector unit (a compare and an add-with-carry per key and candidate), one wave reduction per round:
        // the mask counts below cost two SCALAR instructions per key and candidate (~100 per round and wave; a wave issues one per
        // ~8 cycles).  1024 chunks x 4000 samples: 45 -> 40 us (of which 7 are the loads and stores); with eight keys per thread
        // the reduction costs more than it saves (25 -> 27 us), so those keep the masks.
        unsigned a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            asm("v_cmp_gt_u32_e32 vcc, %3, %6\n\t"
                "v_addc_co_u32_e32 %0, vcc, 0, %0, vcc\n\t"
                "v_cmp_gt_u32_e32 vcc, %4, %6\n\t"
                "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                "v_cmp_gt_u32_e32 vcc, %5, %6\n\t"
                "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc"
                : "+v"(a1), "+v"(a2), "+v"(a3)
                : "v"(c1), "v"(c2), "v"(c3), "v"(ks.k[i])
                : "vcc");
        }
        a = a1 | (a2 << 16);
        b = a3;
        // wave totals in lane 63: four rotations within the rows of 16 lanes, then row 0 -> 1, 2 -> 3 and rows 0-1 -> 2-3 (DPP, no LDS)
#define SLK_DPP_ADD(v, ctrl, rows) v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, false)
        SLK_DPP_ADD(a, 0x121, 0xf); SLK_DPP_ADD(b, 0x121, 0xf);          // row_ror:1
        SLK_DPP_ADD(a, 0x122, 0xf); SLK_DPP_ADD(b, 0x122, 0xf);          // row_ror:2
        SLK_DPP_ADD(a, 0x124, 0xf); SLK_DPP_ADD(b, 0x124, 0xf);          // row_ror:4
        SLK_DPP_ADD(a, 0x128, 0xf); SLK_DPP_ADD(b, 0x128, 0xf);          // row_ror:8
        SLK_DPP_ADD(a, 0x142, 0xa); SLK_DPP_ADD(b, 0x142, 0xa);          // row_bcast:15 into rows 1 and 3
        SLK_DPP_ADD(a, 0x143, 0xc); SLK_DPP_ADD(b, 0x143, 0xc);          // row_bcast:31 into rows 2 and 3
#undef SLK_DPP_ADD
        writer = (threadIdx.x & 63) == 63;
    } else {
        // wave totals straight from the compare masks (v_cmp writes a 64-bit lane mask, s_bcnt1 counts it): no cross-lane traffic
        a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            a += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ks.k[i] < c1)) +
                 ((unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ks.k[i] < c2)) << 16);
            b += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ks.k[i] < c3));
        }
        writer = (threadIdx.x & 63) == 0;
    }
    // four waves exchange through a slot pair that alternates between rounds: one barrier per round is enough
    unsigned *slot = xch + (round & 1) * 8;
    if (writer) { slot[2 * wave] = a; slot[2 * wave + 1] = b; }
    __syncthreads();
    a = slot[0] + slot[2] + slot[4] + slot[6];
    b = slot[1] + slot[3] + slot[5] + slot[7];
    n1 = a & 0xffffu;
    n2 = a >> 16;
    n3 = b;
}

// keys of rank r and r+1 (0-based) of the workgroup's keys; r + 1 < number of real keys when `need_next`.
// After eight rounds (the upper 16 bits of the answer) the keys that share them -- a percent or two of a
// chunk of signal -- are collected in LDS (`cand`, SEL_CAP + 2 words) and wave 0 finishes the lower 16 bits on them alone, four keys
// per lane, no barrier per round; a chunk with more than SEL_CAP such keys (constant or coarsely quantised signal) goes on as before.
#define SEL_CAP 256
template <int EPT>
__device__ void select_pair(const KeySet<EPT> &ks, unsigned r, bool need_next, unsigned *xch, unsigned *cand, unsigned &key_r,
                            unsigned &key_next)
{
    unsigned prefix = 0, below = 0;              // below = keys < prefix
    constexpr int SPLIT = 8;
    auto full_round = [&](int round) {
        const int sh = 30 - 2 * round;
        unsigned n1, n2, n3;
        count_below3<EPT>(ks, prefix | (1u << sh), prefix | (2u << sh), prefix | (3u << sh), xch, round, n1, n2, n3);
        const unsigned d = n3 <= r ? 3u : (n2 <= r ? 2u : (n1 <= r ? 1u : 0u));
        below = d == 3u ? n3 : (d == 2u ? n2 : (d == 1u ? n1 : below));
        prefix |= d << sh;
    };
#pragma unroll 1
    for (int round = 0; round < SPLIT; round++) full_round(round);
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (threadIdx.x == 0) cand[SEL_CAP] = 0u;
        __syncthreads();
        const unsigned top = prefix >> 16;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            if ((ks.k[i] >> 16) == top) {
                const unsigned pos = atomicAdd(&cand[SEL_CAP], 1u);
                if (pos < SEL_CAP) cand[pos] = ks.k[i];
            }
        }
        __syncthreads();
        const unsigned cnt = cand[SEL_CAP];
        if (cnt <= SEL_CAP) {                    // (uniform)
            if (wave == 0) {
                unsigned ck[SEL_CAP / 64];
#pragma unroll
                for (int j = 0; j < SEL_CAP / 64; j++) ck[j] = (unsigned)(lane + 64 * j) < cnt ? cand[lane + 64 * j] : 0xffffffffu;
                const unsigned rr = r - below;   // the rank among them: every key below `prefix` has smaller upper bits
#pragma unroll 1
                for (int round = SPLIT; round < 16; round++) {
                    const int sh = 30 - 2 * round;
                    const unsigned c1 = prefix | (1u << sh), c2 = prefix | (2u << sh), c3 = prefix | (3u << sh);
                    unsigned n1 = 0, n2 = 0, n3 = 0;
#pragma unroll
                    for (int j = 0; j < SEL_CAP / 64; j++) {
                        n1 += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ck[j] < c1));
                        n2 += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ck[j] < c2));
                        n3 += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ck[j] < c3));
                    }
                    const unsigned d = n3 <= rr ? 3u : (n2 <= rr ? 2u : (n1 <= rr ? 1u : 0u));
                    prefix |= d << sh;
                }
                if (lane == 0) cand[SEL_CAP + 1] = prefix;
            }
            __syncthreads();
            prefix = cand[SEL_CAP + 1];
        } else {
#pragma unroll 1
            for (int round = SPLIT; round < 16; round++) full_round(round);
        }
    }
    key_r = prefix;
    key_next = prefix;
    if (need_next) {
        // how many keys are <= key_r, and the smallest key above it
        unsigned le = 0, mn = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < EPT; i++) {
            le += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(ks.k[i] <= prefix));
            mn = min(mn, ks.k[i] > prefix ? ks.k[i] : 0xffffffffu);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mn = min(mn, (unsigned)__shfl_xor(mn, o));
        unsigned *slot = xch + 16;
        const int wave = threadIdx.x >> 6;
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { slot[2 * wave] = le; slot[2 * wave + 1] = mn; }
        __syncthreads();
        le = slot[0] + slot[2] + slot[4] + slot[6];
        mn = min(min(slot[1], slot[3]), min(slot[5], slot[7]));
        if (le < r + 2) key_next = mn;            // rank r+1 lies beyond the run of keys equal to key_r
    }
    __syncthreads();                              // the exchange slots are reused by the next selection
}

template <int EPT>
__global__ void __launch_bounds__(256) med_mad_select_kernel(const float *__restrict__ signal, int chunk_len,
                                                             float *__restrict__ out, long out_chunk_stride,
                                                             long out_sample_stride, float *__restrict__ med_out,
                                                             float *__restrict__ mad_out)
{
    __shared__ unsigned xch[24];
    __shared__ unsigned cand[SEL_CAP + 2];
    const int c = blockIdx.x, tid = threadIdx.x, n = chunk_len;
    const float *sig = signal + (size_t)c * chunk_len;
    float x[EPT];
    KeySet<EPT> ks;
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = tid + 256 * i;
        x[i] = idx < n ? sig[idx] : 0.0f;
        ks.k[i] = idx < n ? f2key(x[i]) : 0xffffffffu;            // padding sorts after every sample
    }
    const bool even = (n & 1) == 0;
    const unsigned r = even ? (unsigned)(n >> 1) - 1 : (unsigned)(n >> 1);
    unsigned ka, kb;
    select_pair<EPT>(ks, r, even, xch, cand, ka, kb);
    const float med = even ? (key2f(ka) + key2f(kb)) / 2.0f : key2f(ka);   // numpy: mean of the two middle samples in float32
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = tid + 256 * i;
        ks.k[i] = idx < n ? f2key(fabsf(x[i] - med)) : 0xffffffffu;
    }
    select_pair<EPT>(ks, r, even, xch, cand, ka, kb);
    const float dm = even ? (key2f(ka) + key2f(kb)) / 2.0f : key2f(ka);
    const float mad = 1.4826f * dm;
    float *o = out + (size_t)c * out_chunk_stride;
#pragma unroll
    for (int i = 0; i < EPT; i++) {
        const int idx = tid + 256 * i;
        if (idx < n) o[(size_t)idx * out_sample_stride] = (x[i] - med) / mad;
    }
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
    if (nchunk == 0) return SLK_OK;
    if (chunk_len > 32768) {      // whole reads: radix selection, any length
        hipLaunchKernelGGL(med_mad_radix_kernel, dim3(nchunk), dim3(1024), 0, slk_stream(stream), signal, chunk_len, out,
                           out_chunk_stride, out_sample_stride, med_out, mad_out);
        return slk_launch_status();
    }
    if (chunk_len >= 1024 && chunk_len <= 4096) { // the basecaller's chunks: selection on keys held in registers
        if (chunk_len <= 2048)
            hipLaunchKernelGGL(med_mad_select_kernel<8>, dim3(nchunk), dim3(256), 0, slk_stream(stream), signal, chunk_len, out,
                               out_chunk_stride, out_sample_stride, med_out, mad_out);
        else
            hipLaunchKernelGGL(med_mad_select_kernel<16>, dim3(nchunk), dim3(256), 0, slk_stream(stream), signal, chunk_len, out,
                               out_chunk_stride, out_sample_stride, med_out, mad_out);
        return slk_launch_status();
    }
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

// Per-window standard deviation (population form, numpy's ndarray.std): the local-variance measure of
// batch.trim_open_pore(var_method='std'), sloika/batch.py:210-211.  One wave per window, sums in double.
__global__ void __launch_bounds__(256) window_std_kernel(const float *__restrict__ signal, int nwin, int win,
                                                          float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nwin) return;
    const float *p = signal + (size_t)w * win;
    double s = 0.0;
    for (int i = lane; i < win; i += 64) s += (double)p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mean = s / win;
    double q = 0.0;
    for (int i = lane; i < win; i += 64) { const double d = (double)p[i] - mean; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) out[w] = (float)sqrt(q / win);
}

// ---- whole reads: the read set lies in one device buffer (batch.upload_reads_windowed); a batch of reads of similar length becomes
// ---- a zero-padded [B][ld] matrix (what 4096 device-to-device copies did before), and reads with samples that are not finite are
// ---- found before they can poison a batch (basecall.py:103-115: the reference's worker skips a read that fails and goes on)
__global__ void __launch_bounds__(256) pack_reads_kernel(const float *__restrict__ src, const long long *__restrict__ start,
                                                         const int *__restrict__ len, float *__restrict__ dst, long ld)
{
    const int b = blockIdx.y;
    const long j0 = (long)blockIdx.x * 1024 + threadIdx.x;
    const int n = len[b];
    const float *s = src + start[b];
    float *d = dst + (size_t)b * ld;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const long j = j0 + 256 * k;
        if (j < ld) d[j] = j < n ? s[j] : 0.0f;
    }
}

__global__ void __launch_bounds__(256) reads_nonfinite_kernel(const float *__restrict__ src, const long long *__restrict__ start,
                                                              const int *__restrict__ len, int *__restrict__ flags)
{
    const int b = blockIdx.y;
    const long j0 = (long)blockIdx.x * 4096 + threadIdx.x;
    const int n = len[b];
    const float *s = src + start[b];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const long j = j0 + 256 * k;
        if (j < n) bad |= (__float_as_uint(s[j]) & 0x7f800000u) == 0x7f800000u;      // exponent all ones: infinity or NaN
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&flags[b], 1);
}

extern "C" int slk_pack_reads_f32(const float *src, const int64_t *start, const int32_t *len, int nread, float *dst, long ld,
                                  slk_stream_t stream)
{
    if (!src || !start || !len || !dst || nread < 0 || ld < 1) return SLK_ERR_INVALID_ARG;
    if (nread == 0) return SLK_OK;
    if (nread > 65535) return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pack_reads_kernel, dim3((unsigned)((ld + 1023) / 1024), nread), dim3(256), 0, slk_stream(stream), src,
                       reinterpret_cast<const long long *>(start), len, dst, ld);
    return slk_launch_status();
}

extern "C" int slk_reads_nonfinite_f32(const float *src, const int64_t *start, const int32_t *len, int nread, int max_len,
                                       int32_t *flags, slk_stream_t stream)
{
    if (!src || !start || !len || !flags || nread < 0 || max_len < 0) return SLK_ERR_INVALID_ARG;
    if (nread == 0 || max_len == 0) return SLK_OK;
    if (nread > 65535) return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(reads_nonfinite_kernel, dim3((max_len + 4095) / 4096, nread), dim3(256), 0, slk_stream(stream), src,
                       reinterpret_cast<const long long *>(start), len, flags);
    return slk_launch_status();
}

// batch.trim_open_pore with max_op_fraction 0 (the CLI's default, bin/basecall_network.py:71) for every read of an uploaded set, on the
// device: np.percentile(spread, 0) is the minimum, the read runs from its first to its last window livelier than that (batch.py:213-220),
// then util.trim_array takes `trim0` / `trim1` samples off the ends (basecall.py:111-112).  One wave per read (a read has a few hundred to
// ~1200 windows).  flags[r]: bit 0 on entry = the read holds a sample that is not finite (slk_reads_nonfinite_f32); on exit bit 1 = no
// whole window or no window livelier than the minimum (the reference's function fails on such a read), bit 2 = nothing left after
// trimming.  A read with any bit set gets length 0.
__global__ void __launch_bounds__(64) open_pore_trim_kernel(const float *__restrict__ spread, const long long *__restrict__ first_win,
                                                            const int *__restrict__ nwin, const long long *__restrict__ first_sample,
                                                            int nread, int window, int trim0, int trim1, long long *__restrict__ start,
                                                            int *__restrict__ len, int *__restrict__ flags)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= nread) return;
    const int n = nwin[r];
    const float *sp = spread + first_win[r];
    float mn = __builtin_inff();
    for (int w = lane; w < n; w += 64) mn = fminf(mn, sp[w]);
#pragma unroll
    for (int o = 32; o; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
    int first = INT_MAX, last = -1;
    for (int w = lane; w < n; w += 64)
        if (sp[w] > mn) {
            first = min(first, w);
            last = max(last, w);
        }
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        first = min(first, __shfl_xor(first, o));
        last = max(last, __shfl_xor(last, o));
    }
    if (lane == 0) {
        int f = flags[r] & 1;
        long long lo = 0, hi = 0;
        if (n < 1 || last < 0) f |= 2;
        else {
            lo = (long long)first * window + trim0;
            hi = ((long long)last + 1) * window - trim1;
            if (hi - lo < 1) f |= 4;
        }
        const bool ok = f == 0;
        start[r] = first_sample[r] + (ok ? lo : 0);
        len[r] = ok ? (int)(hi - lo) : 0;
        flags[r] = f;
    }
}

extern "C" int slk_open_pore_trim_f32(const float *spread, const int64_t *first_win, const int32_t *nwin, const int64_t *first_sample,
                                      int nread, int window, int trim0, int trim1, int64_t *start, int32_t *len, int32_t *flags,
                                      slk_stream_t stream)
{
    if (!spread || !first_win || !nwin || !first_sample || !start || !len || !flags || nread < 0 || window < 1 || trim0 < 0 || trim1 < 0)
        return SLK_ERR_INVALID_ARG;
    if (nread == 0) return SLK_OK;
    hipLaunchKernelGGL(open_pore_trim_kernel, dim3(nread), dim3(64), 0, slk_stream(stream), spread,
                       reinterpret_cast<const long long *>(first_win), nwin, reinterpret_cast<const long long *>(first_sample), nread,
                       window, trim0, trim1, reinterpret_cast<long long *>(start), len, flags);
    return slk_launch_status();
}

extern "C" int slk_window_std_f32(const float *signal, int nwin, int win, float *out, slk_stream_t stream)
{
    if (!signal || !out || nwin < 0 || win < 1) return SLK_ERR_INVALID_ARG;
    if (nwin == 0) return SLK_OK;
    hipLaunchKernelGGL(window_std_kernel, dim3((nwin + 3) / 4), dim3(256), 0, slk_stream(stream), signal, nwin, win, out);
    return slk_launch_status();
}

// ------------------------------------------------------------------------------------------------------
// conv1d.  threadIdx.x <-> output feature (so stores are fully coalesced), threadIdx.y <-> one of PPB (to, b)
// positions handled per block iteration; the filter is staged transposed in LDS (Wt[c][k][o]) so lanes read
// consecutive words; the Cin*winlen input taps of a position are the same address for every lane of a row of the
// block (a broadcast load).  All index arithmetic is 32-bit (one divide per position, none per output element).
// ------------------------------------------------------------------------------------------------------
template <bool W_IN_LDS>
__global__ void __launch_bounds__(512) conv1d_kernel(const float *__restrict__ x, long xs_t, long xs_b,
                                                     const float *__restrict__ W, const float *__restrict__ bias,
                                                     float *__restrict__ y, int T, int B, int Cin, int Cout,
                                                     int winlen, int stride, int pad_l, int Tout, int act)
{
    extern __shared__ float wt[];
    const int ckn = Cin * winlen;
    const int nthreads = blockDim.x * blockDim.y, tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (W_IN_LDS) {
        for (int i = tid; i < ckn * Cout; i += nthreads) {
            int o = i / ckn, ck = i - o * ckn;
            wt[ck * Cout + o] = W[i];
        }
        __syncthreads();
    }
    const unsigned npos = (unsigned)Tout * (unsigned)B;
    for (unsigned p = blockIdx.x * blockDim.y + threadIdx.y; p < npos; p += gridDim.x * blockDim.y) {
        const int to = (int)(p / (unsigned)B), b = (int)(p - (unsigned)to * (unsigned)B);
        const float *xb = x + (size_t)b * xs_b;
        float *yp = y + (size_t)p * Cout;
        const int t0 = to * stride - pad_l;
        for (int o = threadIdx.x; o < Cout; o += blockDim.x) {
            float s = 0.0f;
            for (int c = 0; c < Cin; c++) {
                for (int k = 0; k < winlen; k++) {
                    const int ti = t0 + k;
                    if (ti < 0 || ti >= T) continue;
                    const float xv = xb[(size_t)ti * xs_t + c];
                    const float wv = W_IN_LDS ? wt[(c * winlen + k) * Cout + o] : W[((size_t)o * Cin + c) * winlen + k];
                    s = fmaf(xv, wv, s);
                }
            }
            if (bias) s += bias[o];
            yp[o] = slk_act(act, s);
        }
    }
}

// Single-input-channel specialisation (every raw-signal front end: Cin = 1, winlen = 11, stride 2 or 5).
// One WAVE handles a run of consecutive output steps of ONE chunk: the 64 lanes load 128 consecutive input samples
// with two coalesced loads, every tap is then handed to all lanes with v_readlane (a scalar operand of the FMA), and
// the lanes -- one output feature each, filter taps in registers -- emit up to 24 output rows from those two loads.
// (The generic kernel issues winlen broadcast loads per output row and is latency bound: 0.8 ms vs HBM time 0.06 ms.)
template <int WMAX>
__global__ void __launch_bounds__(256) conv1d_cin1_kernel(const float *__restrict__ x, long xs_t, long xs_b,
                                                          const float *__restrict__ W, const float *__restrict__ bias,
                                                          float *__restrict__ y, int T, int B, int Cout, int winlen,
                                                          int stride, int pad_l, int Tout, int act, int npos_run)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned runs_per_chunk = (unsigned)((Tout + npos_run - 1) / npos_run);
    const unsigned nrun = runs_per_chunk * (unsigned)B, nwave = gridDim.x * 4u;
    for (int o0 = 0; o0 < Cout; o0 += 64) {
        const int o = o0 + lane;
        const bool ok = o < Cout;
        float w[WMAX];
#pragma unroll
        for (int k = 0; k < WMAX; k++) w[k] = (ok && k < winlen) ? W[(size_t)o * winlen + k] : 0.0f;
        const float bv = (bias && ok) ? bias[o] : 0.0f;
        for (unsigned run = blockIdx.x * 4u + wave; run < nrun; run += nwave) {
            // consecutive runs walk the batch first so that neighbouring waves write neighbouring rows of y
            const int b = (int)(run % (unsigned)B), to0 = (int)(run / (unsigned)B) * npos_run;
            const int t0 = to0 * stride - pad_l;                   // input sample of tap 0 of the first position
            const float *xb = x + (size_t)b * xs_b;
            const int ta = t0 + lane, tb2 = t0 + 64 + lane;
            const float v0 = (ta >= 0 && ta < T) ? xb[(size_t)ta * xs_t] : 0.0f;
            const float v1 = (tb2 >= 0 && tb2 < T) ? xb[(size_t)tb2 * xs_t] : 0.0f;
            const int npos = min(npos_run, Tout - to0);
            for (int j = 0; j < npos; j++) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < WMAX; k++) {
                    if (k < winlen) {
                        const int i = j * stride + k;              // wave-uniform sample index within the 128 loaded
                        const float xv = i < 64 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v0), i))
                                                : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v1), i - 64));
                        s = fmaf(xv, w[k], s);
                    }
                }
                if (ok) y[((size_t)(to0 + j) * B + b) * Cout + o] = slk_act(act, s + bv);
            }
        }
    }
}

// Cin = 1, Cout a multiple of 4 (the shipped raw models: 32/64/96/128 features): the lane owns FOUR consecutive output
// features of one position, 64 / (Cout/4) positions are computed side by side, and a row segment leaves as one 16-byte
// store (the one-feature-per-lane kernel above issues a dword store per lane and row: store-issue bound at ~1 TB/s).
// The 128 samples a run needs are parked in LDS (one 512-byte slot per wave); every tap is a broadcast ds_read.
// WL > 0: the window length at compile time (WMAX = WL): no branch per tap (the run-time form tests `k < winlen` sixteen times per
// position group).  The kernel is bound by the instructions it issues per position, not by its 331 MB: with the window length
// fixed, elu without exec-mask updates (common.h) and the samples of the next run requested a run ahead, 1024 chunks x 4000 samples
// -> 96 features take 82 us instead of 118 (64 features: 43 instead of 76, 4.9 TB/s).  Six features per lane (Cout = 96 then
// fills all 64 lanes: four positions of sixteen lanes instead of two of 24) measured 84: its second, 8-byte store costs what the
// idle lanes do.
template <int WMAX, int ACT, int WL>
__global__ void __launch_bounds__(256) conv1d_cin1_vec4_kernel(const float *__restrict__ x, long xs_t, long xs_b,
                                                               const float *__restrict__ W, const float *__restrict__ bias,
                                                               float *__restrict__ y, int T, int B, int Cout, int winlen,
                                                               int stride, int pad_l, int Tout, int act, int npos_run)
{
    __shared__ float xs[4][128];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int FPL = 4;                                         // features per lane
    const int FQ = Cout / FPL, PP = 64 / FQ;                       // feature groups, positions side by side
    const int fq = lane % FQ, pp = lane / FQ;
    const bool lane_ok = pp < PP;
    auto feat = [&](int i) { return 4 * fq + i; };
    float w[FPL][WMAX];
#pragma unroll
    for (int i = 0; i < FPL; i++) {
#pragma unroll
        for (int k = 0; k < WMAX; k++) w[i][k] = (WL > 0 || k < winlen) ? W[(size_t)feat(i) * winlen + k] : 0.0f;
    }
    float bv[FPL];
#pragma unroll
    for (int i = 0; i < FPL; i++) bv[i] = bias ? bias[feat(i)] : 0.0f;
    const unsigned runs_per_chunk = (unsigned)((Tout + npos_run - 1) / npos_run);
    const unsigned nrun = runs_per_chunk * (unsigned)B, nwave = gridDim.x * 4u;
    float *xw = xs[wave];
    const size_t rowpitch = (size_t)B * Cout;                      // floats between two output steps of one chunk
    // the 128 samples of a run are requested one run ahead (a run is ~2 us of arithmetic, about the latency of the load)
    auto fetch = [&](unsigned run, float &v0, float &v1) {
        const int b = (int)(run % (unsigned)B), to0 = (int)(run / (unsigned)B) * npos_run;
        const int t0 = to0 * stride - pad_l;                       // input sample of tap 0 of the first position
        const float *xb = x + (size_t)b * xs_b;
        const int ta = t0 + lane, tb2 = t0 + 64 + lane;
        v0 = (ta >= 0 && ta < T) ? xb[(size_t)ta * xs_t] : 0.0f;
        v1 = (tb2 >= 0 && tb2 < T) ? xb[(size_t)tb2 * xs_t] : 0.0f;
    };
    float v0 = 0.0f, v1 = 0.0f;
    if (blockIdx.x * 4u + wave < nrun) fetch(blockIdx.x * 4u + wave, v0, v1);
    for (unsigned run = blockIdx.x * 4u + wave; run < nrun; run += nwave) {
        // consecutive runs walk the batch first so that neighbouring waves write neighbouring rows of y
        const int b = (int)(run % (unsigned)B), to0 = (int)(run / (unsigned)B) * npos_run;
        xw[lane] = v0;                                             // same wave writes and reads: in-order LDS, no barrier
        xw[64 + lane] = v1;
        if (run + nwave < nrun) fetch(run + nwave, v0, v1);
        const int npos = min(npos_run, Tout - to0);
        float *yp = y + ((size_t)(to0 + pp) * B + b) * Cout;                   // the row of this lane's first position
        for (int j0 = 0; j0 < npos; j0 += PP, yp += (size_t)PP * rowpitch) {
            const int j = j0 + pp;
            const float *xp = xw + (lane_ok && j < npos ? j : 0) * stride;
            float s[FPL];
#pragma unroll
            for (int i = 0; i < FPL; i++) s[i] = 0.0f;
#pragma unroll
            for (int k = 0; k < WMAX; k++) {
                if (WL > 0 || k < winlen) {
                    const float xv = xp[k];
#pragma unroll
                    for (int i = 0; i < FPL; i++) s[i] = fmaf(xv, w[i][k], s[i]);
                }
            }
            // taps first, bias last, as conv.py:107-110 (same order as the one-feature-per-lane kernel)
            if (lane_ok && j < npos) {
#pragma unroll
                for (int i = 0; i < FPL; i++) {
                    s[i] += bv[i];
                    s[i] = ACT >= 0 ? slk_act_t<ACT>(s[i]) : slk_act(act, s[i]);
                }
                *reinterpret_cast<float4 *>(yp + 4 * fq) = make_float4(s[0], s[1], s[2], s[3]);
            }
        }
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
    if ((size_t)Tout * B > 0x7fffffffu) return SLK_ERR_UNSUPPORTED;
    if (Cin == 1 && winlen <= 16 && stride <= 16) {
        int npos_run = (128 - winlen) / stride + 1;               // output steps covered by 128 loaded samples
        size_t nrun = (size_t)((Tout + npos_run - 1) / npos_run) * B, blocks = (nrun + 3) / 4;
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (Cout % 4 == 0 && Cout <= 256 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
            const dim3 grid((unsigned)blocks), block(256);
            hipStream_t st = slk_stream(stream);
#define CONV_VEC4_(A, WM, WL)                                                                                             \
    hipLaunchKernelGGL((conv1d_cin1_vec4_kernel<WM, A, WL>), grid, block, 0, st, x, x_t_stride, x_b_stride, W, bias, y, T, B, \
                       Cout, winlen, stride, pad_l, Tout, act, npos_run)
#define CONV_VEC4(A)                                                                                                    \
    do {                                                                                                                \
        if (winlen == 11) CONV_VEC4_(A, 11, 11); /* the shipped raw models */                                           \
        else CONV_VEC4_(A, 16, 0);                                                                                      \
    } while (0)
            switch (act) {
            case SLK_ACT_TANH: CONV_VEC4(SLK_ACT_TANH); break;
            case SLK_ACT_ELU: CONV_VEC4(SLK_ACT_ELU); break;
            case SLK_ACT_RELU: CONV_VEC4(SLK_ACT_RELU); break;
            case SLK_ACT_LINEAR: CONV_VEC4(SLK_ACT_LINEAR); break;
            default: CONV_VEC4(-1); break;
            }
#undef CONV_VEC4_
#undef CONV_VEC4
            return slk_launch_status();
        }
        hipLaunchKernelGGL(conv1d_cin1_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, slk_stream(stream), x, x_t_stride,
                           x_b_stride, W, bias, y, T, B, Cout, winlen, stride, pad_l, Tout, act, npos_run);
        return slk_launch_status();
    }
    // block = (features rounded up to a multiple of 32, up to 128) x (positions): 256..512 threads
    int bx = Cout >= 128 ? 128 : ((Cout + 31) / 32) * 32;
    int by = 512 / bx;
    size_t npos = (size_t)Tout * B;
    size_t blocks = (npos + by - 1) / by;
    if (blocks > 256 * 8) blocks = 256 * 8;
    size_t wbytes = (size_t)Cin * winlen * Cout * sizeof(float);
    if (wbytes <= 64 * 1024)
        hipLaunchKernelGGL(conv1d_kernel<true>, dim3((unsigned)blocks), dim3(bx, by), wbytes, slk_stream(stream), x,
                           x_t_stride, x_b_stride, W, bias, y, T, B, Cin, Cout, winlen, stride, pad_l, Tout, act);
    else
        hipLaunchKernelGGL(conv1d_kernel<false>, dim3((unsigned)blocks), dim3(bx, by), 0, slk_stream(stream), x,
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
