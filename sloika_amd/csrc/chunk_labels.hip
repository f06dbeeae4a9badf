// chunk_labels.hip -- the label half of raw_chunkify (sloika/tools/chunkify_raw.py:164-210) on gfx950.
//
//   labels_from_mapping_table    chunkify_raw.py:117-136   k-mer text of every mapped block -> state + 1
//   plain labels                 chunkify_raw.py:194-204   label of the last MOVE at or before every downsample_factor-th
//                                                          sample of a chunk; 0 where the block repeats its predecessor
//   interpolated labels          chunkify_raw.py:187-193, 86-114   np.interp of (block mid-time -> reference position),
//                                                          rounded, looked up in the reference string
//
// All of it is integer / index work (plus one float64 interpolation evaluated exactly like numpy's C loop), HBM- and
// latency-bound: one thread per output label, a binary search over the block starts per label.  The reference builds an
// index over EVERY sample (`idx = np.zeros(ub)`, fill_zeros_with_prev) and then keeps every downsample_factor-th entry;
// here only the kept entries are ever computed.
#include "common.h"

struct SlkAlphabet {
    unsigned char letter[8];
};

__device__ __forceinline__ int letter_rank(const SlkAlphabet &a, int nbase, unsigned char c)
{
    int r = -1;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (i < nbase && a.letter[i] == c) r = i;
    return r;
}

// state (index in bio.all_kmers order, first letter most significant) of `klen` letters at `p`; -1 for a foreign letter.
__device__ __forceinline__ int kmer_state(const unsigned char *p, int klen, const SlkAlphabet &a, int nbase)
{
    int s = 0;
    bool ok = true;
    for (int i = 0; i < klen; i++) {
        const int r = letter_rank(a, nbase, p[i]);
        ok = ok && r >= 0;
        s = s * nbase + (r < 0 ? 0 : r);
    }
    return ok ? s : -1;
}

// chunkify_raw.py:117-136: the middle `klen` letters of every `old_klen`-letter k-mer, as state + index_from.
__global__ void __launch_bounds__(256) kmer_labels_kernel(const unsigned char *__restrict__ kmers, long long n, int stride,
                                                          int offset, int klen, SlkAlphabet alpha, int nbase, int index_from,
                                                          int32_t *__restrict__ out, int *__restrict__ status)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = kmer_state(kmers + i * stride + offset, klen, alpha, nbase);
    out[i] = s < 0 ? -1 : s + index_from;
    if (s < 0) atomicOr(status, 1);
}

// last_moved[i] = index (within the read) of the last block j <= i with move[j] > 0, or -1 (what the reference gets from
// idx[starts] = arange + 1 followed by fill_zeros_with_prev, chunkify_raw.py:196-201, without the per-sample array).
// One workgroup per read, tiles of 1024 blocks with a carry.
__global__ void __launch_bounds__(1024) last_moved_kernel(const int64_t *__restrict__ move, const int64_t *__restrict__ ev_off,
                                                          int32_t *__restrict__ last_moved)
{
    __shared__ int wave_last[16];
    __shared__ int carry_s;
    const int r = blockIdx.x;
    const int64_t e0 = ev_off[r];
    const int n = (int)(ev_off[r + 1] - e0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = -1;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        int v = (i < n && move[e0 + i] > 0) ? i : -1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(v, d);
            if (lane >= d) v = max(v, o);
        }
        if (lane == 63) wave_last[wave] = v;
        __syncthreads();
        int pre = carry_s;
        for (int w = 0; w < wave; w++) pre = max(pre, wave_last[w]);
        v = max(v, pre);
        if (i < n) last_moved[e0 + i] = v;
        __syncthreads();
        if (tid == 1023) carry_s = v;
        __syncthreads();
    }
}

// index of the last block whose start is <= s, or -1 (starts are non-decreasing: the table is registered).
__device__ __forceinline__ int last_block_at(const int64_t *__restrict__ start, int n, int64_t s)
{
    int lo = 0, hi = n;                                   // first index with start > s
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= s) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// chunkify_raw.py:194-204.  grid = (label tiles, reads).
__global__ void __launch_bounds__(256) chunk_labels_kernel(const int64_t *__restrict__ start,
                                                           const int32_t *__restrict__ event_label,
                                                           const int32_t *__restrict__ last_moved,
                                                           const int64_t *__restrict__ ev_off,
                                                           const int64_t *__restrict__ nchunk,
                                                           const int64_t *__restrict__ lab_off, int chunk_len, int downsample,
                                                           int nblk, int32_t *__restrict__ labels)
{
    const int r = blockIdx.y;
    const int64_t e0 = ev_off[r];
    const int n = (int)(ev_off[r + 1] - e0);
    const int64_t total = nchunk[r] * nblk;
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total) return;
    const int64_t c = o / nblk;
    const int j = (int)(o - c * nblk);
    const int64_t s = c * chunk_len + (int64_t)j * downsample;
    const int e = last_block_at(start + e0, n, s);
    const int lm = e < 0 ? -1 : last_moved[e0 + e];
    int lab = lm < 0 ? 0 : event_label[e0 + lm];          // np.concatenate([[0], labels])[idx]   :204
    if (j > 0) {                                          // replace_repeats_with_zero, per chunk row :139-142, 202
        const int ep = last_block_at(start + e0, n, s - downsample);
        const int lp = ep < 0 ? -1 : last_moved[e0 + ep];
        if (lp == lm) lab = 0;
    }
    labels[lab_off[r] + o] = lab;
}

// np.interp(t, xp, fp) for one t, the way numpy's arr_interp evaluates it (numpy/_core/src/multiarray/compiled_base.c):
// j = last index with xp[j] <= t; t left of xp[0] -> fp[0], right of xp[n-1] -> fp[n-1], t == xp[j] -> fp[j], otherwise
// slope * (t - xp[j]) + fp[j] with slope = (fp[j+1] - fp[j]) / (xp[j+1] - xp[j]); float64, no contraction (build flags).
//   xp[i] = start[i] + 0.5 * length[i]                                   chunkify_raw.py:95
//   fp[i] = seq_pos[i] + 0.5 * map_k - ref_start            ('+')        chunkify_raw.py:98-99
//         = (ref_stop - seq_pos[i]) + 0.5 * map_k           ('-')        chunkify_raw.py:100-101
struct InterpTable {
    const int64_t *start, *length, *seq_pos;
    int n, map_k, forward;
    int64_t anchor;
    __device__ __forceinline__ double xp(int i) const { return (double)start[i] + 0.5 * (double)length[i]; }
    __device__ __forceinline__ double fp(int i) const
    {
        if (forward) return ((double)seq_pos[i] + 0.5 * (double)map_k) - (double)anchor;
        return (double)(anchor - seq_pos[i]) + 0.5 * (double)map_k;
    }
    __device__ __forceinline__ double at(double t) const
    {
        if (n == 1) return fp(0);
        if (t > xp(n - 1)) return fp(n - 1);
        if (t < xp(0)) return fp(0);
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (xp(mid) <= t) lo = mid + 1; else hi = mid;
        }
        const int j = lo - 1;
        if (j == n - 1) return fp(j);
        const double xj = xp(j), yj = fp(j);
        if (xj == t) return yj;
        const double slope = (fp(j + 1) - yj) / (xp(j + 1) - xj);
        return slope * (t - xj) + yj;
    }
    // chunkify_raw.py:102-103: np.around(pos_interp - 0.5 * k + EPS).astype(int); np.around rounds half to even.
    __device__ __forceinline__ int64_t pos(double t, int k) const { return (int64_t)rint((at(t) - 0.5 * (double)k) + 1e-10); }
};

// chunkify_raw.py:187-193 for one read: block o sits at sample o * downsample (or at times[o] when the caller of
// interpolate_pos / interpolate_labels hands its own times).
__global__ void __launch_bounds__(256) chunk_labels_interp_kernel(InterpTable tab, const unsigned char *__restrict__ reference,
                                                                  int64_t ref_len, int klen, SlkAlphabet alpha, int nbase,
                                                                  int64_t nlabel, int downsample,
                                                                  const double *__restrict__ times, int zero_repeats,
                                                                  int32_t *__restrict__ labels, int64_t *__restrict__ pos_out,
                                                                  int *__restrict__ status)
{
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nlabel) return;
    const int64_t p = tab.pos(times ? times[o] : (double)(o * downsample), klen);
    if (pos_out) pos_out[o] = p;
    if (!labels) return;
    int lab;
    if (p < 0 || p + klen > ref_len) {                    // the reference's dictionary lookup fails on a short k-mer
        atomicOr(status, 2);
        lab = -1;
    } else {
        const int s = kmer_state(reference + p, klen, alpha, nbase);
        if (s < 0) atomicOr(status, 1);
        lab = s + 1;                                      // chunkify_raw.py:112
    }
    if (zero_repeats && o > 0 && tab.pos(times ? times[o - 1] : (double)((o - 1) * downsample), klen) == p)
        lab = 0;                                          // np.ediff1d(pos, to_begin=1) == 0  :191
    labels[o] = lab;
}

static bool make_alphabet(const char *alphabet, int nbase, SlkAlphabet *a)
{
    if (!alphabet || nbase < 1 || nbase > 8) return false;
    for (int i = 0; i < 8; i++) a->letter[i] = i < nbase ? (unsigned char)alphabet[i] : 0;
    return true;
}

extern "C" int slk_kmer_labels_i32(const uint8_t *kmers, int64_t n, int old_klen, int klen, const char *alphabet, int nbase,
                                   int index_from, int32_t *labels_out, int *status, slk_stream_t stream)
{
    SlkAlphabet a;
    if (!kmers || !labels_out || !status || n < 1 || klen < 1 || klen > old_klen || !make_alphabet(alphabet, nbase, &a))
        return SLK_ERR_INVALID_ARG;
    const int offset = (old_klen - klen + 1) / 2;         // chunkify_raw.py:130
    hipLaunchKernelGGL(kmer_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, slk_stream(stream), kmers,
                       (long long)n, old_klen, offset, klen, a, nbase, index_from, labels_out, status);
    return slk_launch_status();
}

extern "C" size_t slk_raw_chunk_labels_workspace_bytes(int64_t nevent)
{
    return nevent < 1 ? 0 : sizeof(int32_t) * (size_t)nevent;
}

extern "C" int slk_raw_chunk_labels_i32(const int64_t *start, const int64_t *move, const int32_t *event_label,
                                        const int64_t *ev_off, int nread, const int64_t *nchunk, const int64_t *lab_off,
                                        int64_t max_nchunk, int chunk_len, int downsample, void *workspace,
                                        size_t workspace_bytes, int32_t *labels_out, slk_stream_t stream)
{
    if (!start || !move || !event_label || !ev_off || !nchunk || !lab_off || !labels_out || nread < 1 || max_nchunk < 1 ||
        chunk_len < 1 || downsample < 1)
        return SLK_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < sizeof(int32_t)) return SLK_ERR_WORKSPACE;
    const int nblk = (chunk_len + downsample - 1) / downsample;          // len(range(0, chunk_len, downsample))
    const int64_t tiles = (max_nchunk * nblk + 255) / 256;
    if (tiles > 0x7fffffffLL || nread > 65535) return SLK_ERR_UNSUPPORTED;
    int32_t *last_moved = static_cast<int32_t *>(workspace);
    hipLaunchKernelGGL(last_moved_kernel, dim3(nread), dim3(1024), 0, slk_stream(stream), move, ev_off, last_moved);
    hipLaunchKernelGGL(chunk_labels_kernel, dim3((unsigned)tiles, nread), dim3(256), 0, slk_stream(stream), start, event_label,
                       last_moved, ev_off, nchunk, lab_off, chunk_len, downsample, nblk, labels_out);
    return slk_launch_status();
}

extern "C" int slk_raw_chunk_labels_interp_i32(const int64_t *start, const int64_t *length, const int64_t *seq_pos,
                                               int64_t nevent, int map_klen, int forward, int64_t ref_anchor,
                                               const uint8_t *reference, int64_t ref_len, int klen, const char *alphabet,
                                               int nbase, int64_t nlabel, int downsample, const double *times,
                                               int zero_repeats, int32_t *labels_out, int64_t *pos_out, int *status,
                                               slk_stream_t stream)
{
    SlkAlphabet a;
    if (!start || !length || !seq_pos || (!labels_out && !pos_out) || !status || nevent < 1 || nevent > 0x7fffffffLL ||
        map_klen < 1 || klen < 1 || nlabel < 1 || downsample < 1)
        return SLK_ERR_INVALID_ARG;
    if (labels_out && (!reference || ref_len < klen)) return SLK_ERR_INVALID_ARG;
    if (!make_alphabet(labels_out ? alphabet : "A", labels_out ? nbase : 1, &a)) return SLK_ERR_INVALID_ARG;
    InterpTable tab{start, length, seq_pos, (int)nevent, map_klen, forward ? 1 : 0, ref_anchor};
    hipLaunchKernelGGL(chunk_labels_interp_kernel, dim3((unsigned)((nlabel + 255) / 256)), dim3(256), 0, slk_stream(stream),
                       tab, reference, ref_len, klen, a, nbase, nlabel, downsample, times, zero_repeats, labels_out, pos_out, status);
    return slk_launch_status();
}
