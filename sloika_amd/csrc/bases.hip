// bases.hip -- decoded k-mer state paths -> base sequences on the device (row f1: what bin/basecall_network.py does with the
// Viterbi path before printing it).
//
// Replaces sloika/bio.py:160-179 (max_overlap), :206-225 (reduce_kmers), :228-237 (kmers_to_sequence) for whole batches.
// A k-mer state is the base-`nbase` number of its letters, first letter most significant (bio.py:12-24 orders all_kmers that
// way), so the reference's string comparisons become integer ones:
//     k1[i:] == k2[:-i]   <=>   s1 mod nbase^(k-i) == s2 div nbase^i
// move(s1, s2) = 0 if stays are allowed and s1 == s2, else the SMALLEST i in 1..k-1 with that property, else k; the sequence
// is the k letters of the first state followed, per transition, by the last min(move, k) letters of the next state.
// Output positions are an exclusive prefix sum of the moves: one workgroup per read, 256 transitions per pass.
#include "common.h"

#define BASES_MAX_K 12

__global__ void __launch_bounds__(256) paths_to_bases_kernel(const int32_t *__restrict__ paths, long ld,
                                                             const int32_t *__restrict__ lens, int klen, int nbase,
                                                             int always_move, unsigned long long alphabet,
                                                             uint8_t *__restrict__ out, long cap,
                                                             int32_t *__restrict__ nbases)
{
    __shared__ int wsum[4];
    __shared__ int carry;
    const int b = blockIdx.x, j = threadIdx.x, lane = j & 63, wv = j >> 6;
    const int n = lens[b];
    const int32_t *p = paths + (size_t)b * ld;
    uint8_t *o = out + (size_t)b * cap;
    int nstate = 1;                                            // nbase^klen
    for (int i = 0; i < klen; i++) nstate *= nbase;
    auto letter = [&](int digit) { return (uint8_t)((alphabet >> (8 * digit)) & 0xff); };
    if (n < 1) {
        if (j == 0) nbases[b] = 0;
        return;
    }
    if (j == 0) {                                              // the first k-mer in full (bio.py:216)
        int t = p[0];
        for (int d = klen - 1; d >= 0; d--) { o[d] = letter(t % nbase); t /= nbase; }
    }
    if (j == 0) carry = klen;
    __syncthreads();
    for (int base = 1; base < n; base += 256) {
        const int i = base + j;
        int mv = 0, s2 = 0;
        if (i < n) {
            const int s1 = p[i - 1];
            s2 = p[i];
            mv = klen;
            if (!always_move && s1 == s2) mv = 0;                // bio.py:170-171
            else {
                int hi = nbase, lo = nstate / nbase;               // nbase^sh, nbase^(klen-sh)
                for (int sh = 1; sh < klen; sh++) {                // smallest shift wins (bio.py:173-176)
                    if (mv == klen && s1 % lo == s2 / hi) mv = sh;
                    hi *= nbase;
                    lo /= nbase;
                }
            }
        }
        // exclusive prefix sum of the moves over the workgroup
        int incl = mv;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wv; w++) before += wsum[w];
        const int pos = before + incl - mv;
        if (i < n) {                                            // the last mv letters of s2 (bio.py:219-224)
            int t = s2;
            for (int d = mv - 1; d >= 0; d--) { o[pos + d] = letter(t % nbase); t /= nbase; }
        }
        __syncthreads();
        if (j == 255) carry = before + incl;
        __syncthreads();
    }
    if (j == 0) nbases[b] = carry;
}

// paths:[B][ld] int32 k-mer states (row b valid for lens[b] entries), as slk_viterbi_kmer_* return them;
// alphabet: the nbase letters packed little-endian into 8 bytes ("ACGT" = 0x54474341);
// out:[B][cap] bytes, cap >= klen * max(lens) (a move never exceeds klen); nbases[b] = letters written for read b.
extern "C" int slk_paths_to_bases(const int32_t *paths, long ld, const int32_t *lens, int B, int klen, int nbase,
                                  int always_move, unsigned long long alphabet, uint8_t *out, long cap, int32_t *nbases,
                                  slk_stream_t stream)
{
    if (!paths || !lens || !out || !nbases || B < 0 || klen < 1 || klen > BASES_MAX_K || nbase < 2 || nbase > 8 || cap < klen)
        return SLK_ERR_INVALID_ARG;
    double states = 1.0;
    for (int i = 0; i < klen; i++) states *= nbase;
    if (states > 2147483647.0) return SLK_ERR_INVALID_ARG;
    if (B == 0) return SLK_OK;
    hipLaunchKernelGGL(paths_to_bases_kernel, dim3(B), dim3(256), 0, slk_stream(stream), paths, ld, lens, klen, nbase,
                       always_move, alphabet, out, cap, nbases);
    return slk_launch_status();
}
