// transducer.hip -- remap DP of the reference on gfx950.
//
//   viterbi_helpers.slip_update   sloika/viterbi_helpers.pyx:12-35  (the reference's only native function)
//   transducer.map_to_sequence    sloika/transducer.py:14-73
//
// slip_update is a running max with decay whose float32 result depends on the order of the repeated `- slip` roundings.
// It is still evaluated by 64 lanes at once, bit for bit (slip_scan_wave below): rounded subtraction is monotone, so it
// distributes over max, and every lane can run the recurrence on its own segment of positions; what a segment inherits from
// the positions before it is one decaying chain that only matters until the first local value beats it.
#include "common.h"

// viterbi_helpers.pyx:12-35 by one wave64, every lane active.  x: n scores (LDS or global), outputs n each.
//
// The reference walks k = 0 .. n-3 with one running pair (fs, fp):   if !(fs >= x[k]) { fs = x[k]; fp = k; }  fs -= slip;
// out[k+2] = (fs, fp).  Float32 `- slip` is monotone, so the running value after any prefix is the max over the earlier
// candidates of (x decayed step by step), and an older chain never falls below a younger one it was not beaten by.  Hence:
//   pass 1   lane s runs the recurrence over its own segment of L positions starting from "nothing yet" (-inf) and stores
//            the results.  They are the sequential results from the first position on where the chain inherited from the
//            left has been beaten (had the local winner there exceeded that chain, its candidate would already have beaten
//            the chain when it entered).
//   pass 2   lane s >= 1 takes the LOCAL end pair of segment s-1 as a chain and walks on from the start of its segment, in
//            lockstep with all other lanes, overwriting the outputs with the decaying chain for as long as chain >= x -- into
//            the following segments too, until it is beaten or the array ends.
// Why the plain overwrite is right: if the true chain at a position comes from a candidate in segment g, then that candidate
// really entered, so segment g's local end IS its true end and lane g+1 walks exactly the true chain; every chain that starts
// further right starts from a value that did not beat it, stays <= it, and dies no later.  Such a younger chain reaches a
// given position EARLIER in the lockstep walk than the older one (it starts nearer), so the older chain's store lands last;
// and a true new candidate beats every chain that is still walking, so nothing overwrites pass 1's values behind it.
// Cost: L + (longest surviving chain) dependent steps instead of n; the worst case (one chain owning the array) is n.
// NaN scores are not supported (the reference's own result for them is order-dependent garbage).
struct SlipOutArrays {                                      // the .pyx's two arrays (global memory, positions as np.int)
    float *from_score;
    int64_t *from_pos;
    __device__ __forceinline__ void put(int j, float c, int p) const
    {
        from_score[j] = c;
        from_pos[j] = p;
    }
};
struct SlipOutPairs {                                       // (score, position) side by side: one ds_write_b64 per position
    float2 *pair;
    __device__ __forceinline__ void put(int j, float c, int p) const { pair[j] = make_float2(c, __int_as_float(p)); }
};

#define SLIP_BATCH 16
// Steps U .. SLIP_BATCH-1 of a chain walk: cv[U] is the chain before step U, cv[U+1] after it.  The decaying chain does not
// depend on x, so all cv are computed up front and every comparison is independent; only the narrowing of the execution mask
// is sequential (nested ifs: one s_and_saveexec per step).
template <int U, class OUT>
__device__ __forceinline__ void slip_walk_steps(const float (&xv)[SLIP_BATCH], const float (&cv)[SLIP_BATCH + 1], int k, int p,
                                                const OUT &out, bool &alive)
{
    if constexpr (U < SLIP_BATCH) {
        if (cv[U] >= xv[U]) {
            out.put(k + U + 2, cv[U + 1], p);
            slip_walk_steps<U + 1>(xv, cv, k, p, out, alive);
        } else {
            alive = false;
        }
    }
}

// PADDED: x may be read up to SLIP_BATCH elements past its end (LDS arrays are laid out with that slack); otherwise reads clamp.
template <bool PADDED, class OUT>
__device__ __forceinline__ void slip_scan_wave(const float *x, int n, float slip, const OUT out)
{
    const int lane = threadIdx.x & 63;
    const int nin = n - 2;
    if (lane == 0) {
        out.put(0, -1e38f, 0);
        out.put(1, -1e38f, 0);
    }
    const int L = ((nin + 63) >> 6) | 1;                    // odd: lanes a segment apart never share an LDS bank
    const int k0 = min(lane * L, nin), k1 = min(k0 + L, nin);
    float c = -INFINITY;
    int p = k0;
    for (int k = k0; k < k1; k += 8) {
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = x[PADDED ? k + u : min(k + u, k1 - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (k + u < k1) {
                if (!(c >= xv[u])) { p = k + u; c = xv[u]; }
                c = c - slip;
                out.put(k + u + 2, c, p);
            }
    }
    c = __shfl_up(c, 1);
    p = __shfl_up(p, 1);
    bool alive = lane > 0 && k0 < nin;
    int k = k0;
    while (__ballot(alive)) {
        if (alive) {
            float xv[SLIP_BATCH], cv[SLIP_BATCH + 1];
#pragma unroll
            for (int u = 0; u < SLIP_BATCH; u++) xv[u] = x[PADDED ? k + u : min(k + u, nin - 1)];
            cv[0] = c;
#pragma unroll
            for (int u = 0; u < SLIP_BATCH; u++) cv[u + 1] = cv[u] - slip;
            if (k + SLIP_BATCH > nin) {                     // the batch that reaches the end of the inputs: nothing beyond walks on
#pragma unroll
                for (int u = 0; u < SLIP_BATCH; u++)
                    if (k + u >= nin) xv[u] = INFINITY;
            }
            slip_walk_steps<0>(xv, cv, k, p, out, alive);
            c = cv[SLIP_BATCH];
            k += SLIP_BATCH;
        }
    }
}

__global__ void __launch_bounds__(64) slip_update_kernel(const float *__restrict__ x, int n, float slip,
                                                         float *__restrict__ from_score, int64_t *__restrict__ from_pos)
{
    slip_scan_wave<false>(x, n, slip, SlipOutArrays{from_score, from_pos});
}

extern "C" int slk_slip_update_f32(const float *x, int n, float slip, float *from_score, int64_t *from_pos,
                                   slk_stream_t stream)
{
    if (!x || !from_score || !from_pos || n < 3) return SLK_ERR_INVALID_ARG;   // pyx:24 writes index 2
    hipLaunchKernelGGL(slip_update_kernel, dim3(1), dim3(64), 0, slk_stream(stream), x, n, slip, from_score, from_pos);
    return slk_launch_status();
}

// One workgroup (4 waves) per read.  LDS: the slip scan's (score, position) pairs, pscore and cscore (each with SLIP_BATCH
// floats of slack behind it, see slip_scan_wave), two emission rows and the sequence: 7 words per position + the slack.  Per event the last wave runs the slip scan over the previous scores while the other three gather the NEXT
// event's emissions ltrans[i+1][seq[j]] into LDS (a dependent global gather that would otherwise sit on the critical path of
// every event); then all four waves do the stay / step / slip comparison for their positions.
#define MAP_LDS_WORDS 7
__device__ __forceinline__ void map_to_sequence_body(float *sm, const float *__restrict__ ltrans, int nev, int nst,
                                                     const int32_t *__restrict__ seq, int npos, float slip,
                                                     const double *__restrict__ prior_initial,
                                                     const double *__restrict__ prior_final, int32_t *__restrict__ vmat,
                                                     float *__restrict__ score_out, int32_t *__restrict__ path_out)
{
    float2 *fsp = reinterpret_cast<float2 *>(sm);
    float *pscore = sm + 2 * npos, *cscore = pscore + npos + SLIP_BATCH, *ce = cscore + npos + SLIP_BATCH, *ce_next = ce + npos;
    int *sq = reinterpret_cast<int *>(ce_next + npos);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int scan_first = nt - 64;                                         // threads of the last wave
    for (int j = tid; j < npos; j += nt) {
        int sj = seq[j];
        sq[j] = sj;
        float p = 0.0f;
        if (prior_initial) p = (float)((double)p + prior_initial[j]);       // transducer.py:39-40
        p += fmaxf(ltrans[sj], ltrans[0]);                                  // transducer.py:41
        pscore[j] = p;
        if (nev > 1) ce[j] = ltrans[(size_t)nst + sj];
    }
    __syncthreads();
    for (int i = 1; i < nev; i++) {
        const float *ct = ltrans + (size_t)i * nst;
        const float ct0 = ct[0];
        if (tid >= scan_first) {
            slip_scan_wave<true>(pscore, npos, slip, SlipOutPairs{fsp});      // transducer.py:56
        } else if (i + 1 < nev) {
            const float *cn = ct + nst;
            for (int j = tid; j < npos; j += scan_first) ce_next[j] = cn[sq[j]];
        }
        __syncthreads();
        int32_t *vm = vmat + (size_t)i * npos;
        for (int j = tid; j < npos; j += nt) {
            const float cej = ce[j];
            float c = pscore[j] + ct0;                                      // stay  :47
            int from = j;
            if (j > 0) {
                float ss = pscore[j - 1] + cej;                             // step  :49-52
                if (ss > c) { c = ss; from = j - 1; }
            }
            const float2 sl = fsp[j];
            float f = sl.x + cej;                                           // slip  :57-59
            if (!(f <= c)) { c = f; from = __float_as_int(sl.y); }
            cscore[j] = c;
            vm[j] = from;
        }
        __syncthreads();
        float *tmp = pscore; pscore = cscore; cscore = tmp;
        tmp = ce; ce = ce_next; ce_next = tmp;
    }
    if (prior_final) {
        for (int j = tid; j < npos; j += nt) pscore[j] = (float)((double)pscore[j] + prior_final[j]);  // :63-64
        __syncthreads();
    }
    if (tid < 64) {
        // np.argmax :68 -- the FIRST maximum: every lane keeps the first maximum of its strided positions, then a butterfly
        // that prefers the larger score and, between equal scores, the smaller position.
        float bv = -INFINITY;
        int best = 0x7fffffff;
        for (int j = tid; j < npos; j += 64) {
            const float v = pscore[j];
            if (v > bv || best == 0x7fffffff) { bv = v; best = j; }
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float ov = __shfl_xor(bv, d);
            const int ob = __shfl_xor(best, d);
            if (ob != 0x7fffffff && (best == 0x7fffffff || ov > bv || (ov == bv && ob < best))) { bv = ov; best = ob; }
        }
        if (tid == 0) {
            score_out[0] = bv;
            path_out[nev - 1] = best;
        }
        // Backtrace :70-71.  One dependent global load per event would cost a memory latency per event; instead the wave
        // fetches, for the next 32 events at once, the 64 traceback entries at and below the current position (a path moves
        // down by 0 or 1 per event, a slip further), and follows the chain through registers with v_readlane.  A slip that
        // leaves the window ends the batch early.
        int cur = best, r = nev - 1;                                        // path_out[r] = cur is known
        while (r >= 1) {
            const int base = max(cur - 63, 0);
            int rowv[32];
#pragma unroll
            for (int l = 0; l < 32; l++) {
                const int row = r - l;
                rowv[l] = (row >= 1 && base + tid < npos) ? vmat[(size_t)row * npos + base + tid] : 0;
            }
            int mine = 0, done = 0;
            bool ok = true;
#pragma unroll
            for (int l = 0; l < 32; l++) {
                if (ok && r - l >= 1 && cur >= base) {
                    cur = __builtin_amdgcn_readlane(rowv[l], cur - base);
                    if (tid == l) mine = cur;
                    done = l + 1;
                } else {
                    ok = false;
                }
            }
            if (tid < done) path_out[r - 1 - tid] = mine;
            r -= done;
        }
    }
}

__global__ void __launch_bounds__(256) map_to_sequence_kernel(const float *__restrict__ ltrans, int nev, int nst,
                                                              const int32_t *__restrict__ seq, int npos, float slip,
                                                              const double *__restrict__ prior_initial,
                                                              const double *__restrict__ prior_final,
                                                              int32_t *__restrict__ vmat, float *__restrict__ score_out,
                                                              int32_t *__restrict__ path_out)
{
    extern __shared__ float sm[];
    map_to_sequence_body(sm, ltrans, nev, nst, seq, npos, slip, prior_initial, prior_final, vmat, score_out, path_out);
}

// Batched remap (the reference maps reads one at a time, bin/chunkify.py remap -> transducer.map_to_sequence): reads of
// different lengths are concatenated; read b owns events ev_off[b]..ev_off[b+1] of ltrans / path_out, positions
// pos_off[b]..pos_off[b+1] of seq and the priors, and ws_off[b].. of the traceback workspace.  One workgroup per read.
__global__ void __launch_bounds__(256) map_to_sequence_batch_kernel(const float *__restrict__ ltrans, int nst,
                                                                    const int64_t *__restrict__ ev_off,
                                                                    const int32_t *__restrict__ seq,
                                                                    const int64_t *__restrict__ pos_off, float slip,
                                                                    const double *__restrict__ prior_initial,
                                                                    const double *__restrict__ prior_final,
                                                                    int32_t *__restrict__ vmat,
                                                                    const int64_t *__restrict__ ws_off,
                                                                    float *__restrict__ score_out,
                                                                    int32_t *__restrict__ path_out)
{
    extern __shared__ float sm[];
    const int b = blockIdx.x;
    const int64_t e0 = ev_off[b], p0 = pos_off[b];
    const int nev = (int)(ev_off[b + 1] - e0), npos = (int)(pos_off[b + 1] - p0);
    if (nev < 1 || npos < 3) {                       // empty read: nothing to map (score -inf, no path)
        if (threadIdx.x == 0) score_out[b] = -INFINITY;
        return;
    }
    map_to_sequence_body(sm, ltrans + e0 * nst, nev, nst, seq + p0, npos, slip, prior_initial ? prior_initial + p0 : nullptr,
                         prior_final ? prior_final + p0 : nullptr, vmat + ws_off[b], score_out + b, path_out + e0);
}

// 160 KB of LDS per CU on gfx950; a request above 64 KB needs the function attribute raised once per device.
#define MAP_LDS_MAX (160 * 1024)
static bool lds_limit_raised(const void *kernel)
{
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MAP_LDS_MAX) == hipSuccess;
}

extern "C" size_t slk_map_to_sequence_workspace_bytes(int nev, int npos)
{
    if (nev < 1 || npos < 1) return 0;
    return sizeof(int32_t) * (size_t)nev * npos;
}

extern "C" int slk_map_to_sequence_f32(const float *ltrans, int nev, int nst, const int32_t *seq, int npos, float slip,
                                       const double *prior_initial, const double *prior_final, void *workspace,
                                       size_t workspace_bytes, float *score_out, int32_t *path_out, slk_stream_t stream)
{
    if (!ltrans || !seq || !score_out || !path_out || nev < 1 || nst < 1 || npos < 3) return SLK_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < slk_map_to_sequence_workspace_bytes(nev, npos)) return SLK_ERR_WORKSPACE;
    const size_t lds = ((size_t)npos * MAP_LDS_WORDS + 2 * SLIP_BATCH) * sizeof(float);
    if (lds > MAP_LDS_MAX) return SLK_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && !SLK_PER_DEVICE(bool, lds_limit_raised(reinterpret_cast<const void *>(map_to_sequence_kernel))))
        return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(map_to_sequence_kernel, dim3(1), dim3(256), lds, slk_stream(stream), ltrans, nev, nst, seq, npos,
                       slip, prior_initial, prior_final, static_cast<int32_t *>(workspace), score_out, path_out);
    return slk_launch_status();
}

extern "C" int slk_map_to_sequence_batch_f32(const float *ltrans, int nst, const int64_t *ev_off, const int32_t *seq,
                                             const int64_t *pos_off, int nread, int max_npos, float slip,
                                             const double *prior_initial, const double *prior_final, void *workspace,
                                             const int64_t *ws_off, float *score_out, int32_t *path_out,
                                             slk_stream_t stream)
{
    if (!ltrans || !ev_off || !seq || !pos_off || !workspace || !ws_off || !score_out || !path_out || nst < 1 || nread < 1 ||
        max_npos < 3)
        return SLK_ERR_INVALID_ARG;
    const size_t lds = ((size_t)max_npos * MAP_LDS_WORDS + 2 * SLIP_BATCH) * sizeof(float);
    if (lds > MAP_LDS_MAX) return SLK_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 &&
        !SLK_PER_DEVICE(bool, lds_limit_raised(reinterpret_cast<const void *>(map_to_sequence_batch_kernel))))
        return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(map_to_sequence_batch_kernel, dim3(nread), dim3(256), lds, slk_stream(stream), ltrans, nst, ev_off, seq,
                       pos_off, slip, prior_initial, prior_final, static_cast<int32_t *>(workspace), ws_off, score_out,
                       path_out);
    return slk_launch_status();
}
