// transducer.hip -- remap DP of the reference on gfx950.
//
//   viterbi_helpers.slip_update   sloika/viterbi_helpers.pyx:12-35  (the reference's only native function)
//   transducer.map_to_sequence    sloika/transducer.py:14-73
//
// slip_update is a running max with decay whose float32 result depends on the order of the repeated
// `- slip` roundings, so it is evaluated as the same sequential recurrence (one lane) to stay bit-identical;
// the stay/step/compare parts of map_to_sequence are lane-parallel over sequence positions and overlap with it.
#include "common.h"

__device__ __forceinline__ void slip_update_seq(const float *x, int n, float slip, float *from_score, int *from_pos)
{
    // viterbi_helpers.pyx:22-33.  The recurrence itself is sequential (its float32 roundings depend on the order), but its
    // inputs are not: eight x values are fetched together ahead of the eight dependent compare / select / subtract steps
    // that consume them, so the loop pays LDS latency once per eight positions instead of once per position.
    from_score[0] = from_score[1] = -1e38f;
    from_pos[0] = from_pos[1] = 0;
    float fs = x[0] - slip;
    int fp = 0;
    from_score[2] = fs;
    from_pos[2] = 0;
    int j = 3;
    for (; j + 8 <= n; j += 8) {
        float xv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) xv[u] = x[j - 2 + u];
        float fo[8];
        int po[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (!(fs >= xv[u])) { fp = j - 2 + u; fs = xv[u]; }
            fs = fs - slip;
            fo[u] = fs;
            po[u] = fp;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { from_score[j + u] = fo[u]; from_pos[j + u] = po[u]; }
    }
    for (; j < n; j++) {
        float xv = x[j - 2];
        if (!(fs >= xv)) { fp = j - 2; fs = xv; }
        fs = fs - slip;
        from_score[j] = fs;
        from_pos[j] = fp;
    }
}

__global__ void slip_update_kernel(const float *__restrict__ x, int n, float slip, float *__restrict__ from_score,
                                   int64_t *__restrict__ from_pos)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    from_score[0] = from_score[1] = -1e38f;
    from_pos[0] = from_pos[1] = 0;
    float fs = x[0] - slip;
    int64_t fp = 0;
    from_score[2] = fs;
    from_pos[2] = 0;
    for (int j = 3; j < n; j++) {
        float xv = x[j - 2];
        if (!(fs >= xv)) { fp = j - 2; fs = xv; }
        fs = fs - slip;
        from_score[j] = fs;
        from_pos[j] = fp;
    }
}

extern "C" int slk_slip_update_f32(const float *x, int n, float slip, float *from_score, int64_t *from_pos,
                                   slk_stream_t stream)
{
    if (!x || !from_score || !from_pos || n < 3) return SLK_ERR_INVALID_ARG;   // pyx:24 writes index 2
    hipLaunchKernelGGL(slip_update_kernel, dim3(1), dim3(64), 0, slk_stream(stream), x, n, slip, from_score, from_pos);
    return slk_launch_status();
}

// One workgroup per read.  LDS: pscore, cscore, fs (float), fp and the sequence (int), each of length npos.
__device__ __forceinline__ void map_to_sequence_body(float *sm, const float *__restrict__ ltrans, int nev, int nst,
                                                     const int32_t *__restrict__ seq, int npos, float slip,
                                                     const double *__restrict__ prior_initial,
                                                     const double *__restrict__ prior_final, int32_t *__restrict__ vmat,
                                                     float *__restrict__ score_out, int32_t *__restrict__ path_out)
{
    float *pscore = sm, *cscore = sm + npos, *fs = sm + 2 * npos;
    int *fp = reinterpret_cast<int *>(sm + 3 * npos);
    int *sq = fp + npos;
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int j = tid; j < npos; j += nt) {
        int sj = seq[j];
        sq[j] = sj;
        float p = 0.0f;
        if (prior_initial) p = (float)((double)p + prior_initial[j]);       // transducer.py:39-40
        p += fmaxf(ltrans[sj], ltrans[0]);                                  // transducer.py:41
        pscore[j] = p;
    }
    __syncthreads();
    for (int i = 1; i < nev; i++) {
        const float *ct = ltrans + (size_t)i * nst;
        const float ct0 = ct[0];
        // slip scan by one lane of the LAST wave, the rest do stay/step meanwhile
        if (tid == nt - 1) slip_update_seq(pscore, npos, slip, fs, fp);     // transducer.py:56
        __syncthreads();
        int32_t *vm = vmat + (size_t)i * npos;
        for (int j = tid; j < npos; j += nt) {
            const float ce = ct[sq[j]];
            float c = pscore[j] + ct0;                                      // stay  :47
            int from = j;
            if (j > 0) {
                float ss = pscore[j - 1] + ce;                              // step  :49-52
                if (ss > c) { c = ss; from = j - 1; }
            }
            float f = fs[j] + ce;                                           // slip  :57-59
            if (!(f <= c)) { c = f; from = fp[j]; }
            cscore[j] = c;
            vm[j] = from;
        }
        __syncthreads();
        float *tmp = pscore; pscore = cscore; cscore = tmp;
    }
    if (prior_final) {
        for (int j = tid; j < npos; j += nt) pscore[j] = (float)((double)pscore[j] + prior_final[j]);  // :63-64
        __syncthreads();
    }
    if (tid == 0) {
        int best = 0;
        for (int j = 1; j < npos; j++) if (pscore[j] > pscore[best]) best = j;   // np.argmax :68
        score_out[0] = pscore[best];
        int cur = best;
        path_out[nev - 1] = cur;
        __threadfence();
        for (int i = 1; i < nev; i++) {                                     // :70-71
            cur = vmat[(size_t)(nev - i) * npos + cur];
            path_out[nev - 1 - i] = cur;
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
    size_t lds = (size_t)npos * 5 * sizeof(float);
    if (lds > 64 * 1024) return SLK_ERR_UNSUPPORTED;
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
    size_t lds = (size_t)max_npos * 5 * sizeof(float);
    if (lds > 64 * 1024) return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(map_to_sequence_batch_kernel, dim3(nread), dim3(256), lds, slk_stream(stream), ltrans, nst, ev_off, seq,
                       pos_off, slip, prior_initial, prior_final, static_cast<int32_t *>(workspace), ws_off, score_out,
                       path_out);
    return slk_launch_status();
}
