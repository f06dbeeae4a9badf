// recurrent.hip -- the sequential part of Gru / Lstm on gfx950.
//
//   Gru.step   sloika/layers.py:1010-1021   (original Cho-style GRU: reset applied BEFORE the recurrent matmul,
//                                            no recurrent bias -- not the cuDNN/MIOpen variant)
//   Lstm.step  sloika/layers.py:677-691     (peepholes, interleaved gate layout row = j*4+g)
//   RNN.run    sloika/layers.py:85-88       (zero initial state, scan over time)
//   Reverse    sloika/layers.py:1449-1450   (time order flag instead of flipping tensors)
//
// Fast path (gru_mfma_kernel): one 256-thread workgroup per tile of 4 chunks, persistent over all T steps.
//   * The hidden-to-hidden weights live in VGPRs for the whole scan (n/SA + n/SB registers per lane).
//   * Contractions use v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction.
//       A operand = state h[k][chunk 0..3]   (4 values, broadcast to every block with CBSZ/ABID)
//       B operand = weights W[out(lane)][k]  (one output neuron per lane)
//       D[vgpr i][lane] += h[k][chunk i] * W[out(lane)][k]
//     i.e. a 64-outputs x 4-chunks x 1-k rank-1 update in 8 cycles, exact fp32 (fmaf chain), at the full fp32
//     matrix rate with a batch tile of only 4 -- which is what lets B=1024 chunks fill all 256 CUs.
//   * Wave w owns neurons [w*n/4, (w+1)*n/4).  Phase A computes its z|r pre-activations, phase B its candidate.
//     When the outputs of a phase fill only part of the 64 lanes, the spare lanes take another K-slice of the
//     same outputs (CBSZ picks one h value per 32/16-lane group) and the slices are summed with two shuffles.
//   * h and r*h are exchanged between the 4 waves through 2 x n x 16 B of LDS; the packed A-operand image
//     (lane = 4*block + chunk) is read back with lane-linear, conflict-free ds_read_b32.
//   * two s_barrier per step; the next step's input projection vI is prefetched a full step ahead.
//
// Portable path (gru_generic_kernel / lstm_generic_kernel): one workgroup per chunk, weights streamed from L2,
// any n and any activation; used for layer sizes the MFMA kernel is not instantiated for.
#include "mfma4.h"

// =====================================================================================================
// generic GRU
// =====================================================================================================
__global__ void __launch_bounds__(256) gru_generic_kernel(const float *__restrict__ vI, const float *__restrict__ sW,
                                                          const float *__restrict__ sW2, float *__restrict__ h_out,
                                                          long ldh, int T, int B, int n, int reverse, int act,
                                                          int gate_act, const int *__restrict__ lens)
{
    extern __shared__ float sm[];
    float *h = sm, *rh = sm + n, *z = sm + 2 * n;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (lens) T = min(max(lens[b], 1), T);                       // ragged batch: this chunk's own length
    for (int j = tid; j < n; j += nt) h[j] = 0.0f;
    __syncthreads();
    for (int s = 0; s < T; s++) {
        const int t = reverse ? T - 1 - s : s;
        const float *vt = vI + ((size_t)t * B + b) * 3 * n;
        for (int o = tid; o < 2 * n; o += nt) {
            const float *w = sW + (size_t)o * n;
            float v = vt[o];
            for (int k = 0; k < n; k++) v = fmaf(h[k], w[k], v);
            float g = slk_act(gate_act, v);
            if (o < n) z[o] = g;
            else rh[o - n] = g * h[o - n];
        }
        __syncthreads();
        for (int j = tid; j < n; j += nt) {
            const float *w = sW2 + (size_t)j * n;
            float v = vt[2 * n + j];
            for (int k = 0; k < n; k++) v = fmaf(rh[k], w[k], v);
            float hbar = slk_act(act, v);
            float hn = z[j] * h[j] + (1.0f - z[j]) * hbar;
            h[j] = hn;
            h_out[((size_t)t * B + b) * ldh + j] = hn;
        }
        __syncthreads();
    }
}

// =====================================================================================================
// generic LSTM
// =====================================================================================================
__global__ void __launch_bounds__(256) lstm_generic_kernel(const float *__restrict__ vW, const float *__restrict__ sW,
                                                           const float *__restrict__ p, float *__restrict__ out,
                                                           long ldo, int T, int B, int n, int reverse, int act,
                                                           int gate_act, const int *__restrict__ lens)
{
    extern __shared__ float sm[];
    float *o_prev = sm, *cell = sm + n, *sum = sm + 2 * n; // sum: [n][4]
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (lens) T = min(max(lens[b], 1), T);                   // ragged batch: this chunk's own length
    for (int j = tid; j < n; j += nt) { o_prev[j] = 0.0f; cell[j] = 0.0f; }
    __syncthreads();
    for (int s = 0; s < T; s++) {
        const int t = reverse ? T - 1 - s : s;
        const float *vt = vW + ((size_t)t * B + b) * 4 * n;
        for (int r = tid; r < 4 * n; r += nt) {
            const float *w = sW + (size_t)r * n;
            float v = vt[r];
            for (int k = 0; k < n; k++) v = fmaf(o_prev[k], w[k], v);
            sum[r] = v;
        }
        __syncthreads();
        for (int j = tid; j < n; j += nt) {
            float st = cell[j];
            float p0 = p ? p[j] : 0.0f, p1 = p ? p[n + j] : 0.0f, p2 = p ? p[2 * n + j] : 0.0f;
            float os = st * slk_act(gate_act, sum[j * 4 + 2] + st * p1);                 // forget   layers.py:686
            os += slk_act(act, sum[j * 4 + 0]) * slk_act(gate_act, sum[j * 4 + 1] + st * p0); // update layers.py:688
            float o = slk_act(act, os) * slk_act(gate_act, sum[j * 4 + 3] + os * p2);    // output   layers.py:690
            cell[j] = os;
            o_prev[j] = o;
            out[((size_t)t * B + b) * ldo + j] = o;
        }
        __syncthreads();
    }
}

// =====================================================================================================
// MFMA GRU
// =====================================================================================================
// N: layer size (multiple of 16).  NWAVES waves share the neurons (N / NWAVES each, a multiple of 4, at most 32):
// 4 waves up to N = 128; N = 144 (the middle layer of models/pretrained.pkl) runs on 12 waves of 12 neurons.
// ACT/GACT: compile-time activation ids, or -1 to use the runtime ids.
template <int N, int ACT, int GACT, int NWAVES = 4>
__global__ void __launch_bounds__(64 * NWAVES, (NWAVES + 3) / 4) gru_mfma_kernel(const float *__restrict__ vI, const float *__restrict__ sW,
                                                          const float *__restrict__ sW2, float *__restrict__ h_out,
                                                          long ldh, int T, int B, int reverse, int act, int gate_act,
                                                          const int *__restrict__ lens)
{
    constexpr int NW = N / NWAVES;                                 // neurons per wave
    constexpr int NT = 64 * NWAVES;                                // threads per workgroup
    constexpr int SA = (2 * NW <= 16) ? 4 : ((2 * NW <= 32) ? 2 : 1); // K-slices, phase A (z|r: 2*NW outputs)
    constexpr int SB = (NW <= 16) ? 4 : ((NW <= 32) ? 2 : 1);       // K-slices, phase B (NW outputs)
    constexpr int LPA = 64 / SA, LPB = 64 / SB;                     // lanes per slice
    constexpr int MA = N / SA, MB = N / SB;                         // MFMAs per phase
    constexpr int GA = 16 / SA, GB = 16 / SB;                       // blocks per broadcast group
    constexpr int CBA = 4 - ilog2(SA), CBB = 4 - ilog2(SB);
    constexpr int NV = N / 16;                                      // packed state registers
    static_assert(N % 16 == 0 && N % NWAVES == 0 && NW % 4 == 0 && NW <= 32, "unsupported GRU size for the MFMA kernel");

    constexpr int KB = N <= 128 ? 8 : 4;        // time steps of vI staged per LDS block (LDS budget)
    constexpr int ROWF4 = 3 * N / 4;            // float4 per (step, chunk) row of vI
    constexpr int BLKF4 = KB * 4 * ROWF4;       // float4 per staged block (a multiple of 64)
    constexpr int NDMA = (BLKF4 / 64 + NWAVES - 1) / NWAVES;   // 1-KiB LDS-DMA instructions per wave per block
    constexpr int BLKF = KB * 4 * 3 * N;        // floats per block

    __shared__ __attribute__((aligned(16))) float vbuf[2 * BLKF];          // vI[2][step in block][chunk][3N]
    __shared__ __attribute__((aligned(16))) float obuf[2 * KB * 4 * N];    // h_out[2][step in block][chunk][N]
    __shared__ __attribute__((aligned(16))) float hbuf[N * 4];             // h[k][chunk]
    __shared__ __attribute__((aligned(16))) float rhbuf[N * 4];            // (r*h)[k][chunk]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    const int blk = lane >> 2, ci = lane & 3;

    // ---- lane roles ----
    const int la = lane % LPA, ga = lane / LPA;
    const bool validA = la < 2 * NW;
    const bool isR = la >= NW;
    const int neuronA = wave * NW + (validA ? (la % NW) : 0);
    const int rowA = isR ? N + neuronA : neuronA;
    const int lb = lane % LPB, gb = lane / LPB;
    const bool validB = lb < NW;
    const int neuronB = wave * NW + (validB ? lb : 0);
    const bool zlane = lane < NW;                 // holds z, c and the new h of neuron wave*NW + lane
    const bool rlane = lane >= NW && lane < 2 * NW;

    // ---- weights -> registers (B operands) ----
    float wA[MA], wB[MB];
    {
        const float *pa = sW + (size_t)rowA * N + ga * MA;
#pragma unroll
        for (int m = 0; m < MA; m++) wA[m] = validA ? pa[m] : 0.0f;
        const float *pb = sW2 + (size_t)neuronB * N + gb * MB;
#pragma unroll
        for (int m = 0; m < MB; m++) wB[m] = validB ? pb[m] : 0.0f;
    }

    // packed-operand read addresses (floats): 4*(g*(N/S) + v*G + q) + chunk
    const int addrA0 = 4 * ((blk / GA) * MA + (blk % GA)) + ci;
    const int addrB0 = 4 * ((blk / GB) * MB + (blk % GB)) + ci;

    for (int i = tid; i < N * 4; i += NT) hbuf[i] = 0.0f;

    // The input projection vI is streamed HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
    // instruction, no registers), one block of KB time steps at a time into a 2-deep ring; block k+2 is issued as
    // soon as block k has been consumed, so a whole block of compute (KB steps) hides the HBM latency and nothing
    // on the per-step path touches global memory except the h_out stores.
    auto dma_block = [&](int s0, int slot) {
#pragma unroll
        for (int j = 0; j < NDMA; j++) {
            const int piece = j * NWAVES + wave;               // 64 float4 = 1 KiB of the block image
            if (piece * 64 < BLKF4) {                          // wave-uniform
                const int idx = piece * 64 + lane;
                const int kk = idx / (4 * ROWF4), r = idx % (4 * ROWF4), c = r / ROWF4, f4 = r % ROWF4;
                const int bc = min(b0 + c, B - 1);
                const int Tc = lens ? min(max(lens[bc], 1), T) : T;      // ragged batch: include/sloika_amd.h
                const int ss = min(s0 + kk, Tc - 1);
                const int tt = reverse ? Tc - 1 - ss : ss;
                const float *src = vI + ((size_t)tt * B + bc) * (3 * N) + 4 * f4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)&vbuf[slot * BLKF + piece * 256],
                                                 16, 0, 0);
            }
        }
    };
    // h_out goes the other way through a second 2-deep LDS ring: per step each owner lane drops its 4 values into
    // the ring (4 ds_write_b32), and once per KB steps the whole workgroup flushes the finished block with 16-byte
    // coalesced stores -- instead of 16 partial-line store instructions per workgroup on every step.
    const bool vec_store = (ldh % 4 == 0) && ((reinterpret_cast<uintptr_t>(h_out) & 15) == 0);
    auto flush_block = [&](int s0, int slot) {
        constexpr int OF4 = KB * 4 * N / 4;                    // float4 per output block
#pragma unroll
        for (int j = 0; j < (OF4 + NT - 1) / NT; j++) {
            const int idx = tid + NT * j;
            const int kk = idx / N, r = idx % N, c = r / (N / 4), f4 = r % (N / 4);
            const int ss = s0 + kk;
            const int Tc = (lens && b0 + c < B) ? min(max(lens[b0 + c], 1), T) : T;
            if (idx < OF4 && ss < Tc && b0 + c < B) {
                const int tt = reverse ? Tc - 1 - ss : ss;
                const float4 v = *reinterpret_cast<const float4 *>(&obuf[slot * (KB * 4 * N) + 4 * idx]);
                float *dst = h_out + ((size_t)tt * B + b0 + c) * ldh + 4 * f4;
                if (vec_store) *reinterpret_cast<float4 *>(dst) = v;
                else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
            }
        }
    };
    const float mask_zr = (validA && ga == 0) ? 1.0f : 0.0f;   // K-slice lanes start their accumulators at 0
    const float mask_c = zlane ? 1.0f : 0.0f;

    dma_block(0, 0);
    if (T > KB) dma_block(KB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < T; s++) {
        const int kk = s % KB, kb = s / KB;
        const float *vrow = vbuf + (kb & 1) * BLKF + kk * (4 * 3 * N);
        if (kk == 0 && s > 0) flush_block(s - KB, (kb - 1) & 1);     // previous block is complete and published

        // ---------------- phase A: z | r ----------------
        float hp[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) hp[v] = hbuf[addrA0 + 4 * v * GA];
        f32x4 a0;
#pragma unroll
        for (int i = 0; i < 4; i++) a0[i] = vrow[i * 3 * N + rowA] * mask_zr;
        const f32x4 hown = *reinterpret_cast<const f32x4 *>(&hbuf[4 * neuronA]);
        f32x4 g;
        if constexpr (NWAVES <= 4) {
            f32x4 accA[4] = {a0, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain<CBA, GA>(hp, wA, accA, std::make_integer_sequence<int, MA>{});
            g = sum_slices<SA>((accA[0] + accA[1]) + (accA[2] + accA[3]));
        } else {
            f32x4 accA[2] = {a0, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain2<CBA, GA>(hp, wA, accA, std::make_integer_sequence<int, MA>{});
            g = sum_slices<SA>(accA[0] + accA[1]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) g[i] = act_sel<GACT>(gate_act, g[i]);
        if (rlane) *reinterpret_cast<f32x4 *>(&rhbuf[4 * neuronA]) = g * hown;
        if (kk == KB - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next block's DMA (issued KB steps ago) landed
        lds_barrier();

        // ---------------- phase B: candidate ----------------
        float rp[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) rp[v] = rhbuf[addrB0 + 4 * v * GB];
        f32x4 c0;                                   // read here, not at the top: 4 fewer registers live through phase A
#pragma unroll
        for (int i = 0; i < 4; i++) c0[i] = vrow[i * 3 * N + 2 * N + neuronB] * mask_c;
        f32x4 cc;
        if constexpr (NWAVES <= 4) {
            f32x4 accB[4] = {c0, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain<CBB, GB>(rp, wB, accB, std::make_integer_sequence<int, MB>{});
            cc = sum_slices<SB>((accB[0] + accB[1]) + (accB[2] + accB[3]));
        } else {
            f32x4 accB[2] = {c0, {0.f, 0.f, 0.f, 0.f}};
            mfma_chain2<CBB, GB>(rp, wB, accB, std::make_integer_sequence<int, MB>{});
            cc = sum_slices<SB>(accB[0] + accB[1]);
        }
        if (zlane) {
            // register-starved instantiations re-read the old state instead of keeping it live across the barrier
            const f32x4 hold = NWAVES <= 4 ? hown : *reinterpret_cast<const f32x4 *>(&hbuf[4 * neuronB]);
            f32x4 hn;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float hbar = act_sel<ACT>(act, cc[i]);
                hn[i] = g[i] * hold[i] + (1.0f - g[i]) * hbar;      // layers.py:1020
            }
            *reinterpret_cast<f32x4 *>(&hbuf[4 * neuronB]) = hn;
            float *orow = obuf + (kb & 1) * (KB * 4 * N) + kk * (4 * N) + neuronB;
#pragma unroll
            for (int i = 0; i < 4; i++) orow[i * N] = hn[i];
        }
        lds_barrier();
        // every wave has finished reading the current block (phase B still reads its candidate rows, so not before this
        // barrier): its ring slot takes the block after next
        if (kk == KB - 1 && s + 1 + KB < T) dma_block(s + 1 + KB, kb & 1);
    }
    flush_block(((T - 1) / KB) * KB, ((T - 1) / KB) & 1);
}

// N = 144 (the middle layer of models/pretrained.pkl) needs 12 waves; only the tanh/sigmoid instantiation fits the
// 170-register budget of 3 waves/SIMD without scratch, so other activations return false and take the generic kernel.
static bool launch_gru_mfma144(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T, int B,
                               int reverse, int act, int gate_act, const int *lens, hipStream_t s)
{
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return false;
    hipLaunchKernelGGL((gru_mfma_kernel<144, SLK_ACT_TANH, SLK_ACT_SIGMOID, 12>), dim3((B + 3) / 4), dim3(64 * 12), 0, s,
                       vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens);
    return true;
}

template <int N>
static int launch_gru_mfma(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T, int B,
                           int reverse, int act, int gate_act, const int *lens, hipStream_t s)
{
    dim3 grid((B + 3) / 4), block(256);
    if (act == SLK_ACT_TANH && gate_act == SLK_ACT_SIGMOID)
        hipLaunchKernelGGL((gru_mfma_kernel<N, SLK_ACT_TANH, SLK_ACT_SIGMOID>), grid, block, 0, s, vI, sW, sW2, h_out, ldh,
                           T, B, reverse, act, gate_act, lens);
    else
        hipLaunchKernelGGL((gru_mfma_kernel<N, -1, -1>), grid, block, 0, s, vI, sW, sW2, h_out, ldh, T, B, reverse, act,
                           gate_act, lens);
    return slk_launch_status();
}

static int gru_recurrent_entry(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T, int B,
                               int n, int reverse, int act, int gate_act, int force_generic, const int32_t *lens,
                               slk_stream_t stream)
{
    if (!vI || !sW || !sW2 || !h_out || T < 1 || B < 1 || n < 1 || ldh < n || !slk_act_valid(act) ||
        !slk_act_valid(gate_act))
        return SLK_ERR_INVALID_ARG;
    hipStream_t s = slk_stream(stream);
    if (!force_generic) {
        switch (n) {
        case 16: return launch_gru_mfma<16>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 32: return launch_gru_mfma<32>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 48: return launch_gru_mfma<48>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 64: return launch_gru_mfma<64>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 80: return launch_gru_mfma<80>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 96: return launch_gru_mfma<96>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 112: return launch_gru_mfma<112>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 128: return launch_gru_mfma<128>(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s);
        case 144:
            if (launch_gru_mfma144(vI, sW, sW2, h_out, ldh, T, B, reverse, act, gate_act, lens, s)) return slk_launch_status();
            break;
        default: break;
        }
    }
    size_t lds = sizeof(float) * 3 * (size_t)n;
    if (lds > 64 * 1024) return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(gru_generic_kernel, dim3(B), dim3(256), lds, s, vI, sW, sW2, h_out, ldh, T, B, n, reverse, act,
                       gate_act, lens);
    return slk_launch_status();
}

extern "C" int slk_gru_recurrent_f32_ex(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh,
                                        int T, int B, int n, int reverse, int act, int gate_act, int force_generic,
                                        slk_stream_t stream)
{
    return gru_recurrent_entry(vI, sW, sW2, h_out, ldh, T, B, n, reverse, act, gate_act, force_generic, nullptr, stream);
}

// Ragged batch: lens[b] in [1, T] valid steps of chunk b (include/sloika_amd.h).
extern "C" int slk_gru_recurrent_ragged_f32(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh,
                                            int T, int B, int n, int reverse, int act, int gate_act, const int32_t *lens,
                                            slk_stream_t stream)
{
    if (!lens) return SLK_ERR_INVALID_ARG;
    return gru_recurrent_entry(vI, sW, sW2, h_out, ldh, T, B, n, reverse, act, gate_act, 0, lens, stream);
}

extern "C" int slk_gru_recurrent_f32(const float *vI, const float *sW, const float *sW2, float *h_out, long ldh, int T,
                                     int B, int n, int reverse, int act, int gate_act, slk_stream_t stream)
{
    return slk_gru_recurrent_f32_ex(vI, sW, sW2, h_out, ldh, T, B, n, reverse, act, gate_act, 0, stream);
}

extern "C" size_t slk_gru_workspace_bytes(int T, int B, int n)
{
    if (T < 1 || B < 1 || n < 1) return 0;
    return sizeof(float) * (size_t)T * B * 3 * n;
}

extern "C" int slk_gru_f32(const float *x, long ldx, const float *iW, const float *sW, const float *sW2,
                           const float *bias, float *y, long ldy, int T, int B, int insize, int n, int reverse, int act,
                           int gate_act, void *workspace, size_t workspace_bytes, slk_stream_t stream)
{
    if (T < 1 || B < 1 || n < 1 || insize < 1) return SLK_ERR_INVALID_ARG;
    // the whole layer in one kernel where an instantiation exists (fp16-split products: float32-grade, not bit-equal to the pair below)
    int rc = slk_gru_bar16_f32(x, ldx, iW, sW, sW2, bias, y, ldy, T, B, insize, n, reverse & 1, act, gate_act, nullptr, nullptr, stream);
    if (rc != SLK_ERR_UNSUPPORTED) return rc;
    if (!workspace || workspace_bytes < slk_gru_workspace_bytes(T, B, n)) return SLK_ERR_WORKSPACE;
    float *vI = static_cast<float *>(workspace);
    rc = slk_gemm_bias_act_f32(x, ldx, iW, bias, vI, 3L * n, (long)T * B, insize, 3 * n, SLK_ACT_LINEAR, stream);
    if (rc != SLK_OK) return rc;
    return slk_gru_recurrent_f32(vI, sW, sW2, y, ldy, T, B, n, reverse, act, gate_act, stream);
}

int slk_lstm_mfma_dispatch(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B, int n,
                           int reverse, int act, int gate_act, const int *lens, hipStream_t s);          // lstm_mfma.hip

static int lstm_recurrent_entry(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B, int n,
                                int reverse, int act, int gate_act, const int32_t *lens, slk_stream_t stream)
{
    if (!vW || !sW || !out || T < 1 || B < 1 || n < 1 || ldo < n || !slk_act_valid(act) || !slk_act_valid(gate_act))
        return SLK_ERR_INVALID_ARG;
    {
        const int rc = slk_lstm_mfma_dispatch(vW, sW, p, out, ldo, T, B, n, reverse, act, gate_act, lens, slk_stream(stream));
        if (rc != SLK_ERR_UNSUPPORTED) return rc;               // lstm_mfma.hip: sizes 16..64, tanh / sigmoid
    }
    size_t lds = sizeof(float) * 6 * (size_t)n;
    if (lds > 64 * 1024) return SLK_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(lstm_generic_kernel, dim3(B), dim3(256), lds, slk_stream(stream), vW, sW, p, out, ldo, T, B, n,
                       reverse, act, gate_act, lens);
    return slk_launch_status();
}

extern "C" int slk_lstm_recurrent_f32(const float *vW, const float *sW, const float *p, float *out, long ldo, int T,
                                      int B, int n, int reverse, int act, int gate_act, slk_stream_t stream)
{
    return lstm_recurrent_entry(vW, sW, p, out, ldo, T, B, n, reverse, act, gate_act, nullptr, stream);
}

// Ragged batch: lens[b] in [1, T] valid steps of chunk b (include/sloika_amd.h).
extern "C" int slk_lstm_recurrent_ragged_f32(const float *vW, const float *sW, const float *p, float *out, long ldo, int T,
                                             int B, int n, int reverse, int act, int gate_act, const int32_t *lens,
                                             slk_stream_t stream)
{
    if (!lens) return SLK_ERR_INVALID_ARG;
    return lstm_recurrent_entry(vW, sW, p, out, ldo, T, B, n, reverse, act, gate_act, lens, stream);
}

extern "C" size_t slk_lstm_workspace_bytes(int T, int B, int n)
{
    if (T < 1 || B < 1 || n < 1) return 0;
    return sizeof(float) * (size_t)T * B * 4 * n;
}

extern "C" int slk_lstm_f32(const float *x, long ldx, const float *iW, const float *sW, const float *bias,
                            const float *p, float *y, long ldy, int T, int B, int insize, int n, int reverse, int act,
                            int gate_act, void *workspace, size_t workspace_bytes, slk_stream_t stream)
{
    if (T < 1 || B < 1 || n < 1 || insize < 1) return SLK_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < slk_lstm_workspace_bytes(T, B, n)) return SLK_ERR_WORKSPACE;
    float *vW = static_cast<float *>(workspace);
    int rc = slk_gemm_bias_act_f32(x, ldx, iW, bias, vW, 4L * n, (long)T * B, insize, 4 * n, SLK_ACT_LINEAR, stream);
    if (rc != SLK_OK) return rc;
    return slk_lstm_recurrent_f32(vW, sW, p, y, ldy, T, B, n, reverse, act, gate_act, stream);
}

// =====================================================================================================
// Device self-test (include/sloika_amd.h): hardware-layout probe for v_mfma_f32_4x4x1_16b_f32 (tests/test_gpu_mfma_probe.py): returns the raw
// accumulator so that the operand/broadcast assumptions of gru_mfma_kernel are checked on the device.
// =====================================================================================================
template <int CB, int AB>
__global__ void mfma4_probe_kernel(const float *a, const float *b, float *d)
{
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = mfma4<CB, AB>(a[threadIdx.x], b[threadIdx.x], c);
    for (int i = 0; i < 4; i++) d[i * 64 + threadIdx.x] = c[i];
}

extern "C" int slk_selftest_mfma4_f32(const float *a, const float *b, float *d, int cbsz, int abid, slk_stream_t stream)
{
    hipStream_t s = slk_stream(stream);
#define PROBE(CB, AB)                                                                              \
    if (cbsz == CB && abid == AB) {                                                                \
        hipLaunchKernelGGL((mfma4_probe_kernel<CB, AB>), dim3(1), dim3(64), 0, s, a, b, d);        \
        return slk_launch_status();                                                                \
    }
    PROBE(0, 0) PROBE(4, 0) PROBE(4, 3) PROBE(4, 15) PROBE(3, 0) PROBE(3, 5) PROBE(2, 1) PROBE(2, 3)
#undef PROBE
    return SLK_ERR_UNSUPPORTED;
}
