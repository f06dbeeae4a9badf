// lstm_mfma.hip -- the recurrence of an Lstm layer (sloika/layers.py:677-691, peepholes included) on gfx950, for layer
// sizes up to 64 (the shipped models/baseline_lstm.py is 64 wide).  The input projection x.iW^T + b comes from the GEMM
// (slk_lstm_f32 in recurrent.hip chains the two); sizes / activations this kernel does not cover take the portable kernel.
//
// Same machinery as the GRU kernels (mfma4.h): one 256-thread workgroup walks FOUR chunks through time; the state lives
// in LDS as the packed A-operand image h[k][chunk] of v_mfma_f32_4x4x1_16b_f32, the recurrent weights in registers.
//   * wave w owns neurons w*N/4 .. ; a lane is (neuron, K-slice): the FOUR gate pre-activations of its neuron are four
//     MFMA chains over the lane's K-slice (the reference interleaves the gate rows, row = 4*neuron + gate,
//     layers.py:682-690), so after the slice sum every lane of a neuron holds all four gates for all four chunks;
//   * the K-slices of a neuron then split the gate math by chunk (slice s evaluates chunk(s) s, s+S, ...), so the five
//     transcendental evaluations per (neuron, chunk) are done once, not once per slice; the cell state of (neuron, chunk)
//     stays in that lane's registers for the whole scan;
//   * the state image is double buffered, so one LDS-only barrier per step suffices;
//   * inputs arrive by LDS-DMA and outputs leave through an LDS ring, a block of 8 steps at a time.
#include "mfma4.h"

// Reduce-scatter over the four K-slices: every lane ends up with the sum, over the four slices, of the accumulator
// component that belongs to ITS chunk (slice s keeps chunk s) -- 3 swaps + 3 adds per gate instead of the 8 + 8 of an
// all-reduce of all four components.  (permlane32_swap(v, v) returns {lower half's v, upper half's v} in all lanes.)
__device__ __forceinline__ float scatter_sum4(f32x4 a, int slice)
{
    // pairs (0,2) and (1,3) live 32 lanes apart: a lane in slices 0/1 keeps components {0,1}, in slices 2/3 keeps {2,3}
    const bool upper = slice >= 2;
    const float keep0 = upper ? a[2] : a[0], keep1 = upper ? a[3] : a[1];
    const float give0 = upper ? a[0] : a[2], give1 = upper ? a[1] : a[3];
    auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(give0), __float_as_uint(give0), false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(give1), __float_as_uint(give1), false, false);
    // partner's "give" is what this half keeps: lower lanes need the upper half's value (index 1) and vice versa
    const float p0 = keep0 + __uint_as_float(upper ? s0[0] : s0[1]);
    const float p1 = keep1 + __uint_as_float(upper ? s1[0] : s1[1]);
    // now slices (0,1) hold partial {chunk 0, chunk 1} sums, slices (2,3) {chunk 2, chunk 3}; partner 16 lanes apart
    const bool odd = slice & 1;
    const float keep = odd ? p1 : p0, give = odd ? p0 : p1;
    auto s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(give), __float_as_uint(give), false, false);
    // permlane16_swap(v, v): result[0] = rows {0,0,2,2} of v, result[1] = rows {1,1,3,3}: the partner row's value
    return keep + __uint_as_float(odd ? s2[0] : s2[1]);
}

template <int N>
__global__ void __launch_bounds__(256) lstm_mfma_kernel(const float *__restrict__ vW, const float *__restrict__ sW,
                                                        const float *__restrict__ peep, float *__restrict__ out, long ldo,
                                                        int T, int B, int reverse, const int *__restrict__ lens)
{
    static_assert(N % 16 == 0 && N <= 64, "lstm_mfma_kernel: sizes 16, 32, 48, 64");
    constexpr int NW = N / 4;                                   // neurons per wave (<= 16)
    constexpr int S = 4;                                        // K-slices: 4 * NW <= 64 lanes
    constexpr int LP = 64 / S;                                  // lanes per slice
    constexpr int M = N / S;                                    // MFMAs per gate chain
    constexpr int G = 16 / S;                                   // blocks per broadcast group
    constexpr int CB = 4 - ilog2(S);
    constexpr int NV = N / 16;                                  // packed state registers
    constexpr int KB = 8;                                       // steps per staged block of inputs / outputs
    constexpr int ROWF4 = N;                                    // float4 per (step, chunk) row of vW (4N floats)
    constexpr int BLKF4 = KB * 4 * ROWF4;                       // float4 per staged input block
    constexpr int NDMA = BLKF4 / 64 / 4;                        // 1-KiB LDS-DMA instructions per wave and block
    static_assert(BLKF4 % 256 == 0, "input block must split evenly over the four waves");
    __shared__ __attribute__((aligned(16))) float hbuf[2][N * 4];             // h[k][chunk], step parity
    __shared__ __attribute__((aligned(16))) float vbuf[2][KB * 4 * 4 * N];    // vW[2][step][chunk][4N]: LDS-DMA ring
    __shared__ __attribute__((aligned(16))) float obuf[2][KB * 4 * N];        // out[2][step][chunk][N]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    const int blk = lane >> 2, ci = lane & 3;
    const int lb = lane % LP, slice = lane / LP;                // slice s evaluates chunk s
    const bool valid = lb < NW;
    const int neuron = wave * NW + (valid ? lb : 0);

    // gate rows of this neuron: 0 candidate, 1 input gate, 2 forget gate, 3 output gate (layers.py:682-690)
    float w[4][M];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float *pw = sW + (size_t)(4 * neuron + g) * N + slice * M;
#pragma unroll
        for (int m = 0; m < M; m++) w[g][m] = valid ? pw[m] : 0.0f;
    }
    const float p_in = peep ? peep[neuron] : 0.0f, p_fg = peep ? peep[N + neuron] : 0.0f, p_out = peep ? peep[2 * N + neuron] : 0.0f;
    const int addr0 = 4 * ((blk / G) * M + (blk % G)) + ci;     // packed-operand read address, register 0

    for (int i = tid; i < N * 4; i += 256) hbuf[0][i] = 0.0f;    // o_prev = 0 (layers.py:677)
    float cell = 0.0f;

    // The input projection streams HBM -> LDS by LDS-DMA, a block of KB steps at a time into a 2-deep ring, and the
    // outputs leave through a second ring flushed with 16-byte stores once per block (as in gru_mfma_kernel): the
    // per-step path touches no global memory, so no step waits on a memory round trip.
    auto dma_block = [&](int s0, int slot) {
#pragma unroll
        for (int j = 0; j < NDMA; j++) {
            const int idx = (j * 4 + wave) * 64 + lane;          // float4 index inside the block image
            const int kk = idx / (4 * ROWF4), r = idx % (4 * ROWF4), c = r / ROWF4, f4 = r % ROWF4;
            const int bc = min(b0 + c, B - 1);
            const int Tc = lens ? min(max(lens[bc], 1), T) : T;          // ragged batch: include/sloika_amd.h
            const int ss = min(s0 + kk, Tc - 1);
            const int tt = reverse ? Tc - 1 - ss : ss;
            const float *src = vW + ((size_t)tt * B + bc) * (4 * N) + 4 * f4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)&vbuf[slot][(j * 4 + wave) * 256], 16, 0, 0);
        }
    };
    const bool vec_store = (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    auto flush_block = [&](int s0, int slot) {
        constexpr int OF4 = KB * 4 * N / 4;                      // float4 per output block
#pragma unroll
        for (int j = 0; j < (OF4 + 255) / 256; j++) {
            const int idx = tid + 256 * j;
            const int kk = idx / N, r = idx % N, c = r / (N / 4), f4 = r % (N / 4);
            const int ss = s0 + kk;
            const int Tc = (lens && b0 + c < B) ? min(max(lens[b0 + c], 1), T) : T;
            if (idx < OF4 && ss < Tc && b0 + c < B) {
                const int tt = reverse ? Tc - 1 - ss : ss;
                const float4 v = *reinterpret_cast<const float4 *>(&obuf[slot][4 * idx]);
                float *dst = out + ((size_t)tt * B + b0 + c) * ldo + 4 * f4;
                if (vec_store) *reinterpret_cast<float4 *>(dst) = v;
                else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
            }
        }
    };

    dma_block(0, 0);
    if (T > KB) dma_block(KB, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int s = 0; s < T; s++) {
        const int kk = s % KB, kb = s / KB;
        if (kk == 0 && s > 0) flush_block(s - KB, (kb - 1) & 1);            // previous block is complete and published
        const float *hp_src = hbuf[s & 1];
        float *hn_dst = hbuf[(s + 1) & 1];
        float hp[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) hp[v] = hp_src[addr0 + 4 * v * G];
        // this lane's four gate inputs of (step, its chunk): one 16-byte LDS read
        const f32x4 use = *reinterpret_cast<const f32x4 *>(&vbuf[kb & 1][(kk * 4 + slice) * (4 * N) + 4 * neuron]);
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int m = 0; m < M; m++) {
            const float a = hp[m / G];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                switch (m % G) {                                 // ABID must be an immediate
                case 0: acc[g] = mfma4<CB, 0>(a, w[g][m], acc[g]); break;
                case 1: acc[g] = mfma4<CB, 1>(a, w[g][m], acc[g]); break;
                case 2: acc[g] = mfma4<CB, 2>(a, w[g][m], acc[g]); break;
                default: acc[g] = mfma4<CB, 3>(a, w[g][m], acc[g]); break;
                }
            }
        }
        const float in0 = scatter_sum4(acc[0], slice) + use[0], in1 = scatter_sum4(acc[1], slice) + use[1];
        const float in2 = scatter_sum4(acc[2], slice) + use[2], in3 = scatter_sum4(acc[3], slice) + use[3];
        float os = cell * slk_sigmoid(in2 + cell * p_fg);                       // forget   layers.py:686
        os += slk_tanh(in0) * slk_sigmoid(in1 + cell * p_in);                   // update   layers.py:688
        const float o = slk_tanh(os) * slk_sigmoid(in3 + os * p_out);           // output   layers.py:690
        cell = os;
        if (valid) {
            hn_dst[4 * neuron + slice] = o;
            obuf[kb & 1][(kk * 4 + slice) * N + neuron] = o;
        }
        if (kk == KB - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // next block's DMA (issued KB steps ago) landed
        lds_barrier();
        // every wave is past its reads of the current input block: its ring slot takes the block after next
        if (kk == KB - 1 && s + 1 + KB < T) dma_block(s + 1 + KB, kb & 1);
    }
    flush_block(((T - 1) / KB) * KB, ((T - 1) / KB) & 1);
}

template <int N>
static int launch_lstm_mfma(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B,
                            int reverse, const int *lens, hipStream_t s)
{
    hipLaunchKernelGGL((lstm_mfma_kernel<N>), dim3((B + 3) / 4), dim3(256), 0, s, vW, sW, p, out, ldo, T, B, reverse, lens);
    return slk_launch_status();
}

// Returns SLK_ERR_UNSUPPORTED when the portable kernel has to be used.
int slk_lstm_mfma_dispatch(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B, int n,
                           int reverse, int act, int gate_act, const int *lens, hipStream_t s)
{
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(vW) & 15) != 0) return SLK_ERR_UNSUPPORTED;
    switch (n) {
    case 16: return launch_lstm_mfma<16>(vW, sW, p, out, ldo, T, B, reverse, lens, s);
    case 32: return launch_lstm_mfma<32>(vW, sW, p, out, ldo, T, B, reverse, lens, s);
    case 48: return launch_lstm_mfma<48>(vW, sW, p, out, ldo, T, B, reverse, lens, s);
    case 64: return launch_lstm_mfma<64>(vW, sW, p, out, ldo, T, B, reverse, lens, s);
    default: return SLK_ERR_UNSUPPORTED;
    }
}
