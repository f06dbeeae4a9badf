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
//   * the state image is double buffered, so one LDS-only barrier per step suffices.
#include "mfma4.h"

template <int N>
__global__ void __launch_bounds__(256) lstm_mfma_kernel(const float *__restrict__ vW, const float *__restrict__ sW,
                                                        const float *__restrict__ peep, float *__restrict__ out, long ldo,
                                                        int T, int B, int reverse)
{
    static_assert(N % 16 == 0 && N <= 64, "lstm_mfma_kernel: sizes 16, 32, 48, 64");
    constexpr int NW = N / 4;                                   // neurons per wave (<= 16)
    constexpr int S = 4;                                        // K-slices: 4 * NW <= 64 lanes
    constexpr int LP = 64 / S;                                  // lanes per slice
    constexpr int M = N / S;                                    // MFMAs per gate chain
    constexpr int G = 16 / S;                                   // blocks per broadcast group
    constexpr int CB = 4 - ilog2(S);
    constexpr int NV = N / 16;                                  // packed state registers
    __shared__ __attribute__((aligned(16))) float hbuf[2][N * 4];   // h[k][chunk], step parity

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b0 = blockIdx.x * 4;
    const int blk = lane >> 2, ci = lane & 3;
    const int lb = lane % LP, slice = lane / LP;                // slice s evaluates chunk s
    const bool valid = lb < NW;
    const int neuron = wave * NW + (valid ? lb : 0);
    const int chunk = b0 + slice;
    const bool store = valid && chunk < B;

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

    // this lane's 4 gate inputs of (step, chunk): one 16-byte load; rows beyond the batch re-read the last chunk
    const int cclamp = chunk < B ? chunk : B - 1;
    auto load_v = [&](int s) {
        const int ss = s < T ? s : T - 1;
        const int t = reverse ? T - 1 - ss : ss;
        return *reinterpret_cast<const float4 *>(vW + ((size_t)t * B + cclamp) * (4 * N) + 4 * neuron);
    };
    float4 vA = load_v(0), vB = vA;
    __syncthreads();

    // two input buffers that swap roles statically (loop unrolled by two): copying a just-requested row would make
    // every step wait for that request
    auto step = [&](int s, const float4 &use, float4 &fill) {
        fill = load_v(s + 1);
        const float *hp_src = hbuf[s & 1];
        float *hn_dst = hbuf[(s + 1) & 1];
        float hp[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) hp[v] = hp_src[addr0 + 4 * v * G];
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
        // slice sums; this lane keeps the values of its own chunk
        float sum[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const f32x4 t4 = sum_slices<S>(acc[g]);
            sum[g] = slice == 0 ? t4[0] : (slice == 1 ? t4[1] : (slice == 2 ? t4[2] : t4[3]));
        }
        const float in0 = sum[0] + use.x, in1 = sum[1] + use.y, in2 = sum[2] + use.z, in3 = sum[3] + use.w;
        float os = cell * slk_sigmoid(in2 + cell * p_fg);                       // forget   layers.py:686
        os += slk_tanh(in0) * slk_sigmoid(in1 + cell * p_in);                   // update   layers.py:688
        const float o = slk_tanh(os) * slk_sigmoid(in3 + os * p_out);           // output   layers.py:690
        cell = os;
        if (valid) hn_dst[4 * neuron + slice] = o;
        if (store) {
            const int t = reverse ? T - 1 - s : s;
            out[((size_t)t * B + chunk) * ldo + neuron] = o;
        }
        lds_barrier();
    };
    for (int s = 0; s < T; s += 2) {
        step(s, vA, vB);
        if (s + 1 < T) step(s + 1, vB, vA);
    }
}

template <int N>
static int launch_lstm_mfma(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B,
                            int reverse, hipStream_t s)
{
    hipLaunchKernelGGL((lstm_mfma_kernel<N>), dim3((B + 3) / 4), dim3(256), 0, s, vW, sW, p, out, ldo, T, B, reverse);
    return slk_launch_status();
}

// Returns SLK_ERR_UNSUPPORTED when the portable kernel has to be used.
int slk_lstm_mfma_dispatch(const float *vW, const float *sW, const float *p, float *out, long ldo, int T, int B, int n,
                           int reverse, int act, int gate_act, hipStream_t s)
{
    if (act != SLK_ACT_TANH || gate_act != SLK_ACT_SIGMOID) return SLK_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(vW) & 15) != 0) return SLK_ERR_UNSUPPORTED;
    switch (n) {
    case 16: return launch_lstm_mfma<16>(vW, sW, p, out, ldo, T, B, reverse, s);
    case 32: return launch_lstm_mfma<32>(vW, sW, p, out, ldo, T, B, reverse, s);
    case 48: return launch_lstm_mfma<48>(vW, sW, p, out, ldo, T, B, reverse, s);
    case 64: return launch_lstm_mfma<64>(vW, sW, p, out, ldo, T, B, reverse, s);
    default: return SLK_ERR_UNSUPPORTED;
    }
}
