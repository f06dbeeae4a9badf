// mfma_issue_probe.hip -- how fast can ONE wave issue v_mfma_f32_4x4x1_16b_f32, and how does the rate change with
// more waves per SIMD?  (Design input for the GRU kernels: their recurrent chain is issued by a single wave per SIMD.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_probe.hip -o /tmp/mfma_issue_probe && /tmp/mfma_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NW>
__global__ void __launch_bounds__(1024) probe(float *out, unsigned long long *cycles, int iters)
{
    float w[NW];
#pragma unroll
    for (int i = 0; i < NW; i++) w[i] = 0.001f * (threadIdx.x + i);
    float a = 1.0f + threadIdx.x;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NW; i++) acc[i % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, w[i], acc[i % NACC], 4, 3, 0);
    }
    const unsigned long long t1 = clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int NACC, int NW>
static void run(int waves_per_simd)
{
    float *out;
    unsigned long long *cyc, h = 0;
    const int threads = 256 * waves_per_simd, iters = 2000;
    hipMalloc(&out, sizeof(float) * 256 * threads);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, NW>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, NW>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / ((double)iters * NW);
    const double flop = 512.0 * iters * NW * (threads / 64) * 256.0;
    printf("acc=%d chain=%3d waves/SIMD=%d : %6.2f cycles per MFMA per wave, %5.2f per SIMD; wall %.3f ms = %.1f TFLOP/s\n", NACC,
           NW, waves_per_simd, per, per / waves_per_simd, ms, flop / ms / 1e9);
    hipFree(out);
    hipFree(cyc);
}

void run_mix();

int main()
{
    run_mix();
    for (int w = 1; w <= 4; w++) run<4, 96>(w);
    for (int w = 1; w <= 4; w++) run<8, 96>(w);
    for (int w = 1; w <= 2; w++) run<2, 96>(w);
    for (int w = 1; w <= 2; w++) run<1, 96>(w);
    return 0;
}

// ---- part 2: how much does a neighbour wave's MFMA stream slow a VALU stream on the same SIMD? ----
// waves 0-3 (one per SIMD) run gate-like math (4 independent sigmoids per iteration); waves 4-7 run MFMAs or idle.
__global__ void __launch_bounds__(512) mix_probe(float *out, unsigned long long *cycles, int iters, int neighbour, int prio)
{
    const int wave = threadIdx.x >> 6;
    float v0 = 0.1f * threadIdx.x, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (wave < 4) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it < iters; it++) {
            v0 = __builtin_amdgcn_rcpf(1.0f + __expf(-v0));
            v1 = __builtin_amdgcn_rcpf(1.0f + __expf(-v1));
            v2 = __builtin_amdgcn_rcpf(1.0f + __expf(-v2));
            v3 = __builtin_amdgcn_rcpf(1.0f + __expf(-v3));
        }
    } else if (neighbour == 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(v0, v1, acc[i & 3], 4, 3, 0);
        }
    } else if (neighbour == 2) {
        for (int it = 0; it < 2 * iters; it++) {
            v0 = v0 * 1.0001f + 0.5f; v1 = v1 * 1.0001f + 0.5f; v2 = v2 * 1.0001f + 0.5f; v3 = v3 * 1.0001f + 0.5f;
        }
    }
    const unsigned long long t1 = clock64();
    f32x4 s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v0 + v1 + v2 + v3;
    if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x == 0) cycles[wave >> 2] = t1 - t0;
}

void run_mix()
{
    float *out;
    unsigned long long *cyc, h[2];
    hipMalloc(&out, sizeof(float) * 256 * 512);
    hipMalloc(&cyc, 16);
    const int iters = 20000;
    const char *names[3] = {"idle", "MFMA 4x4x1 stream", "plain VALU fma stream"};
    for (int prio = 0; prio < 2; prio++)
        for (int nb = 0; nb < 3; nb++) {
            hipLaunchKernelGGL(mix_probe, dim3(256), dim3(512), 0, 0, out, cyc, iters, nb, prio);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            printf("gate wave (prio %d) with neighbour %-22s: %7.1f cycles per 4 sigmoids; neighbour ran %llu cycles\n", prio * 3,
                   names[nb], (double)h[0] / iters, h[1]);
        }
    hipFree(out);
    hipFree(cyc);
}
