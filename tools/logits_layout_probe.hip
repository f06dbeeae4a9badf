// logits_layout_probe.hip -- would a tile-blocked logits layout pay?  Writer: the softmax projection's store pattern (128 rows x
// 64-column tiles, 17 tiles per workgroup); reader: the decoder's (one workgroup per chunk, one row of 1025 floats per step,
// 800 steps, next row prefetched).  Layout R: rows [m][1056] (today).  Layout B: [t][tile][b][64] -- a writer tile is 32 KB
// contiguous, a reader row is 17 pieces of 256 B, 256 KB apart.
//   hipcc -O3 --offload-arch=gfx950 tools/logits_layout_probe.hip -o tools/_build/logits_layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr long T = 800, B = 1024, NT = 17, LD = 1056;
template <int LAYOUT>
__global__ void __launch_bounds__(512) writer(float *y)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const long m0 = (long)blockIdx.x * 128;                 // 128 consecutive chunks of one time step (B % 128 == 0)
    const long t = m0 / B, b0 = m0 % B;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int nt = 0; nt < NT; nt++) {
        for (int q = 0; q < 4; q++) {
            const int rr = 32 * wm + 8 * q + (lane >> 3), cc = 32 * wn + 4 * (lane & 7);      // row-segment pattern (TRSTORE)
            float *dst = LAYOUT == 0 ? y + (m0 + rr) * LD + nt * 64 + cc : y + ((t * NT + nt) * B + b0 + rr) * 64 + cc;
            *reinterpret_cast<float4 *>(dst) = v;
        }
        __syncthreads();
    }
}
template <int LAYOUT>
__global__ void __launch_bounds__(256) reader(const float *y, float *sink)
{
    const int b = blockIdx.x, j = threadIdx.x;              // thread j reads columns 4j .. 4j+3 of the chunk's row
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load = [&](long t) {
        const float *src = LAYOUT == 0 ? y + (t * B + b) * LD + 4 * j : y + ((t * NT + (j >> 4)) * B + b) * 64 + 4 * (j & 15);
        return *reinterpret_cast<const float4 *>(src);
    };
    float4 cur = load(0);
    for (long t = 0; t < T; t++) {
        const float4 nxt = load(t + 1 < T ? t + 1 : t);
        acc.x += cur.x; acc.y += cur.y; acc.z += cur.z; acc.w += cur.w;
        __syncthreads();
        cur = nxt;
    }
    if (acc.x == 123.f) sink[b * 256 + j] = acc.y + acc.z + acc.w;
}
template <typename F>
static float timeit(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; i++) f();
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main()
{
    float *y, *sink;
    hipMalloc(&y, sizeof(float) * T * B * 1088);
    hipMalloc(&sink, sizeof(float) * B * 256);
    hipMemset(y, 0, sizeof(float) * T * B * 1088);
    for (int rnd = 0; rnd < 2; rnd++) {
        const float w0 = timeit([&] { hipLaunchKernelGGL(writer<0>, dim3(T * B / 128), dim3(512), 0, 0, y); });
        const float w1 = timeit([&] { hipLaunchKernelGGL(writer<1>, dim3(T * B / 128), dim3(512), 0, 0, y); });
        const float r0 = timeit([&] { hipLaunchKernelGGL(reader<0>, dim3(B), dim3(256), 0, 0, y, sink); });
        const float r1 = timeit([&] { hipLaunchKernelGGL(reader<1>, dim3(B), dim3(256), 0, 0, y, sink); });
        printf("writer: rows %.3f ms  tile-blocked %.3f ms      reader: rows %.3f ms  tile-blocked %.3f ms\n", w0, w1, r0, r1);
    }
    return 0;
}
