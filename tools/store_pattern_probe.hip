// store_pattern_probe.hip -- how fast can the logits tensor [M][ld] be written with the store patterns the softmax
// projection could use?  (Design input for gemm_rows_f16x3.hip.)
//   hipcc -O3 --offload-arch=gfx950 tools/store_pattern_probe.hip -o tools/_build/store_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

// 128 rows per workgroup, 8 waves as 4 (row groups of 32) x 2 (column halves of a 64-column tile), 17 tiles
// pattern 0: transposed-MFMA layout: lane (r = l%32, h = l/32) stores 16 B at row r, columns 8q + 4h (q = 0..3)
// pattern 1: row-segment layout: lane (er = l/8, ec = l%8) stores 16 B at row 8q + er, columns 4*ec (128-B segments)
// pattern 2: whole rows: each wave owns 16 rows and writes them start to end (64 lanes x 16 B = 1 KiB contiguous)
template <int PATTERN>
__global__ void __launch_bounds__(512) store_probe(float *y, long ld, long M, int N)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long m0 = (long)blockIdx.x * 128;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    const int ntiles = (N + 63) / 64;
    if (PATTERN == 2) {
        for (int rr = 0; rr < 16; rr++) {
            const long row = m0 + wave * 16 + rr;
            if (row >= M) break;
            for (int c = 4 * lane; c + 3 < N; c += 256) *reinterpret_cast<float4 *>(y + row * ld + c) = v;
        }
        return;
    }
    const int wm = wave >> 1, wn = wave & 1;
    for (int nt = 0; nt < ntiles - 1; nt++) {
        for (int q = 0; q < 4; q++) {
            long row;
            int col;
            if (PATTERN == 0) { row = m0 + 32 * wm + (lane & 31); col = nt * 64 + 32 * wn + 8 * q + 4 * (lane >> 5); }
            else { row = m0 + 32 * wm + 8 * q + (lane >> 3); col = nt * 64 + 32 * wn + 4 * (lane & 7); }
            if (row < M) *reinterpret_cast<float4 *>(y + row * ld + col) = v;
        }
        __syncthreads();
    }
}

template <int PATTERN>
static void run(const char *name)
{
    const long M = 819200, ld = 1056;
    const int N = 1025;
    float *y;
    hipMalloc(&y, sizeof(float) * M * ld);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((store_probe<PATTERN>), dim3(M / 128), dim3(512), 0, 0, y, ld, M, N);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((store_probe<PATTERN>), dim3(M / 128), dim3(512), 0, 0, y, ld, M, N);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("%-46s %.3f ms  %.0f GB/s\n", name, ms, 4.0 * M * 1024 / ms / 1e6);
    hipFree(y);
}

int main()
{
    run<0>("transposed-MFMA lanes (32 rows x 32 B / instr)");
    run<1>("row segments (8 rows x 128 B / instr)");
    run<2>("whole rows (1 KiB contiguous / instr)");
    return 0;
}
