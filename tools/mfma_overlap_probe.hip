// Does a wave's own VALU work overlap its MFMAs?  One wave per SIMD (256 threads, 1 block per CU), s_memtime around loops of
//   0: 64 independent MFMAs                       1: 64 x (MFMA + 3 independent v_fma)      2: 64 x (MFMA + 3 v_accvgpr_read-like moves)
//   3: 192 v_fma alone                            4: 64 dependent MFMAs (same accumulator)  5: 64 x (MFMA + 1 v_exp)
//   6: 64 x (MFMA + 6 v_fma)                      7: MFMAs alternating two accumulators
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define FMA(x, y) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y))
#define FMA2(x, y) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(x) : "v"(y))
#define MUL(x, y) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define RCP(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define AREAD(x, a) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
template <int mode>
__global__ void __launch_bounds__(256, 1) probe(unsigned long long *out, float *sink, int reps)
{
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(threadIdx.x * 0.001f + j); b[j] = (_Float16)(j * 0.5f); }
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v0 = threadIdx.x, v1 = 1.5f, v2 = 2.5f, v3 = 0.25f, v4 = 3.f, v5 = 4.f, y = 0.999f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (mode == 0) { MFMA(acc[i], a, b); }
            else if constexpr (mode == 1) { MFMA(acc[i], a, b); FMA(v0, y); FMA(v1, y); FMA(v2, y); }
            else if constexpr (mode == 2) { MFMA(acc[i], a, b); FMA(v0, y); FMA(v1, y); }
            else if constexpr (mode == 3) { FMA(v0, y); FMA(v1, y); FMA(v2, y); }
            else if constexpr (mode == 4) { MFMA(acc[0], a, b); }
            else if constexpr (mode == 5) { MFMA(acc[i], a, b); EXP(v0); }
            else if constexpr (mode == 6) { MFMA(acc[i], a, b); FMA(v0, y); FMA(v1, y); FMA(v2, y); FMA(v3, y); FMA(v4, y); FMA(v5, y); }
            else if constexpr (mode == 7) { MFMA(acc[i & 1], a, b); }
            else if constexpr (mode == 8) { MFMA(acc[i], a, b); FMA(v0, y); }
            else if constexpr (mode == 10) { MUL(v0, y); }
            else if constexpr (mode == 11) { MUL(v0, y); MUL(v1, y); }
            else if constexpr (mode == 12) { EXP(v0); }
            else if constexpr (mode == 13) { EXP(v0); EXP(v1); }
            else if constexpr (mode == 14) { MUL(v0, y); MUL(v1, y); MUL(v2, y); MUL(v3, y); }
            else if constexpr (mode == 15) { MFMA(acc[i], a, b); MUL(v0, y); MUL(v1, y); MUL(v2, y); }
            else if constexpr (mode == 16) { MFMA(acc[i], a, b); MUL(v0, y); MUL(v0, y); MUL(v0, y); }
            else if constexpr (mode == 17) { MFMA(acc[i], a, b); MUL(v0, y); EXP(v0); RCP(v0); }
            else if constexpr (mode == 18) { MUL(v0, y); EXP(v0); RCP(v0); }
            else if constexpr (mode == 9) { MFMA(acc[i], a, b); FMA(v0, y); FMA(v1, y); FMA(v2, y); FMA(v3, y); }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = v0 + v1 + v2 + v3 + v4 + v5;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
int main()
{
    unsigned long long *out; float *sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 256 * 4);
    const char *names[] = {"8 independent MFMAs", "8 x (MFMA + 3 fma)", "8 x (MFMA + 2 fma)", "24 fma alone", "8 dependent MFMAs", "8 x (MFMA + exp)", "8 x (MFMA + 6 fma)", "MFMAs alternating 2 accumulators", "8 x (MFMA + 1 fma)", "8 x (MFMA + 4 fma)", "8 dependent v_mul (1 chain)", "16 v_mul (2 chains)", "8 dependent v_exp", "16 v_exp (2 chains)", "32 v_mul (4 chains)", "8 x (MFMA + 3 independent mul)", "8 x (MFMA + 3 dependent mul)", "8 x (MFMA + mul,exp,rcp dependent)", "8 x (mul,exp,rcp dependent)"};
    const int reps = 1000;
    void (*kern[])(unsigned long long *, float *, int) = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>, probe<11>, probe<12>, probe<13>, probe<14>, probe<15>, probe<16>, probe<17>, probe<18>};
    for (int mode = 0; mode < 19; mode++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern[mode], dim3(256), dim3(256), 0, 0, out, sink, reps);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern[mode], dim3(256), dim3(256), 0, 0, out, sink, reps * 10);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[4];
        hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        printf("%-36s  %.1f memtime ticks per loop body of 8\n", names[mode], (double)h[0] / (reps * 10.0)); 
    }
    // memtime tick rate vs wall clock
    return 0;
}
