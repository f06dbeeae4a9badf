// The two cases csrc/gru_scan1t.hip's n = 144 path relies on, which tools/probes/mfma_read_hazard_probe.hip (v_mfma_f32_16x16x32_f16 only)
// does not cover:
//   A  v_mfma_f32_16x16x16_f16 followed by N wait states and a VALU read of its destination from inline asm;
//   B  v_mfma_f32_16x16x32_f16 followed by N wait states and a v_mfma_f32_16x16x16_f16 that takes its destination as SrcC (both in
//      inline asm: hipcc inserts nothing between them).
// For N = 0..12: how many of the 256 values come out as with a full drain.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_read_hazard_probe2.hip -o /tmp/probe2 && /tmp/probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ void probe_a(const half8 *a, const half8 *b, float *early, float *late)
{
    const half8 av8 = a[threadIdx.x], bv8 = b[threadIdx.x];
    const half4 av = {av8[0], av8[1], av8[2], av8[3]}, bv = {bv8[0], bv8[1], bv8[2], bv8[3]};
    f32x4 acc = {-1.f, -1.f, -1.f, -1.f};
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc) : "v"(av), "v"(bv));
    float e0, e1, e2, e3;
    asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(av), "v"(bv));
    // (two asm statements: hipcc does not know the first one is an MFMA and pads nothing in front of the second)
    asm volatile(".rept %c8\n\ts_nop 0\n\t.endr\n\t"
                 "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
                 : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "n"(N));
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
    float *ep = early + 4 * threadIdx.x, *lp = late + 4 * threadIdx.x;
    ep[0] = e0; ep[1] = e1; ep[2] = e2; ep[3] = e3;
    lp[0] = acc[0]; lp[1] = acc[1]; lp[2] = acc[2]; lp[3] = acc[3];
}

template <int N>
__global__ void probe_b(const half8 *a, const half8 *b, float *early, float *late)
{
    const half8 av8 = a[threadIdx.x], bv8 = b[threadIdx.x];
    const half4 av = {av8[4], av8[5], av8[6], av8[7]}, bv = {bv8[4], bv8[5], bv8[6], bv8[7]};
    f32x4 acc = {-1.f, -1.f, -1.f, -1.f}, ref = {-1.f, -1.f, -1.f, -1.f};
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc), "+v"(ref) : "v"(av8), "v"(bv8), "v"(av), "v"(bv));
    // the pair under test: wide MFMA, N wait states, narrow MFMA accumulating onto it
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                 ".rept %c5\n\ts_nop 0\n\t.endr\n\t"
                 "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\t"
                 "s_nop 7\n\ts_nop 7"
                 : "+v"(acc)
                 : "v"(av8), "v"(bv8), "v"(av), "v"(bv), "n"(N));
    // the same pair with a full drain in between
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\t"
                 "s_nop 7\n\ts_nop 7"
                 : "+v"(ref)
                 : "v"(av8), "v"(bv8), "v"(av), "v"(bv));
    float *ep = early + 4 * threadIdx.x, *lp = late + 4 * threadIdx.x;
    ep[0] = acc[0]; ep[1] = acc[1]; ep[2] = acc[2]; ep[3] = acc[3];
    lp[0] = ref[0]; lp[1] = ref[1]; lp[2] = ref[2]; lp[3] = ref[3];
}

template <class K>
static void run(K kern, const char *what, int n, const half8 *a, const half8 *b, float *e, float *l)
{
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, a, b, e, l);
    std::vector<float> he(256), hl(256);
    hipMemcpy(he.data(), e, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), l, 1024, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < 256; i++) ok += he[i] == hl[i];
    printf("%s N=%2d wait states: %3d of 256 values as after a full drain (drained[0]=%g early[0]=%g)\n", what, n, ok, hl[0], he[0]);
}

int main()
{
    half8 *a, *b;
    float *e, *l;
    hipMalloc(&a, 1024); hipMalloc(&b, 1024); hipMalloc(&e, 1024); hipMalloc(&l, 1024);
    std::vector<_Float16> h(512);
    for (int i = 0; i < 512; i++) h[i] = (_Float16)(0.25f + 0.001f * i);
    hipMemcpy(a, h.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(b, h.data(), 1024, hipMemcpyHostToDevice);
#define R(N) run(probe_a<N>, "A 16x16x16 -> VALU read   ", N, a, b, e, l);
    R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(10) R(12)
#undef R
#define R(N) run(probe_b<N>, "B 16x16x32 -> 16x16x16 SrcC", N, a, b, e, l);
    R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(10) R(12)
    return 0;
}
