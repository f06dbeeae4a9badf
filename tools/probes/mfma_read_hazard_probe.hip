// How many wait states does gfx950 need between v_mfma_f32_16x16x32_f16 and a VALU read of its destination when the read sits in
// inline asm (where hipcc inserts nothing -- the situation of pick_mix in csrc/bar16_common.h)?  For N = 0..12 wait states (s_nop 0
// repeated N times) and PRE = 0 / 3 MFMAs queued in front: lanes whose four reads all returned the finished product.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form tools/probes/mfma_read_hazard_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int PRE>
__global__ void probe(const half8 *a, const half8 *b, float *early, float *late)
{
    const half8 av = a[threadIdx.x], bv = b[threadIdx.x];
    f32x4 pre = {0.f, 0.f, 0.f, 0.f}, acc = {-1.f, -1.f, -1.f, -1.f};
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc), "+v"(pre) : "v"(av), "v"(bv));
#pragma unroll
    for (int i = 0; i < PRE; i++) pre = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, pre, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    float e0, e1, e2, e3;
    asm volatile(".rept %c8\n\ts_nop 0\n\t.endr\n\t"
                 "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
                 : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "n"(N));
    float *ep = early + 4 * threadIdx.x, *lp = late + 4 * threadIdx.x;
    ep[0] = e0; ep[1] = e1; ep[2] = e2; ep[3] = e3;
    lp[0] = acc[0] + pre[0] * 0.f; lp[1] = acc[1]; lp[2] = acc[2]; lp[3] = acc[3];
}

template <int N, int PRE>
static void run(const half8 *a, const half8 *b, float *e, float *l)
{
    hipLaunchKernelGGL((probe<N, PRE>), dim3(1), dim3(64), 0, 0, a, b, e, l);
    std::vector<float> he(256), hl(256);
    hipMemcpy(he.data(), e, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), l, 1024, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < 256; i++) ok += he[i] == hl[i];
    printf("PRE=%d N=%2d wait states: %3d of 256 values read finished (late[0]=%g early[0]=%g)\n", PRE, N, ok, hl[0], he[0]);
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
#define R(N) run<N, 0>(a, b, e, l); run<N, 3>(a, b, e, l);
    R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(12)
    return 0;
}
