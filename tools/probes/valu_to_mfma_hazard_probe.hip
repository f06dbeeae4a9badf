// Does gfx950 interlock a vector write of a register against an MFMA that reads it as its B operand when both sit in inline asm
// (where hipcc counts no wait states)?  The B operand first holds `stale`, is overwritten with `fresh` by v_mov_b32 x4 (KIND 0),
// v_mov_b64 x2 (KIND 1) or v_pk_mul_f32 x2 (KIND 2), and N wait states later the MFMA runs: a product of `stale` in the result means the
// write had not landed.
//   hipcc -O3 --offload-arch=gfx950 -w tools/probes/valu_to_mfma_hazard_probe.hip -o /tmp/probe2 && /tmp/probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// (fixed registers: pairs cannot be named through operand numbers)
template <int N, int KIND>
__global__ void probe_fixed(const half8 *a, const u32x4 *stale, const u32x4 *fresh, float *out)
{
    const half8 av = a[threadIdx.x];
    const u32x4 s = stale[threadIdx.x], f = fresh[threadIdx.x];
    f32x4 acc;
    // v[40:43] = B operand, v[44:47] = fresh, v[48:51] = A, v[52:55] = result
    asm volatile("s_waitcnt vmcnt(0)\n\t"
                 "v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
                 "v_mov_b32 v44, %5\n\tv_mov_b32 v45, %6\n\tv_mov_b32 v46, %7\n\tv_mov_b32 v47, %8\n\t"
                 "v_mov_b32 v48, %9\n\tv_mov_b32 v49, %10\n\tv_mov_b32 v50, %11\n\tv_mov_b32 v51, %12\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 ".if %c14 == 0\n\t"
                 "v_mov_b32 v40, v44\n\tv_mov_b32 v41, v45\n\tv_mov_b32 v42, v46\n\tv_mov_b32 v43, v47\n\t"
                 ".elseif %c14 == 1\n\t"
                 "v_mov_b64 v[40:41], v[44:45]\n\tv_mov_b64 v[42:43], v[46:47]\n\t"
                 ".else\n\t"
                 "v_pk_mul_f32 v[40:41], v[44:45], 1.0 op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[42:43], v[46:47], 1.0 op_sel_hi:[1,0]\n\t"
                 ".endif\n\t"
                 ".rept %c13\n\ts_nop 0\n\t.endr\n\t"
                 "v_mfma_f32_16x16x32_f16 v[52:55], v[48:51], v[40:43], 0\n\t"
                 "s_nop 7\n\ts_nop 7\n\t"
                 "v_mov_b32 %0, v52"
                 : "=v"(acc[0])
                 : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]),
                   "v"(((const unsigned *)&av)[0]), "v"(((const unsigned *)&av)[1]), "v"(((const unsigned *)&av)[2]), "v"(((const unsigned *)&av)[3]),
                   "n"(N), "n"(KIND)
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    out[threadIdx.x] = acc[0];
}

template <int N, int KIND>
static void run(const half8 *a, const u32x4 *s, const u32x4 *f, float *o, const std::vector<float> &want)
{
    hipLaunchKernelGGL((probe_fixed<N, KIND>), dim3(1), dim3(64), 0, 0, a, s, f, o);
    std::vector<float> h(64);
    hipMemcpy(h.data(), o, 256, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int i = 0; i < 64; i++) ok += h[i] == want[i];
    printf("%s, %d wait states: %2d of 64 lanes hold the product of the fresh operand (lane 0: %g, fresh %g)\n",
           KIND == 0 ? "v_mov_b32 x4" : KIND == 1 ? "v_mov_b64 x2" : "v_pk_mul_f32 x2", N, ok, h[0], want[0]);
}

int main()
{
    half8 *a;
    u32x4 *s, *f;
    float *o;
    hipMalloc(&a, 1024); hipMalloc(&s, 1024); hipMalloc(&f, 1024); hipMalloc(&o, 256);
    std::vector<_Float16> ha(512), hs(512), hf(512);
    for (int i = 0; i < 512; i++) { ha[i] = (_Float16)(0.25f + 0.001f * i); hs[i] = (_Float16)3.0f; hf[i] = (_Float16)1.0f; }
    hipMemcpy(a, ha.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(s, hs.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(f, hf.data(), 1024, hipMemcpyHostToDevice);
    // reference: the fresh operand written long before
    std::vector<float> want(64);
    hipLaunchKernelGGL((probe_fixed<12, 0>), dim3(1), dim3(64), 0, 0, a, s, f, o);
    hipMemcpy(want.data(), o, 256, hipMemcpyDeviceToHost);
#define R(N) run<N, 0>(a, s, f, o, want); run<N, 1>(a, s, f, o, want);
    R(0) R(1) R(2) R(3) R(4)
    return 0;
}
