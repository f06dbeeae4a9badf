// coissue_probe.hip -- what do the TWO waves of a SIMD overlap?  One 512-thread workgroup on one CU: waves 0-3 (one per SIMD) run
// body X, their SIMD partners (waves 4-7) run body Y, each wave times its own loop with s_memtime.  Printed: cycles per
// instruction of X alone (partner idle), of Y alone, and of both when they run side by side.  If two bodies overlap perfectly
// the side-by-side figures equal the alone figures; if the SIMD issues ONE instruction per slot whoever it comes from they add.
// (Design input for softmax_viterbi.hip, whose two waves per SIMD run the same program in lock step.)
//
//   hipcc -O3 --offload-arch=gfx950 tools/probes/coissue_probe.hip -o /tmp/coissue_probe && /tmp/coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

enum Body { IDLE = 0, VALU, VALU_DEP, SALU, CMPSEL, TRANS, PK, LDSR, LDSW, NOP, MAXDPP, MFMA, MFMA7, MFMA14, NBODY };
static const char *names[NBODY] = {"idle", "valu(indep)", "valu(dep)", "salu", "cmp+cndmask", "trans", "pk_fma", "ds_read_b32",
                                   "ds_write_b32", "s_nop", "max_dpp", "mfma32x32x16", "mfma+7valu", "mfma+14valu"};
// instructions per loop iteration of each body (for the per-instruction figure)
static const int per_iter[NBODY] = {1, 64, 64, 64, 64, 64, 64, 32, 32, 64, 32, 8, 8, 8};      // (MFMA bodies: per MFMA, its fillers included)

template <int BODY>
__device__ __forceinline__ void body(float (&v)[8], float &lds_v, float *lds, int lane, unsigned &sacc)
{
    if constexpr (BODY == VALU) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(lds_v));
    } else if constexpr (BODY == VALU_DEP) {
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(lds_v));
    } else if constexpr (BODY == SALU) {
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) : : "scc");
    } else if constexpr (BODY == CMPSEL) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++)
                asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(lds_v) : "vcc");
    } else if constexpr (BODY == TRANS) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (BODY == PK) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 p[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
        f2 m = {lds_v, lds_v};
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(m));
#pragma unroll
        for (int i = 0; i < 4; i++) { v[2 * i] = p[i].x; v[2 * i + 1] = p[i].y; }
    } else if constexpr (BODY == LDSR) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[i]) : "v"(lane * 4), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (BODY == LDSW) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("ds_write_b32 %1, %0" ::"v"(v[i]), "v"(lane * 4) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if constexpr (BODY == NOP) {
#pragma unroll
        for (int r = 0; r < 64; r++) asm volatile("s_nop 0");
    } else if constexpr (BODY == MFMA || BODY == MFMA7 || BODY == MFMA14) {
        // eight dependent v_mfma_f32_32x32x16_f16 (one accumulator), with 0 / 7 / 14 independent vector instructions behind each
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        typedef float f16v __attribute__((ext_vector_type(16)));
        static_assert(sizeof(h8) == 16, "");
        h8 a, b;
        for (int i = 0; i < 8; i++) { a[i] = (_Float16)v[i]; b[i] = (_Float16)lds_v; }
        f16v acc;
        for (int i = 0; i < 16; i++) acc[i] = v[i & 7];
        asm volatile("" : "+v"(acc), "+v"(a), "+v"(b));
#pragma unroll
        for (int r = 0; r < 8; r++) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            constexpr int NF = BODY == MFMA ? 0 : (BODY == MFMA7 ? 7 : 14);
#pragma unroll
            for (int i = 0; i < NF; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(lds_v));
        }
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
        v[0] += acc[0];
    } else if constexpr (BODY == MAXDPP) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++)
                asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
    }
}

template <int X, int Y>
__global__ void __launch_bounds__(512) probe(float *out, unsigned long long *cycles, int iters, int mode)
{
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    lds[tid] = 1.0f + tid * 1e-6f;
    lds[tid + 512] = 0.5f;
    __syncthreads();
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = lds[(tid + i) & 1023];
    float lv = lds[tid + 512] * 1e-9f;
    unsigned sacc = __builtin_amdgcn_readfirstlane(wave);
    // mode 0: both; 1: only waves 0-3 run (X alone); 2: only waves 4-7 run (Y alone)
    const bool first = wave < 4;
    const bool active = mode == 0 || (mode == 1 && first) || (mode == 2 && !first);
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (active) {
        t0 = __builtin_amdgcn_s_memtime();
        if (first) {
            for (int it = 0; it < iters; it++) body<X>(v, lv, lds, lane, sacc);
        } else {
            for (int it = 0; it < iters; it++) body<Y>(v, lv, lds, lane, sacc);
        }
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += v[i];
    out[tid] = s + (float)sacc;
    if (lane == 0) cycles[wave] = t1 - t0;
}

template <int X, int Y>
static void run(float *out, unsigned long long *cyc)
{
    const int iters = 500;
    double res[3][2];
    for (int mode = 0; mode < 3; mode++) {
        unsigned long long h[8];
        hipLaunchKernelGGL((probe<X, Y>), dim3(1), dim3(512), 0, 0, out, cyc, iters, mode);
        hipLaunchKernelGGL((probe<X, Y>), dim3(1), dim3(512), 0, 0, out, cyc, iters, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        res[mode][0] = (double)h[0] / iters / per_iter[X];
        res[mode][1] = (double)h[4] / iters / per_iter[Y];
    }
    printf("X = %-13s Y = %-13s | alone: X %6.2f  Y %6.2f cycles/instr | side by side: X %6.2f  Y %6.2f\n", names[X], names[Y],
           res[1][0], res[2][1], res[0][0], res[0][1]);
    fflush(stdout);
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 64);
    run<VALU, VALU>(out, cyc);
    run<VALU_DEP, VALU_DEP>(out, cyc);
    run<VALU, VALU_DEP>(out, cyc);
    run<VALU, SALU>(out, cyc);
    run<SALU, SALU>(out, cyc);
    run<VALU, CMPSEL>(out, cyc);
    run<CMPSEL, CMPSEL>(out, cyc);
    run<VALU, TRANS>(out, cyc);
    run<TRANS, TRANS>(out, cyc);
    run<VALU, PK>(out, cyc);
    run<PK, PK>(out, cyc);
    run<VALU, LDSR>(out, cyc);
    run<LDSR, LDSR>(out, cyc);
    run<VALU, LDSW>(out, cyc);
    run<VALU, NOP>(out, cyc);
    run<NOP, NOP>(out, cyc);
    run<VALU, MAXDPP>(out, cyc);
    run<MAXDPP, MAXDPP>(out, cyc);
    run<SALU, CMPSEL>(out, cyc);
    run<MFMA, MFMA>(out, cyc);
    run<MFMA7, MFMA7>(out, cyc);
    run<MFMA14, MFMA14>(out, cyc);
    run<MFMA7, VALU>(out, cyc);
    run<MFMA14, VALU>(out, cyc);
    run<VALU, MFMA7>(out, cyc);
    return 0;
}
