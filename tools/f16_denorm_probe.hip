// Does v_mfma_f32_16x16x32_f16 flush fp16 subnormal INPUTS?  Does the float->half conversion produce subnormals?
// (decides how the lo halves of the 3-term split must be scaled; gfx950, hipcc defaults)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, float small)
{
    const _Float16 hs = (_Float16)small;               // conversion result (subnormal if small < 6.1e-5)
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
    a[0] = hs;                                         // A[row][k=8*(lane>>4)] = small
    b[0] = (_Float16)1024.0f;                          // B[k][col] = 1024
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = (float)hs; out[1] = acc[0]; out[2] = (float)hs * 1024.0f * 4.0f; }
}
int main()
{
    float *d, h[3];
    hipMalloc(&d, 12);
    for (float s : {1e-3f, 3e-5f, 1e-6f, 1e-7f}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, s);
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("small=%g  half(small)=%g  mfma=%g  expected=%g\n", s, h[0], h[1], h[2]);
    }
    return 0;
}
