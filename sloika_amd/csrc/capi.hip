// capi.hip -- ABI bookkeeping entry points of libsloika_amd.so (see include/sloika_amd.h).
#include "common.h"

extern "C" int slk_abi_version(void) { return SLK_ABI_VERSION; }

extern "C" const char *slk_error_string(int code)
{
    switch (code) {
    case SLK_OK: return "ok";
    case SLK_ERR_INVALID_ARG: return "invalid argument (shape / null pointer)";
    case SLK_ERR_UNSUPPORTED: return "unsupported configuration for this build";
    case SLK_ERR_LAUNCH: return "HIP kernel launch failed";
    case SLK_ERR_WORKSPACE: return "workspace missing or too small";
    case SLK_ERR_NO_DEVICE: return "no HIP device visible";
    default: return "unknown error code";
    }
}

extern "C" int slk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Shader clock the device holds right now (include/sloika_amd.h): one wave reads the shader-cycle counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime), sleeps, reads both again.
__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long *out, int spins)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < spins; i++) __builtin_amdgcn_s_sleep(127);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
}

extern "C" int slk_clock_probe(unsigned long long *out2, int spins, slk_stream_t stream)
{
    if (!out2 || spins < 1 || spins > 100000) return SLK_ERR_INVALID_ARG;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, slk_stream(stream), out2, spins);
    return slk_launch_status();
}
