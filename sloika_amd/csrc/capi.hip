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
