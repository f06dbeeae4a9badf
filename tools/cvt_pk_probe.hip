// v_cvt_pk_f16_f32 vs v_cvt_f16_f32 on gfx950: do they ever disagree?  (hipcc uses both for "(_Float16)x" depending on context)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__global__ void probe(unsigned long long *count, unsigned *examples, unsigned seed_base, int mode)
{
    unsigned long long bad = 0;
    unsigned s = seed_base + blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < 4096; it++) {
        s = s * 1664525u + 1013904223u;
        unsigned bits = s;
        if (mode == 0) bits = (s & 0x807fffffu) | ((100u + ((s >> 23) & 31u)) << 23);   // exponents 2^-27 .. 2^4
        float v = __uint_as_float(bits);
        unsigned pk;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(v));
        unsigned short single;
        asm volatile("v_cvt_f16_f32_e32 %0, %1" : "=v"(single) : "v"(v));
        if ((pk & 0xffffu) != (unsigned)single || (pk >> 16) != (unsigned)single) {
            bad++;
            unsigned slot = atomicAdd(&examples[0], 1u);
            if (slot < 8) { examples[1 + 3 * slot] = bits; examples[2 + 3 * slot] = pk; examples[3 + 3 * slot] = single; }
        }
    }
    atomicAdd(count, bad);
}
int main()
{
    unsigned long long *d, h;
    unsigned *ex, hex[32];
    (void)hipMalloc(&d, 8);
    (void)hipMalloc(&ex, sizeof(hex));
    for (int mode = 0; mode < 2; mode++) {
        (void)hipMemset(d, 0, 8);
        (void)hipMemset(ex, 0, sizeof(hex));
        hipLaunchKernelGGL(probe, dim3(1024), dim3(256), 0, 0, d, ex, 12345u, mode);
        (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hex, ex, sizeof(hex), hipMemcpyDeviceToHost);
        printf("mode %d: %llu mismatches of %llu\n", mode, h, 1024ull * 256 * 4096);
        for (unsigned i = 0; i < hex[0] && i < 8; i++) {
            float v; unsigned b = hex[1 + 3 * i]; memcpy(&v, &b, 4);
            printf("   v=%.9g (0x%08x)  pk=0x%08x  single=0x%04x\n", v, b, hex[2 + 3 * i], hex[3 + 3 * i]);
        }
    }
    return 0;
}
