// lds_flags.h -- progress counters in LDS instead of s_barrier (used by the persistent recurrent kernels).
//
// LDS operations of ONE wave execute in program order, so "data write, then counter write" on the producer and "counter
// read, then data read" on the consumer is a release/acquire pair without any fence; the consumer issues its data reads
// together with the counter read (one LDS round trip per exchange) and retries in the rare case the counter was not there.
#pragma once
#include "common.h"

typedef __attribute__((address_space(3))) int lds_int_t;

__device__ __forceinline__ void publish(int *flags, int idx, int value, int lane)
{
    asm volatile("" ::: "memory");          // the data writes stay ahead of the counter write in program order
    if (lane == 0) *(volatile lds_int_t *)(lds_int_t *)&flags[idx] = value;
    asm volatile("" ::: "memory");
}
// Polling is split in two so that the consumer's data reads travel with the counter read (one LDS round trip):
//   poll_issue  -- ds_read of the watched counter (lane l watches flags[l & (NC-1)], NC = 16 or 32 counters), NOT waited for
//   ... the caller issues its data reads ...
//   poll_result -- waits for everything and tells whether every watched counter had reached the lane's `need`
// (LDS executes a wave's operations in order, so data read after a counter that had arrived is valid data).
template <int NC = 16>
__device__ __forceinline__ int poll_issue(const int *flags, int lane)
{
    int v;
    const unsigned addr = (unsigned)(uintptr_t)(lds_int_t *)&flags[lane & (NC - 1)];
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ bool poll_result(int v, int need)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v)::"memory");
    return __builtin_amdgcn_ballot_w64(v < need) == 0;
}
template <int NC = 16>
__device__ __forceinline__ bool reached(const int *flags, int lane, int need)
{
    return poll_result(poll_issue<NC>(flags, lane), need);
}
// keeps values loaded inside a retry loop from being sunk out of it
__device__ __forceinline__ void keep(f32x4 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep(float &v) { asm volatile("" : "+v"(v)); }
