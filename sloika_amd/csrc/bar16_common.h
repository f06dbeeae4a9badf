// bar16_common.h -- helpers shared by the barrier-stepped Gru kernels (gru_bar16.hip: four chunks per workgroup,
// gru_bar16d.hip: eight): barriers that wait for LDS the cheap way, the five-instruction tanh, 3-term split MFMA sequences and
// the projection's MFMAs with weights named in accumulation registers.
#pragma once
#include <type_traits>

#include "f16split.h"

template <bool REAL = true>
__device__ __forceinline__ void lds_bar()
{
    if constexpr (REAL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// barrier for a wave whose two youngest LDS operations are reads of its OWN data: LDS executes a wave's operations in
// order, so everything older -- the writes the other waves are waiting for -- has been performed once at most two remain
template <bool REAL = true>
__device__ __forceinline__ void lds_bar_2reads()
{
    if constexpr (REAL) asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
}
__device__ __forceinline__ void lds_fence() { asm volatile("" ::: "memory"); }
// tanh through one exp, 1 - 2/(exp(2x)+1), written so that it is five instructions (the same values as slk_tanh: 2x and
// 2r are exact)
__device__ __forceinline__ float tanh5(float x)
{
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
__device__ __forceinline__ float sigmoid4(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// Row 4g+q of a tile's accumulator for the lane (g, q, c) when only column group 0 (lanes q = 0) holds the product: lane
// quartet q of every 16-lane row takes register q of the lanes four, eight, twelve places below it (bank-masked DPP moves
// -- three instructions, as many as the three selects they replace).
__device__ __forceinline__ float gather4(const f32x4 &a)
{
    int r = __float_as_int(a[0]);
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(a[1]), 0x114, 0xf, 0x2, false);      // row_shr:4  -> quartet 1
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(a[2]), 0x118, 0xf, 0x4, false);      // row_shr:8  -> quartet 2
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(a[3]), 0x11c, 0xf, 0x8, false);      // row_shr:12 -> quartet 3
    return __int_as_float(r);
}

template <int V>
using ic = std::integral_constant<int, V>;
template <int B_, int E_, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B_ < E_) {
        f(ic<B_>{});
        static_for<B_ + 1, E_>(f);
    }
}

// acc0 += W0.h, acc1 += W1.h as 3-term splits with the two accumulation chains interleaved: consecutive MFMAs never depend on
// each other (per accumulator the order of the terms is that of mfma3)
__device__ __forceinline__ void mfma3x2(const half8 &w0_hi, const half8 &w0_lo, const half8 &w1_hi, const half8 &w1_lo,
                                        const half8 &h_hi, const half8 &h_lo, f32x4 &acc0, f32x4 &acc1)
{
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0_hi, h_lo, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1_hi, h_lo, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0_lo, h_hi, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1_lo, h_hi, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0_hi, h_hi, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1_hi, h_hi, acc1, 0, 0, 0);
}
// the same for NT tiles of the projection and K block kb of their weights
template <int NT, int KBLK_>
__device__ __forceinline__ void mfma3xn(const half8 (*w_hi)[KBLK_], const half8 (*w_lo)[KBLK_], int kb, const half8 &x_hi,
                                        const half8 &x_lo, f32x4 *acc)
{
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[t][kb], x_lo, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo[t][kb], x_hi, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi[t][kb], x_hi, acc[t], 0, 0, 0);
}

// Projection weights in ACCUMULATION registers (a wave alone on its SIMD has 256 of them next to its 256 ordinary ones), named
// directly as the MFMA's A operand.  hipcc treats those registers as spill space and copies every operand back (four
// v_accvgpr_read per operand and use); operands only ever used through an "a" constraint stay where they are.
__device__ __forceinline__ half8 to_acc_regs(half8 v)
{
    half8 a;
    asm volatile("" : "=a"(a) : "0"(v));
    return a;
}
// a whole tile: acc = sum over K blocks of the 3-term split, first MFMA with a zero C operand (no VALU write of the accumulator
// in front of an instruction the compiler does not know to be an MFMA), then let the matrix pipe drain before ordinary
// instructions read the result (the compiler's hazard bookkeeping does not see asm)
template <int KBLK_>
__device__ __forceinline__ f32x4 tile_mfma_acc(const half8 *w_hi, const half8 *w_lo, const half8 *x_hi, const half8 *x_lo)
{
    f32x4 acc;
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w_hi[0]), "v"(x_lo[0]));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_lo[0]), "v"(x_hi[0]));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_hi[0]), "v"(x_hi[0]));
#pragma unroll
    for (int kb = 1; kb < KBLK_; kb++) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_hi[kb]), "v"(x_lo[kb]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_lo[kb]), "v"(x_hi[kb]));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_hi[kb]), "v"(x_hi[kb]));
    }
    return acc;
}
// (Eight wait states: what hipcc itself puts between a v_mfma_f32_16x16x32_f16 and a vector instruction that reads its result; the
//  hardware does not interlock that read, and seven is where every register arrives -- tools/probes/mfma_read_hazard_probe.hip.)
// ONLY behind v_mfma_f32_16x16x32_f16 (a four-pass MFMA: the figure is that instruction's); a 32x32 tile (eight or sixteen passes) needs
// more -- f32x4 in the signature is the guard: a sixteen-register accumulator does not convert.
__device__ __forceinline__ void mfma_drain(f32x4 &a) { asm volatile("s_nop 7" : "+v"(a)); }
__device__ __forceinline__ void mfma_drain2(f32x4 &a, f32x4 &b) { asm volatile("s_nop 7" : "+v"(a), "+v"(b)); }
// one K block of one tile, accumulator kept across steps (chain waves); FIRST: start from zero
template <bool FIRST>
__device__ __forceinline__ void block_mfma_acc(f32x4 &acc, const half8 &w_hi, const half8 &w_lo, const half8 &x_hi, const half8 &x_lo)
{
    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(w_hi), "v"(x_lo));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_hi), "v"(x_lo));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_lo), "v"(x_hi));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(w_hi), "v"(x_hi));
}

// one tile's product with one K block, three terms (per accumulator the order of mfma3), weights in accumulation registers
template <bool FIRST>
__device__ __forceinline__ void tile_block_mfma(f32x4 &a, const half8 &w_hi, const half8 &w_lo, const half8 &bh, const half8 &bl)
{
    if constexpr (FIRST) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a) : "a"(w_hi), "v"(bl));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a) : "a"(w_hi), "v"(bl));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a) : "a"(w_lo), "v"(bh));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a) : "a"(w_hi), "v"(bh));
}
// z products of one K block for the wave's two tiles, weights in accumulation registers, the two accumulation chains interleaved
// (per accumulator the order of the terms is that of mfma3)
template <bool FIRST>
__device__ __forceinline__ void z_block_mfma(f32x4 &a0, f32x4 &a1, const half8 &w0_hi, const half8 &w0_lo, const half8 &w1_hi,
                                             const half8 &w1_lo, const half8 &bh, const half8 &bl)
{
    if constexpr (FIRST) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a0) : "a"(w0_hi), "v"(bl));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a1) : "a"(w1_hi), "v"(bl));
    } else {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_hi), "v"(bl));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_hi), "v"(bl));
    }
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_lo), "v"(bh));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_lo), "v"(bh));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_hi), "v"(bh));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_hi), "v"(bh));
}
// the same with hook(ic<BASE + i>) after MFMA i = 0..5: pieces of the wave's vector work that fill the issue gaps (a wave whose
// MFMAs follow each other directly waits 16 cycles per instruction for the pipe)
template <bool FIRST, int BASE, class F>
__device__ __forceinline__ void z_block_mfma_hooked(f32x4 &a0, f32x4 &a1, const half8 &w0_hi, const half8 &w0_lo, const half8 &w1_hi,
                                                    const half8 &w1_lo, const half8 &bh, const half8 &bl, F &&hook)
{
    if constexpr (FIRST) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a0) : "a"(w0_hi), "v"(bl));
        hook(ic<BASE + 0>{});
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a1) : "a"(w1_hi), "v"(bl));
        hook(ic<BASE + 1>{});
    } else {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_hi), "v"(bl));
        hook(ic<BASE + 0>{});
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_hi), "v"(bl));
        hook(ic<BASE + 1>{});
    }
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_lo), "v"(bh));
    hook(ic<BASE + 2>{});
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_lo), "v"(bh));
    hook(ic<BASE + 3>{});
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_hi), "v"(bh));
    hook(ic<BASE + 4>{});
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_hi), "v"(bh));
    hook(ic<BASE + 5>{});
}

// ---- two MFMAs per product instead of three: the state's hi and lo halves in DIFFERENT column groups --------------------------
// With four chunks per workgroup the 16 columns of the recurrent MFMAs hold four copies of every chunk (lane (g, q, c) reads
// column 4q + c, the copies differ in q).  Let the copies q = 0, 1 carry the hi half of the state and q = 2, 3 the lo half (one
// operand fetch per K block instead of two: the lane picks its image by q): then W_lo.B and W_hi.B give, in a hi column,
// lo.hi + hi.hi and, in a lo column, lo.lo + hi.lo -- all four terms of (W_hi + W_lo)(h_hi + h_lo) from TWO instructions -- and the
// product for row 4g+j of chunk c is the sum of register j over the columns (c, q) and (c, q ^ 2).  pick_mix does that sum and the
// row selection at once: four v_add_f32 whose first source comes from the lane eight columns away (row_ror:8) and whose bank mask
// lets only the lane quartet q = j write register j's sum -- one instruction more than sel4's three selects.
__device__ __forceinline__ void mfma2x2(const half8 &w0_hi, const half8 &w0_lo, const half8 &w1_hi, const half8 &w1_lo, const half8 &bm,
                                        f32x4 &acc0, f32x4 &acc1)
{
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0_lo, bm, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1_lo, bm, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0_hi, bm, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1_hi, bm, acc1, 0, 0, 0);
}
// the same with the weights in accumulation registers (asm: see z_block_mfma)
template <bool FIRST>
__device__ __forceinline__ void z_block_mfma2(f32x4 &a0, f32x4 &a1, const half8 &w0_hi, const half8 &w0_lo, const half8 &w1_hi,
                                              const half8 &w1_lo, const half8 &bm)
{
    if constexpr (FIRST) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a0) : "a"(w0_lo), "v"(bm));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a1) : "a"(w1_lo), "v"(bm));
    } else {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_lo), "v"(bm));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_lo), "v"(bm));
    }
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w0_hi), "v"(bm));
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w1_hi), "v"(bm));
}
// one tile's product with one K block of the mixed operand
// An operand of an asm MFMA must not have been written by a vector instruction in the instruction slot or two before it: the hardware
// does not interlock that (the MFMA reads what the register held before), and hipcc counts no wait states for what is inside asm.
// Operands normally arrive from LDS behind an s_waitcnt; the exception is a CONSTANT -- the zero state of step 0 -- which hipcc
// materialises with v_mov right in front of its first use (found when a rescheduled step 0 of gru_bar16q_kernel<64,64,true> computed
// its first r gate from whatever the registers held: tests/test_gpu_gru_bar16.py, saved gates).  settle() makes the value opaque (no
// rematerialisation later) and puts the wait states behind its definition, once.
__device__ __forceinline__ void settle(half8 &v) { asm volatile("s_nop 1" : "+v"(v)); }
__device__ __forceinline__ void mfma2(const half8 &w_hi, const half8 &w_lo, const half8 &bm, f32x4 &acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_lo, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_hi, bm, acc, 0, 0, 0);
}
// WS: wait states in front of the first read.  The compiler does not look into asm: a DPP source written by a VALU instruction just
// before needs two, an accumulator written by an MFMA eight counted from that MFMA (mfma_drain) -- callers that have pinned other
// instructions in between pass what is left of the eight.
template <int WS = 2>
__device__ __forceinline__ float pick_mix(const f32x4 &a)
{
    static_assert(WS >= 2 && WS <= 8, "wait states");
    float r;
    asm volatile("s_nop %c5\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x1\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x2\n\t"
                 "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x4\n\t"
                 "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x8"
                 : "=&v"(r)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(WS - 1));
    return r;
}
// ... into a register the caller keeps from step to step (see split2_kept).  Not volatile: the scheduler may move it among the MFMAs
// of another gate; `after` is an operand the statement only waits for -- the value whose producer must lie in front of it (the
// accumulator of the tile whose MFMAs supply the wait states, or the result of the pick that does).
template <int WS = 2>
__device__ __forceinline__ void pick_mix_kept(const f32x4 &a, float &r, float after)
{
    static_assert(WS >= 2 && WS <= 8, "wait states");
    asm("s_nop %c5\n\t"
        "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x1\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x2\n\t"
        "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x4\n\t"
        "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x8"
        : "+v"(r)
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(WS - 1), "v"(after));
}
// The same for gru_bar16d.hip's layout (eight chunks per workgroup: the two copies of a chunk sit FOUR columns apart, copy q&1 = 0 carries
// the hi half and owns rows 4g + {0, 1}, copy 1 the lo half and rows 4g + {2, 3}): value j of the lane = register 2(q&1) + j summed over
// both copies.  Two instructions: the even quartets take their partner four lanes up, the odd ones four lanes down.
__device__ __forceinline__ float pick_mix_d(const f32x4 &a, int j)
{
    float r;
    if (j == 0)
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(r)
                     : "v"(a[0]), "v"(a[2]));
    else
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                     : "=&v"(r)
                     : "v"(a[1]), "v"(a[3]));
    return r;
}
// barrier for a wave whose youngest LDS operation is a read of its own data (lds_bar_2reads with one operand image)
template <bool REAL = true>
__device__ __forceinline__ void lds_bar_1read()
{
    if constexpr (REAL) asm volatile("s_waitcnt lgkmcnt(1)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
}

// One projection tile for both sets, weights in accumulation registers: the two accumulation chains alternate (consecutive
// MFMAs never depend on each other) and hook(ic<i>) runs after MFMA i = 0 .. 6 KBLK - 1 -- the matrix pipe keeps the wave's issue
// port for 4 cycles of every 16, the leader's split of x is cut into pieces that fill the rest.
template <int KBLK_, class F>
__device__ __forceinline__ void tile2_mfma_acc(f32x4 &a0, f32x4 &a1, const half8 *w_hi, const half8 *w_lo, const half8 *x0_hi,
                                               const half8 *x0_lo, const half8 *x1_hi, const half8 *x1_lo, F &&hook)
{
    static_for<0, KBLK_>([&](auto KBC) {
        constexpr int kb = decltype(KBC)::value;
        if constexpr (kb == 0) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a0) : "a"(w_hi[0]), "v"(x0_lo[0]));
            hook(ic<0>{});
            // (x0_lo[0] is listed so that the fresh destination does not take over the registers the MFMA just issued still reads:
            //  tests/test_isa_hygiene.py)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(a1) : "a"(w_hi[0]), "v"(x1_lo[0]), "v"(x0_lo[0]));
            hook(ic<1>{});
        } else {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w_hi[kb]), "v"(x0_lo[kb]));
            hook(ic<6 * kb>{});
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w_hi[kb]), "v"(x1_lo[kb]));
            hook(ic<6 * kb + 1>{});
        }
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w_lo[kb]), "v"(x0_hi[kb]));
        hook(ic<6 * kb + 2>{});
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w_lo[kb]), "v"(x1_hi[kb]));
        hook(ic<6 * kb + 3>{});
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a0) : "a"(w_hi[kb]), "v"(x0_hi[kb]));
        hook(ic<6 * kb + 4>{});
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(a1) : "a"(w_hi[kb]), "v"(x1_hi[kb]));
        hook(ic<6 * kb + 5>{});
    });
}
