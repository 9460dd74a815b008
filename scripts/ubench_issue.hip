// ubench_issue.hip -- issue-rate microbenchmarks for gfx950 (diagnostics, not product code).
//
// Question ("settle the bound"): how many cycles does a SIMD need per wave64 bit-op
// (v_xor / v_alignbit / v_bfi / v_and_or / DPP add / v_readlane / v_cmp) and how many scalar instructions
// can a CU issue per cycle?  Each kernel runs ITER x 64 copies of one instruction on 8 independent
// registers (no dependent stall), W waves per SIMD resident on every CU (grid = 256 CUs x 4 SIMDs x W
// single-wave workgroups).  Reported: SIMD cycles per instruction = wall time x clock / (instructions
// per wave x W), where the clock is measured in the same launch from s_memtime deltas (shader
// cycles, MI355X_MICROARCH.md) against wall time.
//
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_issue.hip -o scripts/ubench_issue && scripts/ubench_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define ITER 2000

#define R8(a) a a a a a a a a
#define BODY64(i0, i1, i2, i3, i4, i5, i6, i7) R8(i0 "\n\t" i1 "\n\t" i2 "\n\t" i3 "\n\t" i4 "\n\t" i5 "\n\t" i6 "\n\t" i7 "\n\t")

#define VKERNEL(name, I0, I1, I2, I3, I4, I5, I6, I7)                                                        \
    __global__ __launch_bounds__(64) void name(unsigned *out, unsigned long long *cyc, unsigned seed)       \
    {                                                                                                        \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u,          \
                 a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u, b = a0 ^ 0x5555u, c = 7u;                       \
        unsigned s0 = seed, s1 = seed * 3u, s2 = seed * 5u, s3 = seed * 7u, s4 = seed + 1u, s5 = seed + 2u,  \
                 s6 = seed + 3u, s7 = seed + 4u;                                                             \
        unsigned long long t0 = __builtin_readcyclecounter();                                                \
        for (int it = 0; it < ITER; it++) {                                                                  \
            asm volatile(BODY64(I0, I1, I2, I3, I4, I5, I6, I7)                                              \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),   \
                           "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)    \
                         : "v"(b), "v"(c)                                                                    \
                         : "vcc", "scc");                                                                    \
        }                                                                                                    \
        unsigned long long t1 = __builtin_readcyclecounter();                                                \
        unsigned r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;          \
        if (r == 0x12345u) out[0] = r;                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 4096) cyc[blockIdx.x] = t1 - t0;                                \
    }

// operands: %0..%7 = VGPR accumulators, %8..%15 = SGPR accumulators, %16 = b, %17 = c
VKERNEL(k_xor, "v_xor_b32 %0, %0, %16", "v_xor_b32 %1, %1, %16", "v_xor_b32 %2, %2, %16", "v_xor_b32 %3, %3, %16",
        "v_xor_b32 %4, %4, %16", "v_xor_b32 %5, %5, %16", "v_xor_b32 %6, %6, %16", "v_xor_b32 %7, %7, %16")
VKERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %16, %17", "v_alignbit_b32 %1, %1, %16, %17", "v_alignbit_b32 %2, %2, %16, %17",
        "v_alignbit_b32 %3, %3, %16, %17", "v_alignbit_b32 %4, %4, %16, %17", "v_alignbit_b32 %5, %5, %16, %17",
        "v_alignbit_b32 %6, %6, %16, %17", "v_alignbit_b32 %7, %7, %16, %17")
VKERNEL(k_bfi, "v_bfi_b32 %0, %16, %0, %17", "v_bfi_b32 %1, %16, %1, %17", "v_bfi_b32 %2, %16, %2, %17", "v_bfi_b32 %3, %16, %3, %17",
        "v_bfi_b32 %4, %16, %4, %17", "v_bfi_b32 %5, %16, %5, %17", "v_bfi_b32 %6, %16, %6, %17", "v_bfi_b32 %7, %16, %7, %17")
VKERNEL(k_and_or, "v_and_or_b32 %0, %0, %16, %17", "v_and_or_b32 %1, %1, %16, %17", "v_and_or_b32 %2, %2, %16, %17",
        "v_and_or_b32 %3, %3, %16, %17", "v_and_or_b32 %4, %4, %16, %17", "v_and_or_b32 %5, %5, %16, %17",
        "v_and_or_b32 %6, %6, %16, %17", "v_and_or_b32 %7, %7, %16, %17")
VKERNEL(k_add, "v_add_u32 %0, %0, %16", "v_add_u32 %1, %1, %16", "v_add_u32 %2, %2, %16", "v_add_u32 %3, %3, %16",
        "v_add_u32 %4, %4, %16", "v_add_u32 %5, %5, %16", "v_add_u32 %6, %6, %16", "v_add_u32 %7, %7, %16")
VKERNEL(k_fma, "v_fma_f32 %0, %0, %16, %17", "v_fma_f32 %1, %1, %16, %17", "v_fma_f32 %2, %2, %16, %17", "v_fma_f32 %3, %3, %16, %17",
        "v_fma_f32 %4, %4, %16, %17", "v_fma_f32 %5, %5, %16, %17", "v_fma_f32 %6, %6, %16, %17", "v_fma_f32 %7, %7, %16, %17")
VKERNEL(k_dpp_add, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf",
        "v_add_u32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %3, %3, %3 row_shr:8 row_mask:0xf bank_mask:0xf",
        "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %5, %5, %5 row_shr:2 row_mask:0xf bank_mask:0xf",
        "v_add_u32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf", "v_add_u32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf")
VKERNEL(k_readlane, "v_readlane_b32 %8, %0, 5", "v_readlane_b32 %9, %1, 6", "v_readlane_b32 %10, %2, 7", "v_readlane_b32 %11, %3, 8",
        "v_readlane_b32 %12, %4, 9", "v_readlane_b32 %13, %5, 10", "v_readlane_b32 %14, %6, 11", "v_readlane_b32 %15, %7, 12")
VKERNEL(k_cmp, "v_cmp_lt_u32 vcc, %0, %16", "v_cmp_lt_u32 vcc, %1, %16", "v_cmp_lt_u32 vcc, %2, %16", "v_cmp_lt_u32 vcc, %3, %16",
        "v_cmp_lt_u32 vcc, %4, %16", "v_cmp_lt_u32 vcc, %5, %16", "v_cmp_lt_u32 vcc, %6, %16", "v_cmp_lt_u32 vcc, %7, %16")
VKERNEL(k_popc, "v_bcnt_u32_b32 %0, %0, %16", "v_bcnt_u32_b32 %1, %1, %16", "v_bcnt_u32_b32 %2, %2, %16", "v_bcnt_u32_b32 %3, %3, %16",
        "v_bcnt_u32_b32 %4, %4, %16", "v_bcnt_u32_b32 %5, %5, %16", "v_bcnt_u32_b32 %6, %6, %16", "v_bcnt_u32_b32 %7, %7, %16")
// scalar unit
VKERNEL(k_salu, "s_add_u32 %8, %8, 3", "s_xor_b32 %9, %9, 5", "s_and_b32 %10, %10, 0xffff", "s_lshl_b32 %11, %11, 1",
        "s_add_u32 %12, %12, 3", "s_xor_b32 %13, %13, 5", "s_or_b32 %14, %14, 9", "s_lshr_b32 %15, %15, 1")
// half vector, half scalar, interleaved (can one wave's VALU and SALU overlap?  can different waves'?)
VKERNEL(k_mix, "v_xor_b32 %0, %0, %16", "s_add_u32 %8, %8, 3", "v_xor_b32 %1, %1, %16", "s_xor_b32 %9, %9, 5",
        "v_xor_b32 %2, %2, %16", "s_add_u32 %10, %10, 3", "v_xor_b32 %3, %3, %16", "s_xor_b32 %11, %11, 5")

// 3 vector : 1 scalar and 1 : 3 -- does the scalar stream ride for free next to the vector stream (separate
// issue ports) or do both share one issue slot per SIMD?
VKERNEL(k_mix31, "v_xor_b32 %0, %0, %16", "v_xor_b32 %1, %1, %16", "v_xor_b32 %2, %2, %16", "s_add_u32 %8, %8, 3",
        "v_xor_b32 %3, %3, %16", "v_xor_b32 %4, %4, %16", "v_xor_b32 %5, %5, %16", "s_xor_b32 %9, %9, 5")
VKERNEL(k_mix13, "s_add_u32 %8, %8, 3", "s_xor_b32 %9, %9, 5", "s_add_u32 %10, %10, 3", "v_xor_b32 %0, %0, %16",
        "s_add_u32 %11, %11, 3", "s_xor_b32 %12, %12, 5", "s_add_u32 %13, %13, 3", "v_xor_b32 %1, %1, %16")
// complex vector op (3.2 cycles alone) next to scalar ops
VKERNEL(k_mixc, "v_alignbit_b32 %0, %0, %16, %17", "s_add_u32 %8, %8, 3", "v_alignbit_b32 %1, %1, %16, %17", "s_xor_b32 %9, %9, 5",
        "v_alignbit_b32 %2, %2, %16, %17", "s_add_u32 %10, %10, 3", "v_alignbit_b32 %3, %3, %16, %17", "s_xor_b32 %11, %11, 5")
VKERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %16, %17 bitop3:0x96", "v_bitop3_b32 %1, %1, %16, %17 bitop3:0x96", "v_bitop3_b32 %2, %2, %16, %17 bitop3:0x96", "v_bitop3_b32 %3, %3, %16, %17 bitop3:0x96", "v_bitop3_b32 %4, %4, %16, %17 bitop3:0x96", "v_bitop3_b32 %5, %5, %16, %17 bitop3:0x96", "v_bitop3_b32 %6, %6, %16, %17 bitop3:0x96", "v_bitop3_b32 %7, %7, %16, %17 bitop3:0x96")
VKERNEL(k_and, "v_and_b32 %0, %0, %16", "v_and_b32 %1, %1, %16", "v_and_b32 %2, %2, %16", "v_and_b32 %3, %3, %16", "v_and_b32 %4, %4, %16", "v_and_b32 %5, %5, %16", "v_and_b32 %6, %6, %16", "v_and_b32 %7, %7, %16")
VKERNEL(k_xnor, "v_xnor_b32 %0, %0, %16", "v_xnor_b32 %1, %1, %16", "v_xnor_b32 %2, %2, %16", "v_xnor_b32 %3, %3, %16", "v_xnor_b32 %4, %4, %16", "v_xnor_b32 %5, %5, %16", "v_xnor_b32 %6, %6, %16", "v_xnor_b32 %7, %7, %16")
VKERNEL(k_not, "v_not_b32 %0, %0", "v_not_b32 %1, %1", "v_not_b32 %2, %2", "v_not_b32 %3, %3", "v_not_b32 %4, %4", "v_not_b32 %5, %5", "v_not_b32 %6, %6", "v_not_b32 %7, %7")
VKERNEL(k_mov, "v_mov_b32 %0, %16", "v_mov_b32 %1, %16", "v_mov_b32 %2, %16", "v_mov_b32 %3, %16", "v_mov_b32 %4, %16", "v_mov_b32 %5, %16", "v_mov_b32 %6, %16", "v_mov_b32 %7, %16")
VKERNEL(k_lshl, "v_lshlrev_b32 %0, 1, %0", "v_lshlrev_b32 %1, 1, %1", "v_lshlrev_b32 %2, 1, %2", "v_lshlrev_b32 %3, 1, %3", "v_lshlrev_b32 %4, 1, %4", "v_lshlrev_b32 %5, 1, %5", "v_lshlrev_b32 %6, 1, %6", "v_lshlrev_b32 %7, 1, %7")
VKERNEL(k_or3, "v_or3_b32 %0, %0, %16, %17", "v_or3_b32 %1, %1, %16, %17", "v_or3_b32 %2, %2, %16, %17", "v_or3_b32 %3, %3, %16, %17", "v_or3_b32 %4, %4, %16, %17", "v_or3_b32 %5, %5, %16, %17", "v_or3_b32 %6, %6, %16, %17", "v_or3_b32 %7, %7, %16, %17")
VKERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %16, vcc", "v_cndmask_b32 %1, %1, %16, vcc", "v_cndmask_b32 %2, %2, %16, vcc", "v_cndmask_b32 %3, %3, %16, vcc", "v_cndmask_b32 %4, %4, %16, vcc", "v_cndmask_b32 %5, %5, %16, vcc", "v_cndmask_b32 %6, %6, %16, vcc", "v_cndmask_b32 %7, %7, %16, vcc")
VKERNEL(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %16, s[10:11]", "v_cndmask_b32_e64 %1, %1, %16, s[10:11]", "v_cndmask_b32_e64 %2, %2, %16, s[10:11]", "v_cndmask_b32_e64 %3, %3, %16, s[10:11]", "v_cndmask_b32_e64 %4, %4, %16, s[10:11]", "v_cndmask_b32_e64 %5, %5, %16, s[10:11]", "v_cndmask_b32_e64 %6, %6, %16, s[10:11]", "v_cndmask_b32_e64 %7, %7, %16, s[10:11]")
VKERNEL(k_xor_s, "v_xor_b32 %0, %8, %0", "v_xor_b32 %1, %9, %1", "v_xor_b32 %2, %10, %2", "v_xor_b32 %3, %11, %3", "v_xor_b32 %4, %12, %4", "v_xor_b32 %5, %13, %5", "v_xor_b32 %6, %14, %6", "v_xor_b32 %7, %15, %7")
VKERNEL(k_alignbit_s, "v_alignbit_b32 %0, %0, %16, %8", "v_alignbit_b32 %1, %1, %16, %9", "v_alignbit_b32 %2, %2, %16, %10", "v_alignbit_b32 %3, %3, %16, %11", "v_alignbit_b32 %4, %4, %16, %12", "v_alignbit_b32 %5, %5, %16, %13", "v_alignbit_b32 %6, %6, %16, %14", "v_alignbit_b32 %7, %7, %16, %15")
VKERNEL(k_min, "v_min_u32 %0, %0, %16", "v_min_u32 %1, %1, %16", "v_min_u32 %2, %2, %16", "v_min_u32 %3, %3, %16", "v_min_u32 %4, %4, %16", "v_min_u32 %5, %5, %16", "v_min_u32 %6, %6, %16", "v_min_u32 %7, %7, %16")
VKERNEL(k_sub, "v_sub_u32 %0, %0, %16", "v_sub_u32 %1, %1, %16", "v_sub_u32 %2, %2, %16", "v_sub_u32 %3, %3, %16", "v_sub_u32 %4, %4, %16", "v_sub_u32 %5, %5, %16", "v_sub_u32 %6, %6, %16", "v_sub_u32 %7, %7, %16")
VKERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %16", "v_lshl_or_b32 %1, %1, 1, %16", "v_lshl_or_b32 %2, %2, 1, %16", "v_lshl_or_b32 %3, %3, 1, %16", "v_lshl_or_b32 %4, %4, 1, %16", "v_lshl_or_b32 %5, %5, 1, %16", "v_lshl_or_b32 %6, %6, 1, %16", "v_lshl_or_b32 %7, %7, 1, %16")
VKERNEL(k_add3, "v_add3_u32 %0, %0, %16, %17", "v_add3_u32 %1, %1, %16, %17", "v_add3_u32 %2, %2, %16, %17", "v_add3_u32 %3, %3, %16, %17", "v_add3_u32 %4, %4, %16, %17", "v_add3_u32 %5, %5, %16, %17", "v_add3_u32 %6, %6, %16, %17", "v_add3_u32 %7, %7, %16, %17")
VKERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %0, %16, %0", "v_mbcnt_lo_u32_b32 %1, %16, %1", "v_mbcnt_lo_u32_b32 %2, %16, %2", "v_mbcnt_lo_u32_b32 %3, %16, %3", "v_mbcnt_lo_u32_b32 %4, %16, %4", "v_mbcnt_lo_u32_b32 %5, %16, %5", "v_mbcnt_lo_u32_b32 %6, %16, %6", "v_mbcnt_lo_u32_b32 %7, %16, %7")
VKERNEL(k_perm, "v_perm_b32 %0, %0, %16, %17", "v_perm_b32 %1, %1, %16, %17", "v_perm_b32 %2, %2, %16, %17", "v_perm_b32 %3, %3, %16, %17", "v_perm_b32 %4, %4, %16, %17", "v_perm_b32 %5, %5, %16, %17", "v_perm_b32 %6, %6, %16, %17", "v_perm_b32 %7, %7, %16, %17")
VKERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 7", "v_bfe_u32 %1, %1, 3, 7", "v_bfe_u32 %2, %2, 3, 7", "v_bfe_u32 %3, %3, 3, 7", "v_bfe_u32 %4, %4, 3, 7", "v_bfe_u32 %5, %5, 3, 7", "v_bfe_u32 %6, %6, 3, 7", "v_bfe_u32 %7, %7, 3, 7")
VKERNEL(k_and_i, "v_and_b32 %0, 15, %0", "v_and_b32 %1, 15, %1", "v_and_b32 %2, 15, %2", "v_and_b32 %3, 15, %3", "v_and_b32 %4, 15, %4", "v_and_b32 %5, 15, %5", "v_and_b32 %6, 15, %6", "v_and_b32 %7, 15, %7")
VKERNEL(k_and_l, "v_and_b32 %0, 0x12345, %0", "v_and_b32 %1, 0x12345, %1", "v_and_b32 %2, 0x12345, %2", "v_and_b32 %3, 0x12345, %3", "v_and_b32 %4, 0x12345, %4", "v_and_b32 %5, 0x12345, %5", "v_and_b32 %6, 0x12345, %6", "v_and_b32 %7, 0x12345, %7")
VKERNEL(k_and_s, "v_and_b32 %0, %8, %0", "v_and_b32 %1, %9, %1", "v_and_b32 %2, %10, %2", "v_and_b32 %3, %11, %3", "v_and_b32 %4, %12, %4", "v_and_b32 %5, %13, %5", "v_and_b32 %6, %14, %6", "v_and_b32 %7, %15, %7")
VKERNEL(k_or, "v_or_b32 %0, %0, %16", "v_or_b32 %1, %1, %16", "v_or_b32 %2, %2, %16", "v_or_b32 %3, %3, %16", "v_or_b32 %4, %4, %16", "v_or_b32 %5, %5, %16", "v_or_b32 %6, %6, %16", "v_or_b32 %7, %7, %16")
VKERNEL(k_bitop3_s, "v_bitop3_b32 %0, %0, %8, %17 bitop3:0x96", "v_bitop3_b32 %1, %1, %9, %17 bitop3:0x96", "v_bitop3_b32 %2, %2, %10, %17 bitop3:0x96", "v_bitop3_b32 %3, %3, %11, %17 bitop3:0x96", "v_bitop3_b32 %4, %4, %12, %17 bitop3:0x96", "v_bitop3_b32 %5, %5, %13, %17 bitop3:0x96", "v_bitop3_b32 %6, %6, %14, %17 bitop3:0x96", "v_bitop3_b32 %7, %7, %15, %17 bitop3:0x96")
VKERNEL(k_bitop3_i, "v_bitop3_b32 %0, %0, 0, %17 bitop3:0x96", "v_bitop3_b32 %1, %1, 0, %17 bitop3:0x96", "v_bitop3_b32 %2, %2, 0, %17 bitop3:0x96", "v_bitop3_b32 %3, %3, 0, %17 bitop3:0x96", "v_bitop3_b32 %4, %4, 0, %17 bitop3:0x96", "v_bitop3_b32 %5, %5, 0, %17 bitop3:0x96", "v_bitop3_b32 %6, %6, 0, %17 bitop3:0x96", "v_bitop3_b32 %7, %7, 0, %17 bitop3:0x96")
VKERNEL(k_mov_s, "v_mov_b32 %0, %8", "v_mov_b32 %1, %9", "v_mov_b32 %2, %10", "v_mov_b32 %3, %11", "v_mov_b32 %4, %12", "v_mov_b32 %5, %13", "v_mov_b32 %6, %14", "v_mov_b32 %7, %15")
VKERNEL(k_mov_i, "v_mov_b32 %0, 0", "v_mov_b32 %1, 0", "v_mov_b32 %2, 0", "v_mov_b32 %3, 0", "v_mov_b32 %4, 0", "v_mov_b32 %5, 0", "v_mov_b32 %6, 0", "v_mov_b32 %7, 0")
VKERNEL(k_lshl_v, "v_lshlrev_b32 %0, %17, %0", "v_lshlrev_b32 %1, %17, %1", "v_lshlrev_b32 %2, %17, %2", "v_lshlrev_b32 %3, %17, %3", "v_lshlrev_b32 %4, %17, %4", "v_lshlrev_b32 %5, %17, %5", "v_lshlrev_b32 %6, %17, %6", "v_lshlrev_b32 %7, %17, %7")
VKERNEL(k_lshr_i, "v_lshrrev_b32 %0, 1, %0", "v_lshrrev_b32 %1, 1, %1", "v_lshrrev_b32 %2, 1, %2", "v_lshrrev_b32 %3, 1, %3", "v_lshrrev_b32 %4, 1, %4", "v_lshrrev_b32 %5, 1, %5", "v_lshrrev_b32 %6, 1, %6", "v_lshrrev_b32 %7, 1, %7")
VKERNEL(k_max, "v_max_u32 %0, %0, %16", "v_max_u32 %1, %1, %16", "v_max_u32 %2, %2, %16", "v_max_u32 %3, %3, %16", "v_max_u32 %4, %4, %16", "v_max_u32 %5, %5, %16", "v_max_u32 %6, %6, %16", "v_max_u32 %7, %7, %16")
VKERNEL(k_xor64, "v_xor_b32_e64 %0, %0, %16", "v_xor_b32_e64 %1, %1, %16", "v_xor_b32_e64 %2, %2, %16", "v_xor_b32_e64 %3, %3, %16", "v_xor_b32_e64 %4, %4, %16", "v_xor_b32_e64 %5, %5, %16", "v_xor_b32_e64 %6, %6, %16", "v_xor_b32_e64 %7, %7, %16")
VKERNEL(k_cmp64, "v_cmp_lt_u32_e64 s[10:11], %0, %16", "v_cmp_lt_u32_e64 s[10:11], %1, %16", "v_cmp_lt_u32_e64 s[10:11], %2, %16", "v_cmp_lt_u32_e64 s[10:11], %3, %16", "v_cmp_lt_u32_e64 s[10:11], %4, %16", "v_cmp_lt_u32_e64 s[10:11], %5, %16", "v_cmp_lt_u32_e64 s[10:11], %6, %16", "v_cmp_lt_u32_e64 s[10:11], %7, %16")
VKERNEL(k_add_s, "v_add_u32 %0, %8, %0", "v_add_u32 %1, %9, %1", "v_add_u32 %2, %10, %2", "v_add_u32 %3, %11, %3", "v_add_u32 %4, %12, %4", "v_add_u32 %5, %13, %5", "v_add_u32 %6, %14, %6", "v_add_u32 %7, %15, %7")
VKERNEL(k_add_i, "v_add_u32 %0, 1, %0", "v_add_u32 %1, 1, %1", "v_add_u32 %2, 1, %2", "v_add_u32 %3, 1, %3", "v_add_u32 %4, 1, %4", "v_add_u32 %5, 1, %5", "v_add_u32 %6, 1, %6", "v_add_u32 %7, 1, %7")
VKERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %16", "v_lshl_add_u32 %1, %1, 1, %16", "v_lshl_add_u32 %2, %2, 1, %16", "v_lshl_add_u32 %3, %3, 1, %16", "v_lshl_add_u32 %4, %4, 1, %16", "v_lshl_add_u32 %5, %5, 1, %16", "v_lshl_add_u32 %6, %6, 1, %16", "v_lshl_add_u32 %7, %7, 1, %16")
VKERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %16", "v_add_co_u32 %1, vcc, %1, %16", "v_add_co_u32 %2, vcc, %2, %16", "v_add_co_u32 %3, vcc, %3, %16", "v_add_co_u32 %4, vcc, %4, %16", "v_add_co_u32 %5, vcc, %5, %16", "v_add_co_u32 %6, vcc, %6, %16", "v_add_co_u32 %7, vcc, %7, %16")
VKERNEL(k_alignbit_i, "v_alignbit_b32 %0, %0, %16, 5", "v_alignbit_b32 %1, %1, %16, 5", "v_alignbit_b32 %2, %2, %16, 5", "v_alignbit_b32 %3, %3, %16, 5", "v_alignbit_b32 %4, %4, %16, 5", "v_alignbit_b32 %5, %5, %16, 5", "v_alignbit_b32 %6, %6, %16, 5", "v_alignbit_b32 %7, %7, %16, 5")
VKERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %16", "v_mul_u32_u24 %1, %1, %16", "v_mul_u32_u24 %2, %2, %16", "v_mul_u32_u24 %3, %3, %16", "v_mul_u32_u24 %4, %4, %16", "v_mul_u32_u24 %5, %5, %16", "v_mul_u32_u24 %6, %6, %16", "v_mul_u32_u24 %7, %7, %16")
VKERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf")
VKERNEL(k_writelane, "v_writelane_b32 %0, %8, 3", "v_writelane_b32 %1, %9, 3", "v_writelane_b32 %2, %10, 3", "v_writelane_b32 %3, %11, 3", "v_writelane_b32 %4, %12, 3", "v_writelane_b32 %5, %13, 3", "v_writelane_b32 %6, %14, 3", "v_writelane_b32 %7, %15, 3")
VKERNEL(k_readfirst, "v_readfirstlane_b32 %8, %0", "v_readfirstlane_b32 %9, %1", "v_readfirstlane_b32 %10, %2", "v_readfirstlane_b32 %11, %3", "v_readfirstlane_b32 %12, %4", "v_readfirstlane_b32 %13, %5", "v_readfirstlane_b32 %14, %6", "v_readfirstlane_b32 %15, %7")
VKERNEL(k_xor_m1, "v_xor_b32 %0, -1, %0", "v_xor_b32 %1, -1, %1", "v_xor_b32 %2, -1, %2", "v_xor_b32 %3, -1, %3", "v_xor_b32 %4, -1, %4", "v_xor_b32 %5, -1, %5", "v_xor_b32 %6, -1, %6", "v_xor_b32 %7, -1, %7")
VKERNEL(k_pk, "v_pk_add_u16 %0, %0, %16", "v_pk_add_u16 %1, %1, %16", "v_pk_add_u16 %2, %2, %16", "v_pk_add_u16 %3, %3, %16", "v_pk_add_u16 %4, %4, %16", "v_pk_add_u16 %5, %5, %16", "v_pk_add_u16 %6, %6, %16", "v_pk_add_u16 %7, %7, %16")
VKERNEL(k_and2, "v_and_b32 %0, %16, %17", "v_and_b32 %1, %16, %17", "v_and_b32 %2, %16, %17", "v_and_b32 %3, %16, %17", "v_and_b32 %4, %16, %17", "v_and_b32 %5, %16, %17", "v_and_b32 %6, %16, %17", "v_and_b32 %7, %16, %17")
VKERNEL(k_cmpsel, "v_cmp_lt_u32 vcc, %0, %16", "v_cndmask_b32 %1, %1, %16, vcc", "v_cmp_lt_u32 vcc, %2, %16", "v_cndmask_b32 %3, %3, %16, vcc", "v_cmp_lt_u32 vcc, %4, %16", "v_cndmask_b32 %5, %5, %16, vcc", "v_cmp_lt_u32 vcc, %6, %16", "v_cndmask_b32 %7, %7, %16, vcc")
// LDS broadcast read + vector ops
typedef void (*kern_t)(unsigned *, unsigned long long *, unsigned);
struct Case { const char *name; kern_t k; int insts_per_iter; const char *what; };

int main()
{
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 4096 * 8);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    printf("# ITER %d x 64 instructions per wave; grid = CUs x 4 SIMDs x W single-wave workgroups\n", ITER);
    printf("%-12s %2s %10s %10s %12s %14s %14s\n", "kernel", "W", "wall_us", "MHz_eff", "wave_cyc/inst", "SIMD_cyc/inst", "CU_inst/cyc");
    Case cases[] = {
        { "v_xor", k_xor, 64, "" }, { "v_alignbit", k_alignbit, 64, "" }, { "v_bfi", k_bfi, 64, "" },
        { "v_and_or", k_and_or, 64, "" }, { "v_add_u32", k_add, 64, "" }, { "v_fma_f32", k_fma, 64, "" },
        { "v_add_dpp", k_dpp_add, 64, "" }, { "v_readlane", k_readlane, 64, "" }, { "v_cmp", k_cmp, 64, "" },
        { "v_bcnt", k_popc, 64, "" }, { "s_alu", k_salu, 64, "" }, { "v+s mix", k_mix, 64, "" }, { "3v+1s mix", k_mix31, 64, "" }, { "1v+3s mix", k_mix13, 64, "" },
        { "valign+s", k_mixc, 64, "" },
        { "v_and imm", k_and_i, 64, "" }, { "v_and lit", k_and_l, 64, "" }, { "v_and sgpr", k_and_s, 64, "" }, { "v_or", k_or, 64, "" }, { "bitop3 sgpr", k_bitop3_s, 64, "" }, { "bitop3 imm", k_bitop3_i, 64, "" }, { "v_mov sgpr", k_mov_s, 64, "" }, { "v_mov imm", k_mov_i, 64, "" }, { "v_lshl vgpr", k_lshl_v, 64, "" }, { "v_lshr imm", k_lshr_i, 64, "" }, { "v_max_u32", k_max, 64, "" }, { "v_xor e64", k_xor64, 64, "" }, { "v_cmp e64", k_cmp64, 64, "" }, { "v_add sgpr", k_add_s, 64, "" }, { "v_add imm", k_add_i, 64, "" }, { "v_lshl_add", k_lshl_add, 64, "" }, { "v_add_co", k_add_co, 64, "" }, { "valign imm", k_alignbit_i, 64, "" }, { "v_mul_u24", k_mul24, 64, "" }, { "v_mov dpp", k_mov_dpp, 64, "" }, { "v_writelane", k_writelane, 64, "" }, { "v_readfirst", k_readfirst, 64, "" }, { "v_xor neg1", k_xor_m1, 64, "" }, { "v_pk_add_u16", k_pk, 64, "" }, { "v_and_b32 x2", k_and2, 64, "" }, { "cmp+cndmask", k_cmpsel, 64, "" },
        { "v_bitop3", k_bitop3, 64, "" }, { "v_and", k_and, 64, "" }, { "v_xnor", k_xnor, 64, "" }, { "v_not", k_not, 64, "" }, { "v_mov", k_mov, 64, "" }, { "v_lshlrev", k_lshl, 64, "" }, { "v_or3", k_or3, 64, "" }, { "v_cndmask", k_cndmask, 64, "" }, { "v_cndmask64", k_cndmask64, 64, "" }, { "v_xor sgpr", k_xor_s, 64, "" }, { "valign sgpr", k_alignbit_s, 64, "" }, { "v_min_u32", k_min, 64, "" }, { "v_sub_u32", k_sub, 64, "" }, { "v_lshl_or", k_lshl_or, 64, "" }, { "v_add3", k_add3, 64, "" }, { "v_mbcnt", k_mbcnt, 64, "" }, { "v_perm", k_perm, 64, "" }, { "v_bfe", k_bfe, 64, "" },
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case &c : cases) {
        for (int W : { 1, 2, 4, 5 }) {
            const int grid = cus * 4 * W;
            hipLaunchKernelGGL(c.k, dim3(grid), dim3(64), 0, 0, out, cyc, 1u);   // warm-up
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(c.k, dim3(grid), dim3(64), 0, 0, out, cyc, 1u);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms);
            }
            std::vector<unsigned long long> h(std::min(grid, 4096));
            hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double wave_cyc = (double)h[h.size() / 2];             // median s_memtime delta of a wave
            const double n_inst = (double)ITER * c.insts_per_iter;
            // all waves run concurrently (one per SIMD slot): kernel wall ~= wave time; the clock follows
            const double mhz = wave_cyc / (best * 1e3);
            const double wave_cpi = wave_cyc / n_inst;
            printf("%-12s %2d %10.1f %10.0f %12.3f %14.3f %14.3f\n", c.name, W, best * 1e3, mhz, wave_cpi,
                   wave_cpi / W, 4.0 * W / wave_cpi);
        }
    }
    return 0;
}
