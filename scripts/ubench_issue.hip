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
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case &c : cases) {
        for (int W : { 1, 2, 4, 5, 8 }) {
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
