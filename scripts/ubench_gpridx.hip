// ubench_gpridx.hip -- does gfx950 execute VGPR index mode (s_set_gpr_idx_on / _idx / _off), is an index written by the scalar unit
// seen by the very next vector instruction (no wait state), and what does "one index change + one v_alignbit" cost next to the
// shipped seed filter's "s_ff1 + s_bitset0 + v_alignbit with an SGPR shift"?  Diagnostics, not product code (round 6: the seed
// filter in read order picks the one-hot plane of a base's symbol by index instead of looping symbol by symbol).
//
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_gpridx.hip -o scripts/ubench_gpridx && scripts/ubench_gpridx
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define ITER 2000

// ---- correctness: v40..v44 = planes "lo" (lane * 16 + k), v48..v52 = planes "hi" (lane * 16 + k + 0x100); for a list of indices
// out[i] = alignbit(hi[idx_i], lo[idx_i], sh) with the index changed immediately before each vector instruction.
__global__ __launch_bounds__(64) void k_check(const unsigned *idx, unsigned *out, int n, int mode)
{
    const unsigned lane = threadIdx.x;
    unsigned res = 0u;
    for (int i = 0; i < n; i++) {
        unsigned ix = (unsigned)__builtin_amdgcn_readfirstlane((int)idx[i]);
        unsigned r;
        if (mode == 0)
            asm volatile("v_lshl_add_u32 v40, %1, 4, 0\n\tv_lshl_add_u32 v41, %1, 4, 1\n\tv_lshl_add_u32 v42, %1, 4, 2\n\t"
                         "v_lshl_add_u32 v43, %1, 4, 3\n\tv_lshl_add_u32 v44, %1, 4, 4\n\t"
                         "v_add_u32 v48, 0x100, v40\n\tv_add_u32 v49, 0x100, v41\n\tv_add_u32 v50, 0x100, v42\n\t"
                         "v_add_u32 v51, 0x100, v43\n\tv_add_u32 v52, 0x100, v44\n\t"
                         "s_set_gpr_idx_on %2, 0x3\n\t"
                         "v_alignbit_b32 %0, v48, v40, 4\n\t"
                         "s_set_gpr_idx_off"
                         : "=&v"(r) : "v"(lane), "s"(ix)
                         : "v40", "v41", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "m0");
        else if (mode == 1)   // the index changed between two vector instructions, mode left on
            asm volatile("v_lshl_add_u32 v40, %1, 4, 0\n\tv_lshl_add_u32 v41, %1, 4, 1\n\tv_lshl_add_u32 v42, %1, 4, 2\n\t"
                         "v_lshl_add_u32 v43, %1, 4, 3\n\tv_lshl_add_u32 v44, %1, 4, 4\n\t"
                         "v_add_u32 v48, 0x100, v40\n\tv_add_u32 v49, 0x100, v41\n\tv_add_u32 v50, 0x100, v42\n\t"
                         "v_add_u32 v51, 0x100, v43\n\tv_add_u32 v52, 0x100, v44\n\t"
                         "s_set_gpr_idx_on %3, 0x3\n\t"
                         "v_alignbit_b32 %0, v48, v40, 9\n\t"
                         "s_set_gpr_idx_idx %2\n\t"
                         "v_alignbit_b32 %0, v48, v40, 4\n\t"
                         "s_set_gpr_idx_off"
                         : "=&v"(r) : "v"(lane), "s"(ix), "s"(4u - ix)
                         : "v40", "v41", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "m0");
        else                  // the index written into M0 directly: s_bfe_u32 m0 of a 16-bit field that carries the enable bits
            asm volatile("v_lshl_add_u32 v40, %1, 4, 0\n\tv_lshl_add_u32 v41, %1, 4, 1\n\tv_lshl_add_u32 v42, %1, 4, 2\n\t"
                         "v_lshl_add_u32 v43, %1, 4, 3\n\tv_lshl_add_u32 v44, %1, 4, 4\n\t"
                         "v_add_u32 v48, 0x100, v40\n\tv_add_u32 v49, 0x100, v41\n\tv_add_u32 v50, 0x100, v42\n\t"
                         "v_add_u32 v51, 0x100, v43\n\tv_add_u32 v52, 0x100, v44\n\t"
                         "s_set_gpr_idx_on %3, 0x3\n\t"
                         "v_alignbit_b32 %0, v48, v40, 9\n\t"
                         "s_bfe_u32 m0, %2, 0x100010\n\t"
                         "v_alignbit_b32 %0, v48, v40, 4\n\t"
                         "s_set_gpr_idx_off"
                         : "=&v"(r) : "v"(lane), "s"((0x3000u | (mode == 3 ? ((unsigned)i & 15u) << 8 : 0u) | ix) << 16), "s"(4u - ix)
                         : "v40", "v41", "v42", "v43", "v44", "v48", "v49", "v50", "v51", "v52", "m0");   // (mode 3: M0[11:8] != 0 must be ignored)
        const unsigned lo = lane * 16u + ix, hi = lo + 0x100u;
        const unsigned want = (unsigned)((((unsigned long long)hi << 32) | lo) >> 4);
        res += r == want ? 0u : 1u;
    }
    out[lane] = res;
}

#define R8(a) a a a a a a a a
// ---- rates.  A: the shipped filter's per-base pattern (s_ff1 + s_bitset0 + v_alignbit with the SGPR shift).  B: index change + v_alignbit
// with an immediate shift.  C: s_bfe_u32 m0 + v_alignbit.  D: v_alignbit alone (SGPR shift).  Eight of each per asm body.
#define KERNEL(name, PRO, BODY)                                                                                            \
    __global__ __launch_bounds__(64) void name(unsigned *out, unsigned long long *cyc, unsigned seed)                  \
    {                                                                                                                  \
        unsigned a0 = threadIdx.x + seed, a1 = a0 * 3u;                                                                \
        unsigned s0 = seed | 0xffff0u, s1 = seed & 3u, s2 = 0x30013002u;                                               \
        unsigned long long t0 = __builtin_readcyclecounter();                                                          \
        for (int it = 0; it < ITER; it++) {                                                                            \
            asm volatile(PRO R8(BODY) "s_set_gpr_idx_off"                                    \
                         : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1), "+s"(s2)                                            \
                         :                                                                                             \
                         : "scc", "m0", "s40", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"); \
            s0 |= 0xffff0u;                                                                                            \
        }                                                                                                              \
        unsigned long long t1 = __builtin_readcyclecounter();                                                          \
        unsigned r = a0 ^ a1 ^ s0 ^ s1;                                                                                \
        if (r == 0x12345u) out[0] = r;                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) cyc[blockIdx.x] = t1 - t0;                                          \
    }
// (in A the index mode is on but the operands %0 / %1 are whatever the compiler picked: with index s1 in 0..3 the reads land on
//  neighbouring registers -- garbage values, same timing)
KERNEL(k_ship, "", "s_ff1_i32_b32 s40, %2\n\ts_bitset0_b32 %2, s40\n\tv_alignbit_b32 %0, v48, v40, s40\n\t")
KERNEL(k_idx, "s_set_gpr_idx_on %3, 0x3\n\t", "s_set_gpr_idx_idx %3\n\tv_alignbit_b32 %0, v48, v40, 5\n\t")
KERNEL(k_bfe, "s_set_gpr_idx_on %3, 0x3\n\t", "s_bfe_u32 m0, %4, 0x100010\n\tv_alignbit_b32 %0, v48, v40, 5\n\t")
KERNEL(k_bare, "", "v_alignbit_b32 %0, v48, v40, %3\n\t")
// three index changes, three alignbits, then the eight fast-rate instructions of a carry-save add (the real mix)
KERNEL(k_group, "s_set_gpr_idx_on %3, 0x3\n\t", "s_set_gpr_idx_idx %3\n\tv_alignbit_b32 v45, v48, v40, 5\n\ts_set_gpr_idx_idx %3\n\tv_alignbit_b32 v46, v48, v40, 6\n\t"
                "s_set_gpr_idx_idx %3\n\tv_alignbit_b32 v47, v48, v40, 7\n\ts_set_gpr_idx_off\n\t"
                "v_bitop3_b32 v53, v45, v46, v47 bitop3:0x69\n\tv_bitop3_b32 v54, v45, v46, v47 bitop3:0x17\n\t"
                "v_and_b32 v55, %0, v53\n\tv_xor_b32 %0, %0, v53\n\tv_bitop3_b32 v53, %1, v54, v55 bitop3:0xe8\n\tv_bitop3_b32 %1, %1, v54, v55 bitop3:0x96\n\t"
                "v_and_b32 v54, %0, v53\n\tv_xor_b32 %0, %0, v53\n\ts_set_gpr_idx_on %3, 0x3\n\t")
KERNEL(k_group_ship, "", "s_ff1_i32_b32 s40, %2\n\ts_bitset0_b32 %2, s40\n\tv_alignbit_b32 v45, v48, v40, s40\n\t"
                     "s_ff1_i32_b32 s40, %2\n\ts_bitset0_b32 %2, s40\n\tv_alignbit_b32 v46, v48, v40, s40\n\t"
                     "s_ff1_i32_b32 s40, %2\n\ts_bitset0_b32 %2, s40\n\tv_alignbit_b32 v47, v48, v40, s40\n\t"
                     "v_bitop3_b32 v53, v45, v46, v47 bitop3:0x69\n\tv_bitop3_b32 v54, v45, v46, v47 bitop3:0x17\n\t"
                     "v_and_b32 v55, %0, v53\n\tv_xor_b32 %0, %0, v53\n\tv_bitop3_b32 v53, %1, v54, v55 bitop3:0xe8\n\tv_bitop3_b32 %1, %1, v54, v55 bitop3:0x96\n\t"
                     "v_and_b32 v54, %0, v53\n\tv_xor_b32 %0, %0, v53\n\t")

typedef void (*kern_t)(unsigned *, unsigned long long *, unsigned);
struct Case { const char *name; kern_t k; int units; };

int main()
{
    unsigned *out, *idx;
    unsigned long long *cyc;
    hipMalloc(&out, 64 * 4);
    hipMalloc(&idx, 64 * 4);
    hipMalloc(&cyc, 4096 * 8);
    unsigned hidx[40];
    for (int i = 0; i < 40; i++) hidx[i] = (unsigned)((i * 7 + 3) % 5);
    hipMemcpy(idx, hidx, sizeof hidx, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; mode++) {
        hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, idx, out, 40, mode);
        unsigned h[64];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        unsigned bad = 0;
        for (int i = 0; i < 64; i++) bad += h[i];
        printf("check mode %d (%s): %u wrong of %d\n", mode,
               mode == 0 ? "s_set_gpr_idx_on, next instruction indexed" : (mode == 1 ? "s_set_gpr_idx_idx between two vector instructions" : (mode == 2 ? "s_bfe_u32 m0 between two vector instructions" : "the same with M0[11:8] != 0")),
               bad, 40 * 64);
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs\n", prop.name, cus);
    printf("%-14s %2s %10s %14s %14s\n", "kernel", "W", "wall_us", "wave_cyc/unit", "SIMD_cyc/unit");
    Case cases[] = { { "ship s+s+v", k_ship, 8 }, { "idx s+v", k_idx, 8 }, { "bfe-m0 s+v", k_bfe, 8 }, { "bare v", k_bare, 8 },
                     { "group idx", k_group, 8 }, { "group ship", k_group_ship, 8 } };
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (const Case &c : cases) {
        for (int W : { 1, 4, 7 }) {
            const int grid = cus * 4 * W;
            hipLaunchKernelGGL(c.k, dim3(grid), dim3(64), 0, 0, out, cyc, 1u);
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
            const double wave_cyc = (double)h[h.size() / 2];
            const double n_unit = (double)ITER * c.units;
            printf("%-14s %2d %10.1f %14.3f %14.3f\n", c.name, W, best * 1e3, wave_cyc / n_unit, wave_cyc / n_unit / W);
        }
    }
    return 0;
}
