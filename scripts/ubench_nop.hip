// Diagnostics: do s_nop / s_waitcnt / s_cbranch count as scalar-ALU instructions in SQ_INSTS_SALU, and what do they cost?
//   hipcc -O3 --offload-arch=gfx950 scripts/ubench_nop.hip -o /tmp/ubench_nop
//   rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH --output-format csv -d /tmp/nop -- /tmp/ubench_nop
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R10(x) x x x x x x x x x x
#define R100(x) R10(R10(x))
#define KERNEL(name, body)                                                         \
    __global__ __launch_bounds__(64) void name(unsigned *out)                      \
    {                                                                              \
        unsigned a = threadIdx.x;                                                  \
        for (int it = 0; it < 100; it++) asm volatile(R100(body) : "+v"(a) : : "scc"); \
        if (a == 0x12345u) out[0] = a;                                             \
    }
KERNEL(k_nop, "s_nop 0\n\t")
KERNEL(k_nop1, "s_nop 1\n\t")
KERNEL(k_waitcnt, "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_sadd, "s_add_u32 s100, s100, 1\n\t")
KERNEL(k_vxor, "v_xor_b32 %0, 1, %0\n\t")
KERNEL(k_branch, "s_cbranch_scc1 1f\n1:\n\t")
KERNEL(k_branch_taken, "s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\t")
int main()
{
    unsigned *out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct { const char *n; void (*k)(unsigned *); } ks[] = { { "s_nop 0", k_nop }, { "s_nop 1", k_nop1 }, { "s_waitcnt", k_waitcnt },
                                                            { "s_add", k_sadd }, { "v_xor", k_vxor }, { "s_cbranch (to next)", k_branch }, { "s_cmp + taken branch", k_branch_taken } };
    for (auto &c : ks) {
        for (int W : { 1, 4, 5 }) {
            hipLaunchKernelGGL(c.k, dim3(256 * 4 * W), dim3(64), 0, 0, out);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(c.k, dim3(256 * 4 * W), dim3(64), 0, 0, out);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-24s W=%d  %8.1f us  %6.2f ns per instruction and SIMD\n", c.n, W, ms * 1e3, ms * 1e6 / (10000.0 * W));
        }
    }
    return 0;
}
