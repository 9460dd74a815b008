// test_rofilter.hip -- unit check of the read-order seed filter (seed_filter_ro, pg_kernels.hip) against a brute-force count per
// window position, on random windows and reads.  Diagnostics / test infrastructure, not product code.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ipindel_amd/csrc -mllvm -disable-machine-licm tests/rofilter_unit.hip -o pindel_amd/test_rofilter
// -DRO_TEST_VARIANT=1: no index changes at all (wrong masks; does the rest of the asm run at full occupancy?)
//                   =2: the index of bases 2 and 3 of a group by s_lshr + s_set_gpr_idx_idx instead of s_bfe_u32 m0
//                   =3: index mode switched on and off around every single v_alignbit
#if RO_TEST_VARIANT == 1
#define RO_IDX_ON(P) "s_nop 0\n\t"
#define RO_IDX_1(P) "s_nop 0\n\t"
#define RO_IDX_2(P) "s_nop 0\n\t"
#define RO_IDX_OFF "s_nop 0\n\t"
#elif RO_TEST_VARIANT == 2
#define RO_IDX_ON(P) "s_set_gpr_idx_on " P ", 0x3\n\t"
#define RO_IDX_1(P) "s_lshr_b32 s92, " P ", 8\n\ts_set_gpr_idx_idx s92\n\t"
#define RO_IDX_2(P) "s_lshr_b32 s92, " P ", 16\n\ts_set_gpr_idx_idx s92\n\t"
#define RO_IDX_OFF "s_set_gpr_idx_off\n\t"
#elif RO_TEST_VARIANT == 3
#define RO_IDX_ON(P) "s_set_gpr_idx_on " P ", 0x3\n\t"
#define RO_IDX_1(P) "s_set_gpr_idx_off\n\ts_lshr_b32 s92, " P ", 8\n\ts_set_gpr_idx_on s92, 0x3\n\t"
#define RO_IDX_2(P) "s_set_gpr_idx_off\n\ts_lshr_b32 s92, " P ", 16\n\ts_set_gpr_idx_on s92, 0x3\n\t"
#define RO_IDX_OFF "s_set_gpr_idx_off\n\t"
#elif RO_TEST_VARIANT == 10
#define RO_TEST_RP 1
#endif
#include "../pindel_amd/csrc/pg_kernels.hip"
#include <stdio.h>
#include <vector>

const PgEnvSwitches *pg_env_switches(void)
{
    static PgEnvSwitches e = {};
    return &e;
}

// same parameter list as pg_search_kernel: the filter fetches B.in from the kernarg segment at PgKArgs' offsets
// out[case][kind][lane][2]: masks of the asm, then of the brute force
__global__ __launch_bounds__(64, 7) void k_test(PgDevRef ref, PgDevParams prm, PgDevBatch B, uint32_t n_cases, uint32_t levels)
{
    __shared__ Lds<2, u32> lds;
    const int lane = threadIdx.x;
    u32 *out = (u32 *)B.out;
    const u32 *winsrc = (const u32 *)B.seq;           // per case 80 words x 3 planes
    for (uint32_t c = 0; c < n_cases; c++) {
        const PgInRec *rec = (const PgInRec *)B.planes + c;      // (a copy: a fault's address says whose access it was)
        for (int w = lane; w < 80; w += 64) lds.win[w] = make_uint4(winsrc[(c * 80 + w) * 3], winsrc[(c * 80 + w) * 3 + 1], winsrc[(c * 80 + w) * 3 + 2], 0u);
        __syncthreads();
        Search S;
        S.win = lds.win;
        S.ro = rec->ro;
        S.rid = c;
        S.rp_lo = (u32)(u64)(uintptr_t)(B.in + c);
        S.rp_hi = (u32)((u64)(uintptr_t)(B.in + c) >> 32);
        S.mm_tab = (const uint8_t *)B.in;
        S.T = (int)(rec->lvl >> 24);
        S.cap_state = (c % 3 == 2) ? (int)(c % 4) : 255;
        const int T = S.T;
        for (int kind = 0; kind < 3; kind++) {
            if (!((levels >> kind) & 1u)) continue;        // (levels: kinds to run | counter slices << 8)
            for (int o1 = 0; o1 < 2; o1++) {
                for (int wide = 0; wide < 2; wide++) {
                    u32 mF = 0u, mB = 0u;
#if RO_TEST_VARIANT != 4
                    if ((levels >> 8) == 4u) {             // four counter slices: up to 16 mismatch levels
                        if (kind == 0) seed_filter_ro<2, 0, 4>(S, o1 != 0, wide != 0, lane, mF, mB);
                        else if (kind == 1) seed_filter_ro<2, 1, 4>(S, o1 != 0, wide != 0, lane, mF, mB);
                        else seed_filter_ro<2, 2, 4>(S, o1 != 0, wide != 0, lane, mF, mB);
                    } else {
                        if (kind == 0) seed_filter_ro<2, 0, 3>(S, o1 != 0, wide != 0, lane, mF, mB);
                        else if (kind == 1) seed_filter_ro<2, 1, 3>(S, o1 != 0, wide != 0, lane, mF, mB);
                        else seed_filter_ro<2, 2, 3>(S, o1 != 0, wide != 0, lane, mF, mB);
                    }
#endif
                    // brute force
                    const u32 G = (S.ro >> (wide ? 4 : 0)) & 15u;
                    const int bound = (int)((S.ro >> (wide ? 16 : 8)) & 0xffu);
                    const int thrA = bound < S.cap_state ? bound : S.cap_state;
                    const int pre = kind == 2 ? 9 : 7;
                    const u32 *P = rec->prog[o1];
                    u32 wantF = 0u, wantB = 0u;
                    for (int bit = 0; bit < 32; bit++) {
                        const int p = 32 * (4 + lane) + bit;            // position index in the window's words (word 2 NB + lane)
                        for (int kb = 0; kb < 2; kb++) {
                            if (kind == 0 && kb == 1) continue;
                            if (kind == 1 && kb == 0) continue;
                            auto refsym = [&](int q) -> int {                  // 0..3 = ACGT, 4 = N
                                const uint4 e = lds.win[q >> 5];
                                const u32 b = (u32)q & 31u;
                                if ((e.z >> b) & 1u) return 4;
                                return (int)(((e.x >> b) & 1u) | (((e.y >> b) & 1u) << 1));
                            };
                            auto progsym = [&](int j) -> int {                  // symbol index of base j as the program holds it
                                if (j == 0) return (int)((P[0] >> 24) & 7u);
                                return (int)((P[(j - 1) / 3] >> (8 * ((j - 1) % 3))) & 7u);
                            };
                            // kind F reads the program's symbol; kind B reads its complement
                            auto readsym = [&](int j) -> int { const int x = progsym(j); return kb == 0 || x == 4 ? x : 3 - x; };
                            auto match = [&](int j) -> bool {
                                const int r = refsym(kb == 0 ? p + j : p - j), x = readsym(j);
                                return x == 4 ? r != 4 : r == x;
                            };
                            if (!match(0)) continue;
                            int cpre = 0, call = 0;
                            for (int j = 1; j <= 3 * (int)G; j++) {
                                const int mm = match(j) ? 0 : 1;
                                call += mm;
                                if (j <= pre) cpre += mm;
                            }
                            if (cpre <= thrA || call <= T - 1) (kb == 0 ? wantF : wantB) |= 1u << bit;
                        }
                    }
                    const size_t o = ((((size_t)c * 3 + kind) * 2 + o1) * 2 + wide) * 64 * 4 + lane * 4;
                    const u32 gF = kind == 1 ? 0u : mF, gB = kind == 1 ? mF : (kind == 2 ? mB : 0u);
                    if (blockIdx.x == 0) {
                        out[o] = gF;
                        out[o + 1] = gB;
                        out[o + 2] = wantF;
                        out[o + 3] = wantB;
                    }
                    // every workgroup of a full-occupancy launch checks itself: B.pool_used[kind] counts the lanes that differ
                    if (gF != wantF || gB != wantB) atomicAdd(B.pool_used + kind, 1u);
                }
            }
        }
        __syncthreads();
    }
}

static uint32_t rnd_state = 12345u;
static uint32_t rnd() { rnd_state = rnd_state * 1664525u + 1013904223u; return rnd_state >> 8; }

static void run(const unsigned kinds, const int slices)
{
    const int n_cases = 64;
    std::vector<PgInRec> recs(n_cases);
    std::vector<uint32_t> win((size_t)n_cases * 80 * 3);
    for (int c = 0; c < n_cases; c++) {
        // a low-entropy window (two letters dominate) so that many positions survive
        for (int w = 0; w < 80; w++) {
            uint32_t lo = 0, hi = 0, nn = 0;
            for (int b = 0; b < 32; b++) {
                const uint32_t r = rnd() % 100;
                const int sym = r < 45 ? 0 : (r < 90 ? 3 : (r < 94 ? 1 : (r < 98 ? 2 : 4)));
                if (sym == 4) nn |= 1u << b;
                else { lo |= (uint32_t)(sym & 1) << b; hi |= (uint32_t)(sym >> 1) << b; }
            }
            win[((size_t)c * 80 + w) * 3] = lo; win[((size_t)c * 80 + w) * 3 + 1] = hi; win[((size_t)c * 80 + w) * 3 + 2] = nn;
        }
        PgInRec &r = recs[c];
        memset(&r, 0, sizeof r);
        const uint32_t T = 3 + rnd() % (slices == 4 ? 14 : 6);      // 3 .. 8 (three counter slices) / 3 .. 16 (four)
        const uint32_t G0 = 4 + rnd() % 3, G1 = G0 + rnd() % 2;
        const uint32_t b0 = rnd() % T, b1 = rnd() % T;
        r.lvl = T << 24;
        r.ro = G0 | (G1 << 4) | (b0 << 8) | (b1 << 16) | PG_RO_OK;
        for (int o = 0; o < 2; o++)
            for (int g = 0; g < 8; g++) {
                uint32_t w = 0x30303030u;
                for (int k = 0; k < 3; k++) {
                    const uint32_t q = rnd() % 100;
                    const uint32_t sym = q < 45 ? 0 : (q < 90 ? 3 : (q < 93 ? 1 : (q < 96 ? 2 : 4)));
                    w |= sym << (8 * k);
                }
                if (g == 0) w |= (rnd() % 2 ? 0u : 3u) << 24;
                r.prog[o][g] = w;
            }
    }
    PgInRec *d_rec; uint32_t *d_win, *d_out;
    const size_t n_out = (size_t)n_cases * 3 * 2 * 2 * 64 * 4;
    hipMalloc(&d_rec, recs.size() * sizeof(PgInRec));
    hipMalloc(&d_win, win.size() * 4);
    hipMalloc(&d_out, n_out * 4);
    hipMemcpy(d_rec, recs.data(), recs.size() * sizeof(PgInRec), hipMemcpyHostToDevice);
    hipMemcpy(d_win, win.data(), win.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d_out, 0, n_out * 4);
    PgDevRef ref = {}; PgDevParams prm = {}; PgDevBatch B = {};
    PgInRec *d_rec2;
    hipMalloc(&d_rec2, recs.size() * sizeof(PgInRec));
    hipMemcpy(d_rec2, recs.data(), recs.size() * sizeof(PgInRec), hipMemcpyHostToDevice);
    B.planes = (const uint64_t *)d_rec2;
    B.in = d_rec; B.seq = (const uint8_t *)d_win; B.out = (PgOutRec *)d_out;
    uint32_t *d_cnt;
    hipMalloc(&d_cnt, 16);
    B.pool_used = d_cnt;
    for (int grid : { 1, 256, 256 * 4 * 7 }) {
        hipMemset(d_cnt, 0, 16);
        hipLaunchKernelGGL(k_test, dim3(grid), dim3(64), 0, 0, ref, prm, B, (uint32_t)n_cases, kinds | ((unsigned)slices << 8));
        hipError_t e = hipDeviceSynchronize();
        uint32_t cnt[4];
        hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost);
        printf("slices %d grid %5d: %s; lanes that differ from the brute force: F %u B %u DUAL %u\n", slices, grid, hipGetErrorString(e), cnt[0], cnt[1], cnt[2]);
    }
    std::vector<uint32_t> out(n_out);
    hipMemcpy(out.data(), d_out, n_out * 4, hipMemcpyDeviceToHost);
    const char *kn[3] = { "F", "B", "DUAL" };
    for (int kind = 0; kind < 3; kind++) {
        size_t bad = 0, tot = 0, surv = 0, shown = 0;
        for (int c = 0; c < n_cases; c++)
            for (int o1 = 0; o1 < 2; o1++)
                for (int wide = 0; wide < 2; wide++)
                    for (int lane = 0; lane < 64; lane++) {
                        const size_t o = ((((size_t)c * 3 + kind) * 2 + o1) * 2 + wide) * 64 * 4 + lane * 4;
                        tot += 2;
                        surv += __builtin_popcount(out[o + 2]) + __builtin_popcount(out[o + 3]);
                        if (out[o] != out[o + 2] || out[o + 1] != out[o + 3]) {
                            bad++;
                            if (shown++ < 4) printf("  %s case %d o1 %d wide %d lane %d: got %08x %08x want %08x %08x\n", kn[kind], c, o1, wide, lane, out[o], out[o + 1], out[o + 2], out[o + 3]);
                        }
                    }
        printf("slices %d kind %-4s: %zu lane results differ of %zu (survivors in the expectation: %zu)\n", slices, kn[kind], bad, tot / 2, surv);
    }
    hipFree(d_rec); hipFree(d_rec2); hipFree(d_win); hipFree(d_out); hipFree(d_cnt);
}

int main(int argc, char **argv)
{
    const unsigned kinds = argc > 1 ? (unsigned)atoi(argv[1]) : 7u;
    setvbuf(stdout, nullptr, _IONBF, 0);
    run(kinds, 3);      // three counter slices: up to 8 mismatch levels
    run(kinds, 4);      // four: up to 16
    return 0;
}
