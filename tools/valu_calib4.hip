// Round-6 additions (SDWA byte selects, the seed-offset idioms of sketch_wave_kernel) to the per-class VALU issue calibration; round 3's header follows.
// Round-3 additions to the per-class VALU issue calibration (tools/valu_calib2.hip): the classes the
// shipped window kernel is made of that round 2 had not measured -- v_and / v_or / v_sub / v_not / v_lshrrev, v_bitop3,
// v_lshl_add_u32, v_add_lshl_u32, v_max3_u32, v_min_u32 with a DPP operand, v_sub_co + v_addc (the borrow idiom),
// v_cmp + v_addc (the shift-accumulate idiom), v_cndmask on VCC behind a v_cmp (with the hazard nop the compiler inserts),
// v_bfrev, v_bcnt, v_lshl_add_u64 -- SIMD cycles per wave64 instruction at 8 waves per SIMD (span of all waves, s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_calib3.hip -o /tmp/valu_calib3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define ALL8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")
#define ALL32(M) ALL8(M) ALL8(M) ALL8(M) ALL8(M)

#define DEF(NAME, M, NI)                                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint64_t *stamps, int iters)                           \
    {                                                                                                                      \
        uint32_t r[8], s = threadIdx.x * 2654435761u + 12345u, t = threadIdx.x ^ 0x5bd1e995u;                              \
        for (int i = 0; i < 8; i++) r[i] = s * (i + 3) + t;                                                                \
        const uint64_t c0 = __builtin_amdgcn_s_memtime();                                                                  \
        const uint64_t w0 = __builtin_amdgcn_s_memrealtime();                                                              \
        for (int it = 0; it < iters; it++)                                                                                 \
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                         : "v"(s), "v"(t) : "vcc");                                                                       \
        const uint64_t c1 = __builtin_amdgcn_s_memtime();                                                                  \
        const uint64_t w1 = __builtin_amdgcn_s_memrealtime();                                                              \
        uint32_t acc = 0;                                                                                                  \
        for (int i = 0; i < 8; i++) acc ^= r[i];                                                                           \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                                         \
        if ((threadIdx.x & 63) == 0) {                                                                                     \
            const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6);                                                       \
            stamps[4 * wv] = c1 - c0; stamps[4 * wv + 1] = w1 - w0; stamps[4 * wv + 2] = w0; stamps[4 * wv + 3] = w1;      \
        }                                                                                                                  \
    }

#define M_xor(D) "v_xor_b32 " D ", %8, " D "\n"
#define M_bfe(D) "v_bfe_u32 " D ", " D ", 8, 8\n"
#define M_lshr_and(D) "v_lshrrev_b32 " D ", 8, " D "\n v_and_b32 " D ", 0x78, " D "\n"
#define M_and_sdwa(D) "v_and_b32_sdwa " D ", %8, " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define M_mov_sdwa(D) "v_mov_b32_sdwa " D ", " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n"
#define M_lshl_sdwa(D) "v_lshlrev_b32_sdwa " D ", %8, " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define M_add_sdwa(D) "v_add_u32_sdwa " D ", %8, " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n"
#define M_perm(D) "v_perm_b32 " D ", " D ", %8, %9\n"
#define M_bfi(D) "v_bfi_b32 " D ", %8, " D ", %9\n"
#define M_andor(D) "v_and_or_b32 " D ", " D ", %8, %9\n"

DEF(xor, M_xor, 32) DEF(bfe, M_bfe, 32) DEF(lshr_and, M_lshr_and, 64) DEF(and_sdwa, M_and_sdwa, 32) DEF(mov_sdwa, M_mov_sdwa, 32)
DEF(lshl_sdwa, M_lshl_sdwa, 32) DEF(add_sdwa, M_add_sdwa, 32) DEF(perm, M_perm, 32) DEF(bfi, M_bfi, 32) DEF(andor, M_andor, 32)

typedef void (*kern_t)(uint32_t *, uint64_t *, int);
struct Op { const char *name; kern_t k; int ni; };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; uint64_t *stamps;
    hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipMalloc(&stamps, (size_t)cus * 8 * 4 * 32);
    std::vector<uint64_t> hs((size_t)cus * 8 * 16);
    const Op ops[] = {{"v_xor_b32 (reference: 2.5)", k_xor, 32}, {"v_bfe_u32", k_bfe, 32}, {"v_lshrrev_b32 + v_and_b32 (per instr)", k_lshr_and, 64},
                      {"v_and_b32_sdwa src1_sel:BYTE_1", k_and_sdwa, 32}, {"v_mov_b32_sdwa src0_sel:BYTE_2", k_mov_sdwa, 32},
                      {"v_lshlrev_b32_sdwa src1_sel:BYTE_1", k_lshl_sdwa, 32}, {"v_add_u32_sdwa src1_sel:BYTE_3", k_add_sdwa, 32},
                      {"v_perm_b32", k_perm, 32}, {"v_bfi_b32", k_bfi, 32}, {"v_and_or_b32", k_andor, 32}};
    printf("# %s, %d CUs; iters=%d; SIMD cycles per wave64 VALU instruction from the span of all waves (s_memtime), clock from s_memrealtime\n", prop.gcnArchName, cus, iters);
    for (const Op &op : ops) {
        printf("%-60s", op.name);
        for (int W = 2; W <= 8; W *= 2) {
            const int blocks = cus * W;
            hipLaunchKernelGGL(op.k, dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(op.k, dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
            hipDeviceSynchronize();
            hipMemcpy(hs.data(), stamps, (size_t)blocks * 4 * 32, hipMemcpyDeviceToHost);
            std::vector<double> ghz;
            uint64_t t_lo = ~0ull, t_hi = 0;
            for (int wv = 0; wv < blocks * 4; wv++) {
                ghz.push_back((double)hs[4 * wv] / ((double)hs[4 * wv + 1] * 10.0));
                t_lo = std::min(t_lo, hs[4 * wv + 2]); t_hi = std::max(t_hi, hs[4 * wv + 3]);
            }
            std::sort(ghz.begin(), ghz.end());
            const double g = ghz[ghz.size() / 2];
            const double span_us = (double)(t_hi - t_lo) / 100.0;
            printf(" | W=%d %5.2f cyc (%.2f GHz)", W, span_us * 1e-6 * g * 1e9 / ((double)iters * op.ni * W), g);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
