// Per-instruction-class VALU issue calibration on gfx950: for every class the sketch
// kernel is made of, the SIMD cycles one wave64 instruction costs at 1, 2, 4 and 8 waves per SIMD, from
// in-kernel s_memtime stamps (shader clock) and s_memrealtime (100 MHz) -> also the clock held under load.
// Independent chains (8 destination registers per wave), explicit inline asm so the compiler cannot fuse.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_calib2.hip -o /tmp/valu_calib2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

enum { OP_XOR, OP_ADD, OP_ALIGNBIT, OP_BFI, OP_ADDCO_PAIR, OP_CMP_U64, OP_CMP_U32, OP_CNDMASK, OP_MIN_U32, OP_MIN3_U32,
       OP_FMA_F32, OP_XAD, OP_LSHL_OR, OP_AND_OR, OP_LSHL_B64, OP_MOV, OP_PERM, OP_MUL_LO, OP_MAD_U64, OP_PK_MIN_U16,
       OP_LSHLREV, OP_BFE, OP_OR3, OP_ADDC_VCC, OP_MBCNT, OP_DS_READ_B64, OP_DS_READ_B32, OP_DS_WRITE_B32,
       OP_MIX_SKETCH, OP_COUNT };
static const char *op_name[OP_COUNT] = {
    "v_xor_b32", "v_add_u32", "v_alignbit_b32", "v_bfi_b32", "v_add_co_u32+v_addc_co_u32 (per instr)", "v_cmp_lt_u64", "v_cmp_lt_u32",
    "v_cndmask_b32", "v_min_u32", "v_min3_u32", "v_fma_f32", "v_xad_u32", "v_lshl_or_b32", "v_and_or_b32", "v_lshlrev_b64", "v_mov_b32",
    "v_perm_b32", "v_mul_lo_u32", "v_mad_u64_u32", "v_pk_min_u16", "v_lshlrev_b32", "v_bfe_u32", "v_or3_b32",
    "v_addc_co_u32 (vcc chain)", "v_mbcnt_lo_u32_b32", "ds_read_b64", "ds_read_b32", "ds_write_b32",
    "mix: 2 alignbit + 2 bfi + 4 xor + add_co + addc + cmp_u64 + 2 cndmask (per instr)"};

#define ALL8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")
#define ALL32(M) ALL8(M) ALL8(M) ALL8(M) ALL8(M)

template <int OP>
__global__ __launch_bounds__(256) void calib_kernel(uint32_t *out, uint64_t *stamps, int iters)
{
    __shared__ uint32_t lds[2048];
    uint32_t r[8], s = threadIdx.x * 2654435761u + 12345u, t = threadIdx.x ^ 0x5bd1e995u;
    uint64_t q[8];
    float f[8];
    uint32_t o[4] = {s, t, s ^ t, s + t};
    float fs = (float)s, ft = (float)t;
    uint64_t s64 = ((uint64_t)s << 32) | t;
    for (int i = 0; i < 8; i++) { r[i] = s * (i + 3) + t; q[i] = ((uint64_t)r[i] << 32) | (r[i] * 7u); f[i] = (float)r[i]; }
    lds[threadIdx.x] = s; lds[threadIdx.x + 256] = t;
    uint32_t la = (threadIdx.x * 8u) & 0x1FF8u;
    __syncthreads();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    const uint64_t w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (OP == OP_XOR) {
#define M(D) "v_xor_b32 " D ", %8, " D "\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_ADD) {
#define M(D) "v_add_u32 " D ", %8, " D "\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_ALIGNBIT) {
#define M(D) "v_alignbit_b32 " D ", " D ", %8, 31\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_BFI) {
#define M(D) "v_bfi_b32 " D ", %8, " D ", %9\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_CMP_U32) {
#define M(D) "v_cmp_lt_u32 vcc, " D ", %8\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_CNDMASK) {
#define M(D) "v_cndmask_b32 " D ", " D ", %8, vcc\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_MIN_U32) {
#define M(D) "v_min_u32 " D ", %8, " D "\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_MIN3_U32) {
#define M(D) "v_min3_u32 " D ", %8, " D ", %9\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_XAD) {
#define M(D) "v_xad_u32 " D ", " D ", %8, %9\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_LSHL_OR) {
#define M(D) "v_lshl_or_b32 " D ", " D ", 1, %8\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_AND_OR) {
#define M(D) "v_and_or_b32 " D ", " D ", %8, %9\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_MOV) {
#define M(D) "v_mov_b32 " D ", %8\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_PERM) {
#define M(D) "v_perm_b32 " D ", " D ", %8, %9\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_MUL_LO) {
#define M(D) "v_mul_lo_u32 " D ", " D ", %8\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_PK_MIN_U16) {
#define M(D) "v_pk_min_u16 " D ", " D ", %8\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_LSHLREV) {
#define M(D) "v_lshlrev_b32 " D ", 1, " D "\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_BFE) {
#define M(D) "v_bfe_u32 " D ", " D ", 3, 7\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_OR3) {
#define M(D) "v_or3_b32 " D ", " D ", %8, %9\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_ADDC_VCC) {
#define M(D) "v_addc_co_u32 " D ", vcc, " D ", " D ", vcc\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_MBCNT) {
#define M(D) "v_mbcnt_lo_u32_b32 " D ", %8, " D "\n"
            asm volatile(ALL32(M) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_FMA_F32) {
#define M(D) "v_fma_f32 " D ", %8, " D ", %9\n"
            asm volatile(ALL32(M) : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : "v"(fs), "v"(ft));
#undef M
        }
        else if (OP == OP_CMP_U64) {
#define M(D) "v_cmp_lt_u64 vcc, " D ", %8\n"
            asm volatile(ALL32(M) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(s64), "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_LSHL_B64) {
#define M(D) "v_lshlrev_b64 " D ", 1, " D "\n"
            asm volatile(ALL32(M) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(s64), "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_MAD_U64) {
#define M(D) "v_mad_u64_u32 " D ", vcc, %9, %10, " D "\n"
            asm volatile(ALL32(M) : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(s64), "v"(s), "v"(t) : "vcc");
#undef M
        }
        else if (OP == OP_ADDCO_PAIR) {
#define P(L, H) "v_add_co_u32 " L ", vcc, %8, " L "\n v_addc_co_u32 " H ", vcc, %9, " H ", vcc\n"
#define P4 P("%0", "%4") P("%1", "%5") P("%2", "%6") P("%3", "%7")
            asm volatile(P4 P4 P4 P4 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s), "v"(t) : "vcc");
#undef P4
#undef P
        }
        else if (OP == OP_DS_READ_B64) {
#define M(D) "ds_read_b64 " D ", %8\n"
            asm volatile(ALL32(M) "s_waitcnt lgkmcnt(0)\n" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]) : "v"(la) : "memory");
#undef M
        }
        else if (OP == OP_DS_READ_B32) {
#define M(D) "ds_read_b32 " D ", %8\n"
            asm volatile(ALL32(M) "s_waitcnt lgkmcnt(0)\n" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(la) : "memory");
#undef M
        }
        else if (OP == OP_DS_WRITE_B32) {
#define M(D) "ds_write_b32 %8, " D "\n"
            asm volatile(ALL32(M) "s_waitcnt lgkmcnt(0)\n" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(la) : "memory");
#undef M
        }
        else if (OP == OP_MIX_SKETCH) {
            /* the per-k-mer instruction mix of the sketch kernel: srol + sror on 32-bit halves, table xors, h0 = fwd + rev,
               one 64-bit argmin combine (13 instructions), two independent chains, four times */
#define MX(A, B, C2, D2, O0, O1, Q0, Q1) \
            "v_alignbit_b32 " A ", " A ", " B ", 31\n v_bfi_b32 " B ", %12, " B ", %13\n v_xor_b32 " A ", %12, " A "\n v_xor_b32 " B ", %13, " B "\n" \
            "v_alignbit_b32 " C2 ", " C2 ", " D2 ", 1\n v_bfi_b32 " D2 ", %13, " D2 ", %12\n v_xor_b32 " C2 ", %12, " C2 "\n v_xor_b32 " D2 ", %13, " D2 "\n" \
            "v_add_co_u32 " O0 ", vcc, " A ", " C2 "\n v_addc_co_u32 " O1 ", vcc, " B ", " D2 ", vcc\n" \
            "v_cmp_lt_u64 vcc, " Q0 ", " Q1 "\n v_cndmask_b32 " O0 ", " O0 ", %12, vcc\n v_cndmask_b32 " O1 ", " O1 ", %13, vcc\n"
#define MX2 MX("%0", "%1", "%2", "%3", "%8", "%9", "%14", "%15") MX("%4", "%5", "%6", "%7", "%10", "%11", "%15", "%14")
            asm volatile(MX2 MX2 MX2 MX2 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                         "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]) : "v"(s), "v"(t), "v"(q[0]), "v"(q[1]) : "vcc");
#undef MX2
#undef MX
        }
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t w1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = o[0] ^ o[1] ^ o[2] ^ o[3];
    for (int i = 0; i < 8; i++) acc ^= r[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32) ^ __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ lds[(threadIdx.x * 7) & 2047];
    if ((threadIdx.x & 63) == 0) {
        const uint32_t wv = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[4 * wv] = c1 - c0;
        stamps[4 * wv + 1] = w1 - w0;
        stamps[4 * wv + 2] = w0;
        stamps[4 * wv + 3] = w1;
    }
}

static int instr_per_iter(int op)
{
    if (op == OP_MIX_SKETCH) return 4 * 26;
    return 32;
}

typedef void (*kern_t)(uint32_t *, uint64_t *, int);
template <int OP> static void fill(kern_t *tab) { tab[OP] = calib_kernel<OP>; if constexpr (OP + 1 < OP_COUNT) fill<OP + 1>(tab); }

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint32_t *out; uint64_t *stamps;
    const int maxblocks = cus * 8;
    hipMalloc(&out, (size_t)maxblocks * 256 * 4);
    hipMalloc(&stamps, (size_t)maxblocks * 4 * 32);
    std::vector<uint64_t> hs((size_t)maxblocks * 16);
    kern_t tab[OP_COUNT];
    fill<0>(tab);
    printf("# %s, %d CUs; iters=%d; per waves/SIMD W: SIMD cycles per wave64 instruction from the median wave (s_memtime / instr / W) and from the span of all waves, clock GHz (s_memtime / s_memrealtime), chip G wave-instr/s (hipEvent)\n",
           prop.gcnArchName, cus, iters);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int op = 0; op < OP_COUNT; op++) {
        printf("%-46s", op_name[op]);
        for (int W = 1; W <= 8; W *= 2) {
            const int blocks = cus * W;
            double best_cyc = 0, best_ghz = 0, best_rate = 0, best_span = 0, best_ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(a);
                hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                hipMemcpy(hs.data(), stamps, (size_t)blocks * 4 * 32, hipMemcpyDeviceToHost);
                std::vector<double> cyc, ghz;
                uint64_t t_lo = ~0ull, t_hi = 0;
                for (int wv = 0; wv < blocks * 4; wv++) {
                    cyc.push_back((double)hs[4 * wv]);
                    ghz.push_back((double)hs[4 * wv] / ((double)hs[4 * wv + 1] * 10.0)); /* 100 MHz ticks -> ns */
                    t_lo = std::min(t_lo, hs[4 * wv + 2]); t_hi = std::max(t_hi, hs[4 * wv + 3]);
                }
                best_span = (double)(t_hi - t_lo) / 100.0; /* us */
                best_ms = ms;
                std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
                const double n = (double)iters * instr_per_iter(op);
                best_cyc = cyc[cyc.size() / 2] / (n * W);
                best_ghz = ghz[ghz.size() / 2];
                best_rate = n * blocks * 4 / (ms * 1e-3) / 1e9;
            }
            /* cycles per instruction per SIMD from the wall span of all waves (first loop entry to last loop exit) */
            const double span_cyc = best_span * 1e-6 * best_ghz * 1e9 / ((double)iters * instr_per_iter(op) * W);
            printf(" | W=%d wave %5.2f span %5.2f cyc/instr/SIMD %4.2f GHz %6.1f G/s (span %.0f us, event %.0f us)", W, best_cyc, span_cyc, best_ghz,
                   best_rate, best_span, best_ms * 1e3);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
