/* gfx950 instruction wrappers used by the kernels (the SIMT mock under tests/sim supplies its
 * own file of the same name with portable bodies). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

/* ({hi,lo} >> sh)[31:0], sh in 0..31 : v_alignbit_b32 */
__device__ __forceinline__ uint32_t ntl_alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

/* (x >> off) & ((1 << width) - 1) : v_bfe_u32 */
__device__ __forceinline__ uint32_t ntl_bfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }

/* (a & mask) | (b & ~mask) : v_bfi_b32 (written as asm: the compiler re-associates the C form into more instructions) */
template <uint32_t MASK>
__device__ __forceinline__ uint32_t ntl_bfi(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "n"(MASK), "v"(a), "v"(b));
    return r;
}

/* (acc << 1) | (a != b): v_cmp_ne_u32 + v_addc_co_u32 (acc + acc + carry) */
__device__ __forceinline__ uint32_t ntl_shl1_or_ne(uint32_t acc, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cmp_ne_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %3, %3, vcc" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
    return r;
}

/* (acc << 1) | (a <= b) */
__device__ __forceinline__ uint32_t ntl_shl1_or_le(uint32_t acc, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cmp_le_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %3, %3, vcc" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
    return r;
}

/* (acc << 1) | (a < b), and d = a - b (wrapping): v_sub_co_u32 (the borrow is the comparison) + v_addc_co_u32 */
__device__ __forceinline__ uint32_t ntl_shl1_or_lt_diff(uint32_t acc, uint32_t a, uint32_t b, uint32_t &d)
{
    uint32_t r, dd;
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %4, %4, vcc" : "=v"(r), "=&v"(dd) : "v"(a), "v"(b), "v"(acc) : "vcc");
    d = dd;
    return r;
}

/* minimum over the 16 lanes of a row (lanes 16i..16i+15), result in every lane: four v_min_u32 with DPP row rotations */
__device__ __forceinline__ uint32_t ntl_row_min16(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false); v = t < v ? t : v; /* row_ror:8 */
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xF, 0xF, false); v = t < v ? t : v; /* row_ror:4 */
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xF, 0xF, false); v = t < v ? t : v; /* row_ror:2 */
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xF, 0xF, false); v = t < v ? t : v; /* row_ror:1 */
    return v;
}

/* (acc << 1) | (a == b) */
__device__ __forceinline__ uint32_t ntl_shl1_or_eq(uint32_t acc, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cmp_eq_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %3, %3, vcc" : "=v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
    return r;
}

/* minimum / sum over the 4 lanes of a quad, result in every lane: DPP quad_perm [1,0,3,2] then [2,3,0,1] */
__device__ __forceinline__ uint32_t ntl_quad_min(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false); v = t < v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false); v = t < v ? t : v;
    return v;
}
__device__ __forceinline__ uint32_t ntl_quad_sum(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
    return v;
}

/* minimum over the 64 lanes of a wavefront, in every lane (uniform): the rows' minima, then four v_readlane and three s_min */
__device__ __forceinline__ uint32_t ntl_wave_min(uint32_t v)
{
    v = ntl_row_min16(v);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

__device__ __forceinline__ uint32_t ntl_row_max16(uint32_t v)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xF, 0xF, false); v = t > v ? t : v;
    t = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xF, 0xF, false); v = t > v ? t : v;
    return v;
}

/* x + x, kept an addition (the compiler turns it into the slower v_lshlrev_b32 x, 1) */
__device__ __forceinline__ uint32_t ntl_double(uint32_t x)
{
    uint32_t r;
    asm("v_add_u32 %0, %1, %1" : "=v"(r) : "v"(x));
    return r;
}

/* bit reversal of a 32-bit word: v_bfrev_b32 */
__device__ __forceinline__ uint32_t ntl_brev(uint32_t x) { return __builtin_bitreverse32(x); }

/* number of set bits of `mask` below this lane: v_mbcnt_lo/hi */
__device__ __forceinline__ uint32_t ntl_mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

/* x*d + k in IEEE double with two roundings (no FMA contraction): the reference computes
 * `args.x * abs(...) + args.k` in Python floats (bin/ntlink_utils.py:230-231). */
__device__ __forceinline__ double ntl_mul_add_rn(double x, double d, double k)
{
    return __dadd_rn(__dmul_rn(x, d), k);
}

/* streaming (non-temporal) 64-bit global accesses: data touched once should not evict reused lines from L2 */
__device__ __forceinline__ uint64_t ntl_stream_load(const uint64_t *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void ntl_stream_store(uint64_t *p, uint64_t v) { __builtin_nontemporal_store(v, p); }

/* inclusive prefix sum over the 64 lanes of a wavefront: four DPP row shifts inside the rows of 16, then the row totals
   broadcast forward (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six v_add_u32 with DPP operands */
__device__ __forceinline__ uint32_t ntl_wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false); /* row_shr:2 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false); /* row_shr:4 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false); /* row_shr:8 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false); /* row_bcast:15 -> rows 1, 3 */
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false); /* row_bcast:31 -> rows 2, 3 */
    return v;
}

/* Wave-synchronous hand-off through LDS: LDS operations of one wavefront execute in program order, so lanes of a wavefront may
   exchange data through LDS without a workgroup barrier -- the compiler only has to keep the accesses in order (no instruction). */
__device__ __forceinline__ void ntl_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* the value of the first active lane in every lane (an SGPR): v_readfirstlane_b32 */
__device__ __forceinline__ uint32_t ntl_readfirstlane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

/* four consecutive words from a pointer that is only 4-byte aligned: one global_load_dwordx4 */
struct __attribute__((packed, aligned(4))) ntl_u32x4_a4 { uint32_t x, y, z, w; };
__device__ __forceinline__ uint4 ntl_load4_a4(const uint32_t *p)
{
    const ntl_u32x4_a4 v = *(const ntl_u32x4_a4 *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

/* an 8-byte LDS read that stays where the source puts it (volatile: the optimiser would sink it next to its first use) */
__device__ __forceinline__ uint2 ntl_lds_load2_ordered(const uint2 *p)
{
    typedef __attribute__((address_space(3))) const volatile uint64_t lds_u64; /* said to be LDS: a volatile access through a generic pointer is a FLAT load */
    const uint64_t v = *(lds_u64 *)p;
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

/* a read-only u64 array that is SAID to live in LDS: where a kernel reads the same table from LDS or from global memory by a uniform
   flag, the compiler may merge the two reads into one through a selected generic pointer and then fails to lower its own cast
   ("Illegal instruction detected ... $src_shared_base", ROCm 7.2) -- pointers of two address spaces cannot be merged */
typedef __attribute__((address_space(3))) const uint64_t ntl_lds_cu64;
#define NTL_LDS_CU64(p) ((ntl_lds_cu64 *)(p))

/* hides a value's origin from the optimiser (no instruction) */
#define NTL_OPAQUE(v) asm volatile("" : "+v"(v))

/* *p++ = (v & ~mask) | T for a pointer into LDS and a constant T <= 64, as the three instructions it is: v_bfi_b32 with T as an
   inline constant, ds_write_b32, v_add_u32 (the compiler makes a v_mov of the constant, a temporary of the incremented pointer
   and a v_mov of that).  The write is not in the compiler's count of outstanding LDS operations; LDS operations of a wavefront
   complete in order, so a wait the compiler computes for one of ITS reads can only come out stricter, never laxer. */
template <int T>
__device__ __forceinline__ void ntl_lds_push_tagged(uint32_t *&p, uint32_t mask, uint32_t v)
{
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    lds_u32 *q = (lds_u32 *)p;
    uint32_t tmp;
    asm volatile("v_bfi_b32 %1, %2, %3, %4\n\tds_write_b32 %0, %1\n\tv_add_u32 %0, 4, %0" : "+v"(q), "=&v"(tmp) : "v"(mask), "n"(T), "v"(v) : "memory");
    p = (uint32_t *)q;
}

/* a - b, 0 where it would wrap: v_sub_u32 with the clamp bit */
__device__ __forceinline__ uint32_t ntl_sub_sat(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }

/* minimum of three: v_min3_u32 */
__device__ __forceinline__ uint32_t ntl_min3(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
}

/* bit i: k_i <= lim, bit 4 set: four compare + add-with-carry pairs as ONE asm statement (between separate statements the compiler
   pads with s_nop) */
__device__ __forceinline__ uint32_t ntl_le4_mask(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t lim)
{
    uint32_t acc = 1u;
    asm("v_cmp_le_u32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %2, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %3, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
        "v_cmp_le_u32 vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc"
        : "+v"(acc) : "v"(k3), "v"(k2), "v"(k1), "v"(k0), "v"(lim) : "vcc");
    return acc;
}

/* At most 96 scalar registers for the kernels of the MAIN stream (round 6).  A SIMD holds 800 of them and a wavefront is given its
   count rounded up to 16, plus 16 (MI355X_MICROARCH.md, residency): six window wavefronts (78 -> 96 each) leave 224, and with the 104-106
   the lookup and map kernels took when left to themselves (-> 128) only ONE of their wavefronts fitted beside them on a SIMD -- a CU's
   second lookup workgroup never became resident, which is why that kernel's time beside the window stage did not depend on how many
   workgroups it was given (profiles/HISTORY.md, round 6).  With 96 (-> 112) two fit. */
#define NTL_MAIN_STREAM_SGPRS __attribute__((amdgpu_num_sgpr(96)))

/* Wavefront issue priority 0..3 (s_setprio).  The latency-bound kernels of the MAIN stream (index lookup, mapping, gathers) raise
   theirs: beside the window stage's resident, issue-bound wavefronts their few instructions between two memory round trips
   should not queue behind a stream of rolling steps. */
#ifndef NTL_NO_SETPRIO
#define NTL_PRIO_LATENCY_BOUND() asm volatile("s_setprio 3")
#else
#define NTL_PRIO_LATENCY_BOUND() ((void)0)
#endif

/* eight bytes from any address: one global_load_dwordx2 (the target's unaligned access mode) */
__device__ __forceinline__ uint64_t ntl_load_u64_a1(const uint8_t *p)
{
    typedef uint64_t __attribute__((aligned(1))) u64_a1;
    return *(const __attribute__((address_space(1))) u64_a1 *)p; /* said to be global memory: through a generic pointer it is a FLAT load */
}

/* four bytes from / eight bytes to any address of global memory */
__device__ __forceinline__ uint32_t ntl_load_u32_a1(const uint8_t *p)
{
    typedef uint32_t __attribute__((aligned(1))) u32_a1;
    return *(const __attribute__((address_space(1))) u32_a1 *)p;
}
__device__ __forceinline__ void ntl_store_u64_a1(uint8_t *p, uint64_t v)
{
    typedef uint64_t __attribute__((aligned(1))) u64_a1;
    *(__attribute__((address_space(1))) u64_a1 *)p = v;
}
