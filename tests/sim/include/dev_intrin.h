/* Portable bodies of the gfx950 instruction wrappers, for the SIMT mock (tests/sim). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

inline uint32_t ntl_alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (sh & 31));
}

inline uint32_t ntl_bfe(uint32_t x, uint32_t off, uint32_t width) { return (x >> off) & ((1u << width) - 1u); }

template <uint32_t MASK> inline uint32_t ntl_bfi(uint32_t a, uint32_t b) { return (a & MASK) | (b & ~MASK); }

inline uint32_t ntl_shl1_or_ne(uint32_t acc, uint32_t a, uint32_t b) { return (acc << 1) | (a != b ? 1u : 0u); }
inline uint32_t ntl_shl1_or_le(uint32_t acc, uint32_t a, uint32_t b) { return (acc << 1) | (a <= b ? 1u : 0u); }

inline uint32_t ntl_shl1_or_lt_diff(uint32_t acc, uint32_t a, uint32_t b, uint32_t &d)
{
    d = a - b;
    return (acc << 1) | (a < b ? 1u : 0u);
}

inline uint32_t ntl_row_min16(uint32_t v)
{
    const int l = (int)(sim::tid & 63u);
    for (int d = 8; d >= 1; d >>= 1) {
        const uint32_t t = __shfl(v, (l & ~15) | ((l + d) & 15));
        v = t < v ? t : v;
    }
    return v;
}

inline uint32_t ntl_wave_min(uint32_t v)
{
    const int l = (int)(sim::tid & 63u);
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t t = __shfl(v, (l + d) & 63);
        v = t < v ? t : v;
    }
    return v;
}

inline uint32_t ntl_shl1_or_eq(uint32_t acc, uint32_t a, uint32_t b) { return (acc << 1) | (a == b ? 1u : 0u); }

inline uint32_t ntl_quad_min(uint32_t v)
{
    const int l = (int)(sim::tid & 63u);
    for (int d = 1; d <= 2; d <<= 1) {
        const uint32_t t = __shfl(v, l ^ d);
        v = t < v ? t : v;
    }
    return v;
}

inline uint32_t ntl_quad_sum(uint32_t v)
{
    const int l = (int)(sim::tid & 63u);
    for (int d = 1; d <= 2; d <<= 1) v += __shfl(v, l ^ d);
    return v;
}

inline uint32_t ntl_row_max16(uint32_t v)
{
    const int l = (int)(sim::tid & 63u);
    for (int d = 8; d >= 1; d >>= 1) {
        const uint32_t t = __shfl(v, (l & ~15) | ((l + d) & 15));
        v = t > v ? t : v;
    }
    return v;
}

inline uint32_t ntl_double(uint32_t x) { return x + x; }

inline uint32_t ntl_brev(uint32_t x)
{
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}

inline uint32_t ntl_mbcnt(unsigned long long mask)
{
    unsigned l = sim::tid & 63;
    return (uint32_t)__builtin_popcountll(mask & ((1ull << l) - 1));
}

inline double ntl_mul_add_rn(double x, double d, double k)
{
    volatile double m = x * d; /* built with -ffp-contract=off; volatile keeps the two roundings */
    return m + k;
}

inline uint64_t ntl_stream_load(const uint64_t *p) { return *p; }
inline void ntl_stream_store(uint64_t *p, uint64_t v) { *p = v; }

inline uint32_t ntl_wave_incl_scan(uint32_t v)
{
    const int l = (int)(sim::tid & 63u);
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl(v, l >= d ? l - d : l);
        if (l >= d) v += t;
    }
    return v;
}

inline void ntl_wave_sync() { pthread_barrier_wait(&sim::cur->wbar[sim::tid >> 6]); }
inline uint32_t ntl_readfirstlane(uint32_t v) { return __shfl(v, 0); }
inline uint4 ntl_load4_a4(const uint32_t *p) { return make_uint4(p[0], p[1], p[2], p[3]); }
inline uint2 ntl_lds_load2_ordered(const uint2 *p) { return *p; }
typedef const uint64_t ntl_lds_cu64;
#define NTL_LDS_CU64(p) ((ntl_lds_cu64 *)(p))
#define NTL_OPAQUE(v) ((void)(v))
template <int T> inline void ntl_lds_push_tagged(uint32_t *&p, uint32_t mask, uint32_t v) { *p++ = (v & ~mask) | (uint32_t)T; }
inline uint32_t ntl_sub_sat(uint32_t a, uint32_t b) { return a > b ? a - b : 0u; }
inline uint32_t ntl_min3(uint32_t a, uint32_t b, uint32_t c) { const uint32_t m = a < b ? a : b; return m < c ? m : c; }
inline uint32_t ntl_le4_mask(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t lim)
{
    return 16u | (k3 <= lim ? 8u : 0u) | (k2 <= lim ? 4u : 0u) | (k1 <= lim ? 2u : 0u) | (k0 <= lim ? 1u : 0u);
}
#define NTL_PRIO_LATENCY_BOUND() ((void)0)
#define NTL_MAIN_STREAM_SGPRS
inline uint64_t ntl_load_u64_a1(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t ntl_load_u32_a1(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline void ntl_store_u64_a1(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
