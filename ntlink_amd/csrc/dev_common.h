/*
 * Device-side building blocks shared by the kernels: ntHash arithmetic on packed 2-bit
 * sequence, workgroup scans.
 *
 * ntHash restated from the published algorithm (the reference
 * uses it through btllib's `indexlr`, ntLink:199,223):
 *   fwd(p) = XOR_j srol^(k-1-j)(seed[s_(p+j)])      rev(p) = XOR_j srol^j(seed[comp s_(p+j)])
 *   h0 = fwd + rev (window minimum is taken on h0)   h1 = t ^ (t >> 27), t = h0 * (1 ^ k*MULTISEED)
 * srol rotates bits 0..32 and bits 33..63 left by one as two separate rings.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <dev_intrin.h> /* found through -I: csrc/ for hipcc, tests/sim/include for the SIMT mock */

#define NTL_INF 0xFFFFFFFFFFFFFFFFull
#define NTL_NONE 0xFFFFFFFFu
#define NTL_LEAD_PAD 16u /* bases of padding in front of the first sequence of a batch */

/* srol / sror on the two 32-bit halves (lo = bits 0..31, hi bit 0 = bit 32 of the 33-bit ring, hi bits 1..31 = the
 * 31-bit ring): 5 and 4 VALU instructions, every one written out (v_alignbit / v_bfi / v_lshl_or) */
__device__ __forceinline__ uint64_t srol1(uint64_t x)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t nlo = (lo << 1) | (hi & 1u);                 /* bit 32 -> bit 0 (v_and + v_lshl_or) */
    const uint32_t t = ntl_alignbit(hi, lo, 31);                /* (hi << 1) | (lo >> 31): bit 31 -> bit 32 */
    const uint32_t nhi = ntl_bfi<2u>(hi >> 30, t);              /* bit 63 -> bit 33 */
    return ((uint64_t)nhi << 32) | nlo;
}

__device__ __forceinline__ uint64_t sror1(uint64_t x)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t nlo = ntl_alignbit(hi, lo, 1);               /* bit 32 -> bit 31 */
    const uint32_t t = ntl_alignbit(hi >> 1, hi, 1);            /* (hi >> 1) with bit 33 -> bit 63 */
    const uint32_t nhi = ntl_bfi<1u>(lo, t);                    /* bit 0 -> bit 32 */
    return ((uint64_t)nhi << 32) | nlo;
}

/* rotate the 33-bit ring left by a (1..32) and the 31-bit ring left by b (1..30) on the halves: 7 VALU
 * instructions whatever the amounts (compile-time or uniform) */
__device__ __forceinline__ uint64_t srot_h(uint64_t x, uint32_t a, uint32_t b)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t y = ntl_alignbit(hi, lo, 1);                 /* bits 1..32 of the 33-bit ring */
    const uint32_t nlo = ntl_alignbit(lo, y, 32u - a);          /* (lo << a) | (ring33 >> (33 - a)) */
    const uint32_t b32 = (lo >> ((32u - a) & 31u)) & 1u;        /* ring bit 32-a -> bit 32 (a = 32: bit 0) */
    const uint32_t r31 = hi >> 1;
    const uint32_t o = ntl_alignbit(r31, hi & ~1u, 32u - b);    /* (r31 << b) | (r31 >> (31 - b)); bit 31 is dropped below */
    const uint32_t nhi = (o << 1) | b32;
    return ((uint64_t)nhi << 32) | nlo;
}

/* rotate the 33-bit ring left by a (0..32) and the 31-bit ring left by b (0..30) */
__device__ __forceinline__ uint64_t srot(uint64_t x, uint32_t a, uint32_t b)
{
    uint64_t r33 = x & 0x1FFFFFFFFull;
    uint32_t r31 = (uint32_t)(x >> 33);
    r33 = ((r33 << a) | (r33 >> (33u - a))) & 0x1FFFFFFFFull;
    r31 = ((r31 << b) | (b ? r31 >> (31u - b) : 0u)) & 0x7FFFFFFFu;
    return r33 | ((uint64_t)r31 << 33);
}

/* the same for amounts that may be zero (uniform): the halves form needs a, b >= 1 */
__device__ __forceinline__ uint64_t srot_u(uint64_t x, uint32_t a, uint32_t b)
{
    return (a == 0 || b == 0) ? srot(x, a, b) : srot_h(x, a, b);
}

/* 16 consecutive bases starting at global base index gp, base j in bits [2j, 2j+2) */
__device__ __forceinline__ uint32_t load_bases16(const uint32_t *__restrict__ packed, uint64_t gp)
{
    uint64_t wi = gp >> 4;
    uint32_t a = (uint32_t)gp & 15u;
    uint32_t w0 = packed[wi], w1 = packed[wi + 1];
    return ntl_alignbit(w1, w0, 2u * a);
}

__device__ __forceinline__ uint32_t load_base(const uint32_t *__restrict__ packed, uint64_t gp)
{
    return (packed[gp >> 4] >> (2u * ((uint32_t)gp & 15u))) & 3u;
}

/*
 * Hash of the k-mer starting at gp, from scratch, four bases per step:
 *   g4[byte] = {XOR_j srol^(3-j)(seed[b_j]), XOR_j sror^(3-j)(seed[3-b_j])},  b_j = (byte >> 2j) & 3
 *   f = srol^4(f) ^ g4[byte][0]          (Horner form of fwd)
 *   u = sror^4(u) ^ g4[byte][1]          (u = XOR_j sror^(k-1-j)(seedc_j);  rev = srol^(k-1)(u))
 * The k % 4 trailing bases take single steps with seed_tab[c] = {seed[c], seed[3-c]}.
 * g4 (4 KB, k-independent) is read from global memory and lives in L1.
 */
__device__ __forceinline__ void hash_init_loop(const uint32_t *__restrict__ packed, uint64_t gp, int k,
                                               const uint64_t (*__restrict__ g4)[2], const uint64_t (*seed_tab)[2],
                                               uint64_t &fwd, uint64_t &rev)
{
    uint64_t f = 0, u = 0;
    for (int i = 0; i < k; i += 16) {
        const uint32_t s = load_bases16(packed, gp + (uint64_t)i);
        const int nb = k - i < 16 ? k - i : 16;
        const int ng = nb >> 2;
        for (int g = 0; g < ng; g++) {
            const uint32_t byte = (s >> (8 * g)) & 255u;
            const uint64_t gf = g4[byte][0], gu = g4[byte][1];
            f = srot_h(f, 4, 4) ^ gf;
            u = srot_h(u, 29, 27) ^ gu;
        }
        for (int j = ng * 4; j < nb; j++) {
            const uint32_t c = (s >> (2 * j)) & 3u;
            f = srol1(f) ^ seed_tab[c][0];
            u = sror1(u) ^ seed_tab[c][1];
        }
    }
    fwd = f;
    rev = srot_u(u, (uint32_t)(k - 1) % 33u, (uint32_t)(k - 1) % 31u);
}

/* k <= 64: straight-line form, EIGHT bases per step through g8 (65536 entries x 16 B = 1 MB, L2-resident,
 * k-independent: g8[w16] = {XOR_j srol^(7-j)(seed[b_j]), XOR_j sror^(7-j)(seed[3-b_j])}, b_j = (w16 >> 2j) & 3):
 *   f = srol^8(f) ^ g8[w16].f,  u = sror^8(u) ^ g8[w16].u
 * then one four-base step (g4) if k % 8 >= 4 and single steps for the last k % 4 bases.  All table
 * entries are requested before the first one is used: one memory latency per init. */
__device__ __forceinline__ void hash_init(const uint32_t *__restrict__ packed, uint64_t gp, int k,
                                          const uint64_t (*__restrict__ g8)[2], const uint64_t (*g4)[2],
                                          const uint64_t (*seed_tab)[2], uint64_t &fwd, uint64_t &rev)
{
    if (k > 64) { hash_init_loop(packed, gp, k, g4, seed_tab, fwd, rev); return; }
    const uint64_t wi = gp >> 4;
    const uint32_t a2 = 2u * ((uint32_t)gp & 15u);
    uint32_t raw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) raw[i] = (16 * i < k + 16) ? packed[wi + i] : 0u; /* k + 15 bases at most */
    uint32_t s[4];
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = ntl_alignbit(raw[i + 1], raw[i], a2);
    const int n8 = k >> 3;
    uint64_t gf[8], gu[8];
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const uint32_t w16 = (s[g >> 1] >> (16 * (g & 1))) & 0xFFFFu;
        const bool on = g < n8;
        gf[g] = on ? g8[w16][0] : 0ull;
        gu[g] = on ? g8[w16][1] : 0ull;
    }
    uint64_t f = 0, u = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        if (g < n8) { /* uniform */
            f = srot_h(f, 8, 8) ^ gf[g];
            u = srot_h(u, 25, 23) ^ gu[g];
        }
    }
    int j = n8 * 8;
    if (k - j >= 4) {
        const uint32_t byte = (s[(j >> 4) & 3] >> (2 * (j & 15))) & 255u;
        f = srot_h(f, 4, 4) ^ g4[byte][0];
        u = srot_h(u, 29, 27) ^ g4[byte][1];
        j += 4;
    }
    for (; j < k; j++) { /* k % 4 trailing bases */
        const uint32_t c = load_base(packed, gp + (uint64_t)j);
        f = srol1(f) ^ seed_tab[c][0];
        u = sror1(u) ^ seed_tab[c][1];
    }
    fwd = f;
    rev = srot_u(u, (uint32_t)(k - 1) % 33u, (uint32_t)(k - 1) % 31u);
}

/* The same with four bases per step only (g4, e.g. an LDS copy): for callers whose table lookups should
 * not leave the CU (the emit kernel: few k-mers per lane, latency-bound). */
__device__ __forceinline__ void hash_init_g4(const uint32_t *__restrict__ packed, uint64_t gp, int k,
                                             const uint64_t (*g4)[2], const uint64_t (*seed_tab)[2],
                                             uint64_t &fwd, uint64_t &rev)
{
    if (k > 64) { hash_init_loop(packed, gp, k, g4, seed_tab, fwd, rev); return; }
    const uint64_t wi = gp >> 4;
    const uint32_t a2 = 2u * ((uint32_t)gp & 15u);
    uint32_t raw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) raw[i] = (16 * i < k + 16) ? packed[wi + i] : 0u;
    uint32_t s[4];
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = ntl_alignbit(raw[i + 1], raw[i], a2);
    const int ng = k >> 2;
    uint64_t f = 0, u = 0;
#pragma unroll
    for (int g = 0; g < 16; g++) {
        if (g < ng) { /* uniform */
            const uint32_t byte = (s[g >> 2] >> (8 * (g & 3))) & 255u;
            f = srot_h(f, 4, 4) ^ g4[byte][0];
            u = srot_h(u, 29, 27) ^ g4[byte][1];
        }
    }
    for (int j = ng * 4; j < k; j++) {
        const uint32_t c = load_base(packed, gp + (uint64_t)j);
        f = srol1(f) ^ seed_tab[c][0];
        u = sror1(u) ^ seed_tab[c][1];
    }
    fwd = f;
    rev = srot_u(u, (uint32_t)(k - 1) % 33u, (uint32_t)(k - 1) % 31u);
}

/* The same with TWO four-base groups per rotation: with g4r[b] = {srot^4(g4[b].f), sror^4(g4[b].u)} beside g4,
 *   f' = srot^4(srot^4(f) ^ g4[b0].f) ^ g4[b1].f = srot^8(f) ^ g4r[b0].f ^ g4[b1].f      (the rotations are linear over XOR)
 * -- half the split rotations (7 VALU instructions each) of hash_init_g4; a group that is left over (k % 8 >= 4) takes the single step. */
__device__ __forceinline__ void hash_init_g4p(const uint32_t *__restrict__ packed, uint64_t gp, int k,
                                              const uint64_t (*g4)[2], const uint64_t (*g4r)[2], const uint64_t (*seed_tab)[2],
                                              uint64_t &fwd, uint64_t &rev)
{
    if (k > 64) { hash_init_loop(packed, gp, k, g4, seed_tab, fwd, rev); return; }
    const uint64_t wi = gp >> 4;
    const uint32_t a2 = 2u * ((uint32_t)gp & 15u);
    uint32_t raw[5];
#pragma unroll
    for (int i = 0; i < 5; i++) raw[i] = (16 * i < k + 16) ? packed[wi + i] : 0u;
    uint32_t s[4];
#pragma unroll
    for (int i = 0; i < 4; i++) s[i] = ntl_alignbit(raw[i + 1], raw[i], a2);
    const int ng = k >> 2, np = ng >> 1;
    uint64_t f = 0, u = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (q < np) { /* uniform */
            const uint32_t w16 = s[q >> 1] >> (16 * (q & 1));
            const uint32_t b0 = w16 & 255u, b1 = (w16 >> 8) & 255u;
            f = srot_h(f, 8, 8) ^ g4r[b0][0] ^ g4[b1][0];
            u = srot_h(u, 25, 23) ^ g4r[b0][1] ^ g4[b1][1];
        }
    }
    if (ng & 1) {
        const int g = ng - 1;
        const uint32_t byte = (s[(g >> 2) & 3] >> (8 * (g & 3))) & 255u;
        f = srot_h(f, 4, 4) ^ g4[byte][0];
        u = srot_h(u, 29, 27) ^ g4[byte][1];
    }
    for (int j = ng * 4; j < k; j++) {
        const uint32_t c = load_base(packed, gp + (uint64_t)j);
        f = srol1(f) ^ seed_tab[c][0];
        u = sror1(u) ^ seed_tab[c][1];
    }
    fwd = f;
    rev = srot_u(u, (uint32_t)(k - 1) % 33u, (uint32_t)(k - 1) % 31u);
}

/* hash_init_g4p for U k-mers at once (round 5).  hash_init_g4p's loads sit behind uniform branches on k (which words a k-mer needs),
 * so two calls in a row are compiled as two chains, one behind the other: nothing of the second k-mer is asked for before the first
 * is hashed.  Here the base words of every k-mer are loaded in front of everything else (`packed` carries 4096 bases of padding behind its
 * last sequence: no bounds to check), and each uniform branch holds the step of ALL U k-mers: U memory round trips in
 * flight instead of one. */
template <int U>
__device__ __forceinline__ void hash_init_g4p_multi(const uint32_t *__restrict__ packed, const uint64_t (&gp)[U], int k,
                                                    const uint64_t (*g4)[2], const uint64_t (*g4r)[2], const uint64_t (*seed_tab)[2],
                                                    uint64_t (&fwd)[U], uint64_t (&rev)[U])
{
    if (k > 64) {
#pragma unroll
        for (int x = 0; x < U; x++) hash_init_loop(packed, gp[x], k, g4, seed_tab, fwd[x], rev[x]);
        return;
    }
    /* four words in ONE load per k-mer (a k-mer of up to 49 bases lies in them wherever it starts), the fifth only for longer k: every
       lane's k-mer sits in another cache line, so a load instruction is 64 line lookups in the CU's vector cache whatever its width --
       and those lookups, not the latency, are what the lookup kernel's time is made of (five one-word loads: 22.4 ms per C3 step
       alone; three or four as before: 20.7; one: see DESIGN 4.2) */
    uint32_t raw[U][5];
#pragma unroll
    for (int x = 0; x < U; x++) {
        const uint4 v = ntl_load4_a4(packed + (gp[x] >> 4));
        raw[x][0] = v.x; raw[x][1] = v.y; raw[x][2] = v.z; raw[x][3] = v.w; raw[x][4] = 0u;
    }
    if (k > 49) {
#pragma unroll
        for (int x = 0; x < U; x++) raw[x][4] = packed[(gp[x] >> 4) + 4];
    }
    uint32_t s[U][4];
#pragma unroll
    for (int x = 0; x < U; x++) {
        const uint32_t a2 = 2u * ((uint32_t)gp[x] & 15u);
#pragma unroll
        for (int i = 0; i < 4; i++) s[x][i] = ntl_alignbit(raw[x][i + 1], raw[x][i], a2);
    }
    const int ng = k >> 2, np = ng >> 1;
    uint64_t f[U], u[U];
#pragma unroll
    for (int x = 0; x < U; x++) { f[x] = 0; u[x] = 0; }
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (q < np) { /* uniform */
#pragma unroll
            for (int x = 0; x < U; x++) {
                const uint32_t w16 = s[x][q >> 1] >> (16 * (q & 1));
                const uint32_t b0 = w16 & 255u, b1 = (w16 >> 8) & 255u;
                f[x] = srot_h(f[x], 8, 8) ^ g4r[b0][0] ^ g4[b1][0];
                u[x] = srot_h(u[x], 25, 23) ^ g4r[b0][1] ^ g4[b1][1];
            }
        }
    }
    if (ng & 1) {
        const int g = ng - 1;
#pragma unroll
        for (int x = 0; x < U; x++) {
            const uint32_t byte = (g < 4 ? s[x][0] : (g < 8 ? s[x][1] : (g < 12 ? s[x][2] : s[x][3]))) >> (8 * (g & 3)) & 255u;
            f[x] = srot_h(f[x], 4, 4) ^ g4[byte][0];
            u[x] = srot_h(u[x], 29, 27) ^ g4[byte][1];
        }
    }
    for (int j = ng * 4; j < k; j++) {
#pragma unroll
        for (int x = 0; x < U; x++) {
            const uint32_t c = load_base(packed, gp[x] + (uint64_t)j);
            f[x] = srol1(f[x]) ^ seed_tab[c][0];
            u[x] = sror1(u[x]) ^ seed_tab[c][1];
        }
    }
#pragma unroll
    for (int x = 0; x < U; x++) {
        fwd[x] = f[x];
        rev[x] = srot_u(u[x], (uint32_t)(k - 1) % 33u, (uint32_t)(k - 1) % 31u);
    }
}

/* Exclusive scan of one value per thread over a workgroup of NT threads; returns the prefix and
 * the workgroup total.  s_tmp must hold NT entries. */
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_tmp, uint32_t &total)
{
    /* a DPP scan inside every wavefront, the wavefront totals through LDS: two barriers (the log-step scan over LDS this
       replaces took eighteen for 256 threads, in kernels whose time is the latency of exactly such steps) */
    static_assert(NT % 64 == 0 && NT <= 1024, "whole wavefronts");
    const int t = threadIdx.x;
    const uint32_t incl = ntl_wave_incl_scan(v);
    if ((t & 63) == 63) s_tmp[t >> 6] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int wv = 0; wv < NT / 64; wv++) {
        const uint32_t x = s_tmp[wv];
        before += wv < (t >> 6) ? x : 0u;
        all += x;
    }
    total = all;
    __syncthreads(); /* s_tmp may be written again */
    return before + incl - v;
}
