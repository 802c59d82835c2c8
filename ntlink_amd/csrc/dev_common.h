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

__device__ __forceinline__ uint64_t srol1(uint64_t x)
{
    uint64_t m = ((x & 0x8000000000000000ull) >> 30) | ((x & 0x100000000ull) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFull) | m;
}

__device__ __forceinline__ uint64_t sror1(uint64_t x)
{
    uint64_t m = ((x & 0x200000000ull) << 30) | ((x & 1ull) << 32);
    return ((x >> 1) & 0xFFFFFFFEFFFFFFFFull) | m;
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

/* seed_tab[c] = {seed[c], seed[3-c]}.  Hash of the k-mer starting at gp, from scratch. */
__device__ __forceinline__ void hash_init(const uint32_t *__restrict__ packed, uint64_t gp, int k,
                                          const uint64_t (*seed_tab)[2], uint64_t &fwd, uint64_t &rev)
{
    uint64_t f = 0, u = 0;
    for (int i = 0; i < k; i += 16) {
        uint32_t s = load_bases16(packed, gp + (uint64_t)i);
        int nb = k - i < 16 ? k - i : 16;
        for (int j = 0; j < nb; j++) {
            uint32_t c = (s >> (2 * j)) & 3u;
            f = srol1(f) ^ seed_tab[c][0];
            u = sror1(u) ^ seed_tab[c][1]; /* u = XOR_j sror^(k-1-j)(seedc_j) */
        }
    }
    for (int i = 1; i < k; i++) u = srol1(u); /* rev = srol^(k-1)(u) */
    fwd = f;
    rev = u;
}

/* Exclusive scan of one value per thread over a workgroup of NT threads; returns the prefix and
 * the workgroup total.  s_tmp must hold NT entries. */
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *s_tmp, uint32_t &total)
{
    const int t = threadIdx.x;
    s_tmp[t] = v;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) {
        uint32_t add = t >= d ? s_tmp[t - d] : 0u;
        __syncthreads();
        s_tmp[t] += add;
        __syncthreads();
    }
    uint32_t incl = s_tmp[t];
    total = s_tmp[NT - 1];
    __syncthreads();
    return incl - v;
}
