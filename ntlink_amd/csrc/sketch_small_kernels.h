/*
 * sketch_small_kernel<W, MULTI>: the exact window pass for SMALL windows, 2 <= w <= 15 -- the sketches of the stages around `pair`
 * (ntLink:243-251 overlap k15 w5, bin/ntlink_patch_gaps.py:417-441 k20 w10: SURVEY rows f3 / f4).  Round 6, VERDICT r5 item 8 (b).
 *
 * What it replaces for these windows: sketch_mask_kernel<4 | 1, 256, ..> -- four (w >= 4) or one k-mer per lane, so a from-scratch hash
 * (table lookups in L2) per four k-mers, the window's elements read back from LDS one by one and one LDS atomic per window.  The
 * threshold pass of the large windows has nothing to sparsify here: at w = 5 every third k-mer is a minimizer.
 *
 * Same strips as every other pass (strip_table_kernel: 4096 consecutive k-mer ordinals, 256 lanes x 16; a strip owns the windows that
 * start at its elements 1 .. NWO, NWO = 4096 - 17: the last lane only supplies elements), same 64-bit hashes, same result: bit g of
 * the global bitmask is set iff the k-mer at base g is the rightmost minimum of some window of w consecutive k-mers of its sequence
 * (btllib Indexlr as pinned by the reference's golden TSVs; the positions such windows' minima take are monotone, so the set of
 * distinct minima IS the emitted list and OR-ing a bit per window needs no "the minimum moved" test).  Who does what:
 *   1. a lane hashes its first k-mer from scratch and rolls 15 steps: 16 k-mers per from-scratch hash; the seed-table index of a
 *      step (leaving base + 4 x entering base) is a nibble of two words made once per lane;
 *   2. the first w - 1 hashes of every lane go through LDS to the lane on its left -- the only exchange, one barrier;
 *   3. the minima of the 16 windows that start in the lane's block, in REGISTERS, w known at compile time: over the lane's 16 + w - 1
 *      hashes the prefix and suffix minima of blocks of w elements (a window = the suffix of its block from its first element and
 *      the prefix of the next block up to its last: 2.8 minima per window whatever w; w = 2: one pairwise minimum per element);
 *      ties go to the right-hand operand (Indexlr keeps the rightmost of equal hashes -- identical k-mers in a window are
 *      common at these sizes); every entry carries its element number;
 *   4. a lane's windows set bits in a 31-bit word of its own, which goes into the strip's LDS bitmask with two atomics per LANE, and
 *      the strip's 128 words into the global bitmask as before.
 * MULTI = true: strips that cross non-ACGT runs -- a lane walks the run table (as sketch_mask_kernel<.., MULTI = true> does) and
 * positions come from LDS; rare, and launched only for batches that have such strips.
 * A hash of 2^64 - 1 is never a minimum (see 3 in the kernel).
 */
#pragma once
#include "sketch_kernels.h"

/* the smaller of two entries, `b` lying to the right of `a` (or reaching further right): ties go to b */
__device__ __forceinline__ void small_min_right(uint64_t &ah, uint32_t &ap, const uint64_t bh, const uint32_t bp)
{
    const bool take = bh <= ah;
    ah = take ? bh : ah;
    ap = take ? bp : ap;
}

#ifndef NTL_SMALL_VH_FROM
#define NTL_SMALL_VH_FROM 3 /* windows from this size on take their minima from block prefixes and suffixes (below) */
#endif

/* wavefronts per SIMD a window size is compiled for: 66 / 74 / 90 / 104 vector registers at w = 3 / 5 / 10 / 15 left to the compiler;
   held to 6 wavefronts the smallest windows gain a few per cent and the larger ones spill (w = 10: 724 -> 621 Gbases/s, w = 12: 718 ->
   413; profiles/r07g_small_window_launch_bounds_block_minima.txt, and r07b_* for the doubling-table form) */
template <int W>
constexpr int small_waves_per_simd()
{
#ifdef NTL_SMALL_WPE_ALL
    return NTL_SMALL_WPE_ALL; /* experiments: tools/build_variant.py */
#else
    return W <= 5 ? 6 : 1;
#endif
}

template <int W, bool MULTI>
__global__ __launch_bounds__(256, (small_waves_per_simd<W>())) void sketch_small_kernel(SketchArgs A)
{
    static_assert(W >= 2 && W <= 15, "windows of 2 .. 15 k-mers");
    constexpr int C = 16, NT = 256;
    constexpr int X = W - 1;                 /* elements of the next lane's block a window of this lane may reach */
    constexpr int N = C + X;                 /* elements a lane looks at */
    constexpr int NBW = C * NT / 32;         /* words of the strip-local bitmask */
    constexpr int P2 = W >= 8 ? 8 : (W >= 4 ? 4 : 2); /* the largest power of two <= W */
    __shared__ uint64_t s_h[(MULTI ? C : X) * NT]; /* element t of lane L at [t * NT + L] (MULTI: all sixteen, the lanes' walk stores them) */
    __shared__ uint32_t s_pos[MULTI ? C * NT : 1];  /* MULTI: position in the sequence of element e at [e] */
    __shared__ uint32_t s_bits[NBW + 1];
    __shared__ uint64_t s_roll[16][2], s_seed[4][2];

    /* consecutive strips (which share halo bases and strip-table lines) go to one XCD (sketch_mask_kernel) */
    const uint32_t per_xcd = gridDim.x >> 3; /* the grid is a multiple of 8 */
    const uint32_t strip = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const int L = threadIdx.x;
    const SketchGeom G = A.G;
    if (strip >= A.nstrips) return;
    const StripInfo I = A.strip_tab[strip];
    if (I.seq == NTL_NONE) return; /* the grid is an upper bound of the number of strips */
    if ((I.multi != 0) != MULTI) return;
    if (L < 16) { s_roll[L][0] = A.roll_tab[L][0]; s_roll[L][1] = A.roll_tab[L][1]; }
    if (L < 4) { s_seed[L][0] = A.seed_tab[L][0]; s_seed[L][1] = A.seed_tab[L][1]; }
    if (L <= NBW) s_bits[L] = 0;
    __syncthreads();

    const int64_t e_lane = (int64_t)I.E0 + (int64_t)L * C; /* ordinal of this lane's element 0 */
    uint64_t H[N];
    uint32_t P[N];

    /* ---- 1: the hashes of the lane's 16 k-mers ---- */
    if (!MULTI) {
#pragma unroll
        for (int t = 0; t < C; t++) H[t] = NTL_INF;
        if (e_lane < (int64_t)I.M) {
            /* (element -1 of a sequence's first strip is virtual: hashed like a real k-mer -- the bases in front of the sequence are
               padding or the previous sequence -- so that rolling out of it is exact, then voided) */
            const uint64_t gp = (uint64_t)((int64_t)I.base + I.P0 + (int64_t)L * C);
            uint64_t fwd, rev;
            hash_init(A.T.packed, gp, G.k, A.g8, A.g4, s_seed, fwd, rev);
            H[0] = fwd + rev;
            const uint32_t so = load_bases16(A.T.packed, gp);
            const uint32_t si = load_bases16(A.T.packed, gp + (uint64_t)G.k);
            /* nibble i of ze: the seed-table index of the step over base 2 i, of zo: over base 2 i + 1 */
            const uint32_t ze = (so & 0x33333333u) | ((si << 2) & 0xCCCCCCCCu);
            const uint32_t zo = ((so >> 2) & 0x33333333u) | (si & 0xCCCCCCCCu);
#pragma unroll
            for (int t = 1; t < C; t++) {
                const uint32_t z = ((t - 1) & 1) ? zo : ze;
                const uint32_t idx = (z >> (4 * ((t - 1) >> 1))) & 15u;
                fwd = srol1(fwd) ^ s_roll[idx][0];
                rev = sror1(rev ^ s_roll[idx][1]);
                H[t] = fwd + rev;
            }
            if (e_lane < 0 || e_lane + C > (int64_t)I.M) { /* strip edges only */
#pragma unroll
                for (int t = 0; t < C; t++) {
                    const int64_t e = e_lane + t;
                    if (e < 0 || e >= (int64_t)I.M) H[t] = NTL_INF;
                }
            }
        }
    } else {
        /* walk the run table; the hash starts again at every run crossing */
        uint32_t g = I.run;
        const uint32_t g1 = A.T.seq_run_first[I.seq + 1];
        uint64_t fwd = 0, rev = 0, gp = 0;
        uint32_t run_end = 0; /* ordinal one past the current run */
        bool have = false;
        for (int t = 0; t < C; t++) {
            const int64_t e = e_lane + t;
            uint64_t hv = NTL_INF;
            uint32_t pv = 0;
            if (e >= 0 && e < (int64_t)I.M) {
                const uint32_t eo = (uint32_t)e;
                if (!have || eo >= run_end) {
                    while (g + 1 < g1 && (A.run_n[g] == 0 || eo >= A.run_ord[g] + A.run_n[g])) g++;
                    run_end = A.run_ord[g] + A.run_n[g];
                    pv = A.T.run_start[g] + (eo - A.run_ord[g]);
                    gp = I.base + pv;
                    hash_init(A.T.packed, gp, G.k, A.g8, A.g4, s_seed, fwd, rev);
                    have = true;
                } else {
                    const uint32_t cin = load_base(A.T.packed, gp + (uint64_t)G.k);
                    const uint32_t cout = load_base(A.T.packed, gp);
                    const uint32_t idx = (cin << 2) | cout;
                    fwd = srol1(fwd) ^ s_roll[idx][0];
                    rev = sror1(rev ^ s_roll[idx][1]);
                    gp++;
                    pv = (uint32_t)(gp - I.base);
                }
                hv = fwd + rev;
            }
            s_h[t * NT + L] = hv;
            s_pos[L * C + t] = pv;
        }
#pragma unroll
        for (int t = 0; t < C; t++) H[t] = s_h[t * NT + L];
    }

    /* ---- 2: the first w - 1 hashes of every lane to the lane on its left ---- */
    if (!MULTI) {
#pragma unroll
        for (int t = 0; t < X; t++) s_h[t * NT + L] = H[t];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < X; t++) H[C + t] = L + 1 < NT ? s_h[t * NT + L + 1] : NTL_INF;

    /* ---- 3: the minima of the windows that start at the lane's 16 elements ---- */
    /* A hash of 2^64 - 1 is never a minimum (btllib's "no hash yet" value).  Elements outside the sequence have it, but the windows
       that reach them are not real windows and set no bit; what is left is a real window whose w k-mers ALL hash to 2^64 - 1.  Such
       a window holds one of every P2 <= w consecutive elements: a test of those, a vote, and -- never, in practice -- every such
       element gets the number 31, the bit of a lane's word that is not flushed */
    bool some_inf = false;
#pragma unroll
    for (int i = 0; i < N; i += P2) some_inf |= H[i] == NTL_INF;
    const bool wave_inf = __ballot(some_inf) != 0ull;
#pragma unroll
    for (int i = 0; i < N; i++) P[i] = (uint32_t)i;
    if (wave_inf) {
#pragma unroll
        for (int i = 0; i < N; i++) P[i] = H[i] == NTL_INF ? 31u : (uint32_t)i;
    }
    /* w >= NTL_SMALL_VH_FROM: blocks of w elements; the minimum of every element's prefix of its block (left to right) and of its
       suffix (right to left, in place).  A window that starts at element j of a block is that suffix at j and the prefix at
       j + w - 1 in the next block (the suffix alone where it starts a block): about 2.8 minima per window whatever w -- 45 per lane
       at w = 10 -- where the doubling table (spans 1, 2, 4, 8 in place, a window = the two spans of the largest power of two that
       cover it) takes 3.3 (w = 5), 5 (w = 10: 80 per lane), 6 (w = 15).  w = 2: the doubling table (one level). */
    constexpr bool VH = W >= NTL_SMALL_VH_FROM;
    constexpr int NF = VH ? N : 1;
    uint64_t Fh[NF];
    uint32_t Fp[NF];
    if constexpr (VH) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (i % W == 0) { Fh[i] = H[i]; Fp[i] = P[i]; }
            else {
                const bool take = H[i] <= Fh[i - 1]; /* element i lies to the right of its prefix: a tie goes to it */
                Fh[i] = take ? H[i] : Fh[i - 1];
                Fp[i] = take ? P[i] : Fp[i - 1];
            }
        }
#pragma unroll
        for (int i = N - 2; i >= 0; i--)
            if (i % W != W - 1) small_min_right(H[i], P[i], H[i + 1], P[i + 1]); /* entry i + 1 holds its suffix already */
    } else {
#pragma unroll
        for (int s = 1; 2 * s <= W; s *= 2) {
#pragma unroll
            for (int i = 0; i + s < N; i++) small_min_right(H[i], P[i], H[i + s], P[i + s]); /* in place: entry i + s still holds the span s */
        }
    }
    /* windows [s, s + w) in strip elements, s = 16 L + j: real where they start at a k-mer of the sequence (element 0 of a sequence's
       first strip is not one) and end inside it; the strip's own where s <= NWO (s = 0 is the previous strip's last window: the same
       bit twice) */
    const int64_t last_start = (int64_t)I.M - (int64_t)G.w - (int64_t)I.E0; /* s <= last_start: the window ends inside the sequence */
    const int s0 = L * C;
    const bool owns = s0 <= G.NWO && (int64_t)s0 <= last_start;
    const bool inside = e_lane >= 0 && s0 + (C - 1) <= G.NWO && (int64_t)(s0 + C - 1) <= last_start;
    const bool all_inside = __ballot(owns && !inside) == 0ull;
    uint32_t bits = 0;
    if (owns) {
#pragma unroll
        for (int j = 0; j < C; j++) {
            uint64_t mh = H[j];
            uint32_t mp = P[j];
            if constexpr (VH) {
                if (j % W) small_min_right(mh, mp, Fh[VH ? j + W - 1 : 0], Fp[VH ? j + W - 1 : 0]);
            } else if (W != P2) small_min_right(mh, mp, H[j + W - P2], P[j + W - P2]);
            if (all_inside) bits |= 1u << mp;
            else {
                const int s = s0 + j;
                const bool valid = e_lane + j >= 0 && s <= G.NWO && (int64_t)s <= last_start;
                bits |= valid ? 1u << mp : 0u;
            }
        }
    }
    bits &= 0x7FFFFFFFu; /* (bit 31: an element that is never a minimum) */

    /* ---- 4: the lane's word (elements 16 L .. 16 L + 30) into the strip's bitmask, the strip's into the global one ---- */
    if (bits) {
        const uint32_t sh = 16u * ((uint32_t)L & 1u);
        atomicOr(&s_bits[L >> 1], bits << sh);
        if (sh && (bits >> 16)) atomicOr(&s_bits[(L >> 1) + 1], bits >> 16);
    }
    __syncthreads();
    if (L < NBW) {
        uint32_t word = s_bits[L];
        if (!MULTI) {
            if (word) { /* strip element 32 L starts at global bit g0 */
                const uint64_t g0 = (uint64_t)((int64_t)I.base + I.P0 + 32 * (int64_t)L);
                const uint32_t sh = (uint32_t)g0 & 31u;
                atomicOr(&A.mask[g0 >> 5], word << sh);
                if (sh && (word >> (32u - sh))) atomicOr(&A.mask[(g0 >> 5) + 1], word >> (32u - sh));
            }
        } else {
            while (word) {
                const uint32_t b = (uint32_t)__ffs(word) - 1u;
                word &= word - 1u;
                const uint64_t g = I.base + s_pos[32u * (uint32_t)L + b];
                atomicOr(&A.mask[g >> 5], 1u << ((uint32_t)g & 31u));
            }
        }
    }
}
