/*
 * Sketch kernels: packed 2-bit sequence -> (k,w) minimizers.
 *
 * Replaces btllib `indexlr --long --pos --strand` (ntLink:199,223) = NtHash(seq,2,k) +
 * Indexlr::minimize.  Semantics reproduced bit for bit:
 *   - only k-mers made of ACGT are hashed; the sliding window runs over the ORDINAL of valid
 *     k-mers (an N run does not consume window slots);
 *   - the window minimum is taken on h0 = fwd + rev; ties -> the rightmost k-mer;
 *   - a minimizer is emitted when the window minimum moves to a new k-mer (the sequence of
 *     rightmost minima is monotone, so "position > last emitted" == "argmin changed");
 *   - h0 == 2^64-1 is never emitted.
 *
 * Decomposition (DESIGN.md "sketch"):
 *   seq_meta_kernel   per sequence: valid-k-mer ordinals of its ACGT runs, window count, strips
 *   sketch_mask_kernel  one workgroup = one STRIP of NT*C consecutive valid k-mers of one
 *                     sequence.  Lane L owns C consecutive k-mers ("block" L): rolls their hashes
 *                     in registers, stages them in LDS (transposed [t][L], conflict-free), then
 *                     evaluates every window that starts in its block as
 *                         argmin( suffix of own block | whole blocks between | prefix of a later block )
 *                     and sets one bit per emitted minimizer in a global bitmask (1 bit / base).
 *   emit_kernel       bitmask -> 16-byte records {h1, pos, strand|seq}: ranks the set bits
 *                     (scan), rehashes only the emitted k-mers (2/(w+1) of all) and applies the
 *                     64x64 multiply of the second hash there.
 */
#pragma once
#include "dev_common.h"
#include "index_common.h"

struct SeqTables {
    const uint32_t *packed;        /* 2-bit bases, 16 per word, NTL_LEAD_PAD bases of front padding */
    const uint64_t *seq_base;      /* [nseq+1] global base index of each sequence start */
    const uint32_t *seq_run_first; /* [nseq+1] first ACGT run of each sequence */
    const uint32_t *run_start;     /* [nruns] start of the run inside its sequence */
    const uint32_t *run_len;       /* [nruns] bases */
    uint32_t nseq;
};

/* k-dependent tables, produced by seq_meta_kernel */
struct KTables {
    uint32_t *run_n;       /* [nruns] valid k-mers of the run = max(0, len-k+1) */
    uint32_t *run_ord;     /* [nruns] ordinal of its first k-mer within the sequence */
    uint32_t *seq_M;       /* [nseq] valid k-mers of the sequence */
    uint32_t *seq_nstrips; /* [nseq] */
};

struct SketchGeom {
    int k, w;
    int a;   /* (w - C) / C : whole blocks always covered by a window besides its own */
    int r0;  /* (w - C) % C */
    int LW;  /* lanes that own windows (the others only supply data) */
    int NWO; /* windows owned per strip = LW*C - 1 */
};

__global__ void seq_meta_kernel(SeqTables T, KTables K, int k, int w, int NWO)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= T.nseq) return;
    uint32_t ord = 0;
    for (uint32_t g = T.seq_run_first[s]; g < T.seq_run_first[s + 1]; g++) {
        uint32_t len = T.run_len[g];
        uint32_t n = len >= (uint32_t)k ? len - (uint32_t)k + 1u : 0u;
        K.run_n[g] = n;
        K.run_ord[g] = ord;
        ord += n;
    }
    K.seq_M[s] = ord;
    uint32_t nwin = ord >= (uint32_t)w ? ord - (uint32_t)w + 1u : 0u;
    K.seq_nstrips[s] = (nwin + (uint32_t)NWO - 1u) / (uint32_t)NWO;
}

struct StripInfo {
    uint32_t seq;
    int32_t E0;      /* ordinal of the strip's first element (may be -1) */
    uint32_t M;      /* valid k-mers of the sequence */
    uint32_t run;    /* run containing the first real element */
    int32_t multi;   /* strip spans more than one ACGT run */
    int64_t P0;      /* single-run strips: position (in the sequence) of element E0 */
    uint64_t base;   /* seq_base */
};

/*
 * One entry per strip, computed once per sketch call (one thread per sequence walks its strips), so
 * that a strip's workgroup starts from a single 40-byte load instead of a chain of dependent ones.
 * strip_first[s] + i -> strip i of sequence s, first ordinal i*NWO - 1.
 */
/* what sketch_wave_kernel needs of a strip, in one 16-byte load: the global base index of element 0, and the number of elements that
   lie in the sequence (0: not a strip of its kind -- past the last strip, or across non-ACGT runs) */
struct __attribute__((aligned(16))) StripLite {
    uint64_t g0;
    uint32_t hi;  /* bits 0..15: the elements; bit 16 (STRIP_FIRST): the first strip of its sequence -- element 0 is virtual, there is no window 0 */
    uint32_t p0;  /* position in the sequence of element 0 (2^32 - 1 for a first strip that starts at the sequence's first base) */
};
#define STRIP_FIRST 0x10000u

/*
 * Per-strip minimizer LISTS (round 5) -- what the window passes of the large windows write instead of bits in a bitmask of one
 * bit per base (which had to be written with atomics, counted, read again, expanded and cleared: 2.3 GB of traffic per 3.9-Gbases
 * launch for 0.5 GB of bits, and most of the emit kernel's instructions).  Every strip owns the windows that START at its elements
 * 1 .. NWO, and a minimizer belongs to the strip that owns the FIRST window it is the minimum of (the windows a k-mer is the
 * minimum of are consecutive, and their minima move monotonically), i.e. a strip lists
 *       { argmin of window s : 1 <= s <= NWO }  \  { argmin of window 0 }
 * in position order -- every minimizer of the sequence exactly once over its strips, in order over the strips.  A list's entries
 * are positions in the sequence (u32).  Strip s's list lives in its slot ent[s * slot ..] when it fits (cnt[s] <= slot), else in a
 * pool behind the slots, at ent[ovf_base + ent[s * slot] ..]: low-complexity sequence has up to one minimizer per base.  A pool that
 * runs out sets ctl[1]; the counts are still right, and the sketch is made again through the bitmask (sketch_finalize).
 */
struct StripLists {
    uint32_t *cnt;     /* [nstrips + 1] minimizers per strip, zeroed per sketch; NULL: the passes write the bitmask */
    uint32_t *ent;
    uint32_t slot;     /* entries per slot */
    uint32_t ovf_base; /* first entry of the pool */
    uint32_t ovf_cap;  /* its entries */
    uint32_t *ctl;     /* [0] next free pool entry, [1] != 0: the pool ran out */
};

/* where a strip's `total` entries go: an index into ent (total > slot: a piece of the pool; NTL_NONE when the pool has run out).
   One lane calls it.  (All of ent is addressed with 32 bits: the host takes the bitmask path for anything larger.) */
__device__ __forceinline__ uint32_t strip_list_place(const StripLists &Ls, uint32_t strip, uint32_t total)
{
    Ls.cnt[strip] = total;
    if (total <= Ls.slot) return strip * Ls.slot;
    const uint32_t off = atomicAdd(&Ls.ctl[0], total);
    if (off > Ls.ovf_cap || total > Ls.ovf_cap - off) { Ls.ctl[1] = 1u; return NTL_NONE; }
    Ls.ent[(uint64_t)strip * Ls.slot] = off;
    return Ls.ovf_base + off;
}

/* A workgroup's strip-local emission bitmask (bit i: element i of the strip is listed) as the strip's list; pos_of(i) = the position
   in the sequence of element i.  All NT threads call it; s_tmp holds NT words. */
template <int NT, int NBW, typename F>
__device__ __forceinline__ void strip_bits_to_list(const StripLists &Ls, uint32_t strip, const uint32_t *s_bits, uint32_t *s_tmp, F pos_of,
                                                   uint32_t drop = NTL_NONE)
{
    const int L = threadIdx.x;
    uint32_t m = L < NBW ? s_bits[L] : 0u;
    if ((drop >> 5) == (uint32_t)L) m &= ~(1u << (drop & 31u)); /* an element that is not this strip's to list */
    uint32_t total;
    uint32_t at = block_excl_scan<NT>((uint32_t)__popc(m), s_tmp, total);
    if (L == 0) s_tmp[0] = strip_list_place(Ls, strip, total);
    __syncthreads();
    const uint32_t first = s_tmp[0];
    if (first != NTL_NONE) {
        uint32_t *const dst = Ls.ent + first;
        while (m) {
            const uint32_t b = (uint32_t)__ffs(m) - 1u;
            m &= m - 1u;
            dst[at++] = pos_of(32u * (uint32_t)L + b);
        }
    }
    __syncthreads(); /* s_tmp may be written again */
}

/* what the window passes count into, zeroed here (one thread per word) instead of by fills of their own: the strips' minimizer counts
   with the lists' two control words behind them, and the head of the fast pass's buffer (its two counters and sketch_wave_kernel's
   chunk counters) */
struct StripZero {
    uint32_t *cnt; uint32_t ncnt;
    uint32_t *head; uint32_t nhead;
};

__global__ void strip_table_kernel(SeqTables T, const uint32_t *run_n, const uint32_t *run_ord, const uint32_t *seq_M,
                                   const uint32_t *strip_first, int NWO, int strip_elems, StripInfo *tab, uint32_t cap, StripLite *lite,
                                   StripZero Z)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; /* one thread per strip */
    for (uint32_t j = i; j < Z.ncnt; j += gridDim.x * blockDim.x) Z.cnt[j] = 0u;
    if (i < Z.nhead) Z.head[i] = 0u;
    if (i >= cap) return;
    if (i >= strip_first[T.nseq]) { /* the grid of the sketch kernel is an upper bound */
        tab[i].seq = NTL_NONE;
        if (lite) { lite[i].g0 = 0; lite[i].hi = 0; lite[i].p0 = 0; }
        return;
    }
    uint32_t lo = 0, hi = T.nseq; /* largest s with strip_first[s] <= i */
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (strip_first[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t s = lo;
    StripInfo I;
    I.seq = s;
    I.E0 = (int32_t)((i - strip_first[s]) * (uint32_t)NWO) - 1;
    I.M = seq_M[s];
    I.base = T.seq_base[s];
    const uint32_t e_lo = I.E0 < 0 ? 0u : (uint32_t)I.E0;
    uint32_t e_hi = (uint32_t)(I.E0 + strip_elems);
    if (e_hi > I.M) e_hi = I.M;
    /* run containing e_lo: last run with run_ord <= e_lo, then past runs without k-mers */
    const uint32_t g0 = T.seq_run_first[s], g1 = T.seq_run_first[s + 1];
    uint32_t rl = g0, rh = g1;
    while (rh - rl > 1) {
        const uint32_t mid = (rl + rh) >> 1;
        if (run_ord[mid] <= e_lo) rl = mid; else rh = mid;
    }
    while (rl > g0 && run_n[rl] == 0) rl--;
    while (rl + 1 < g1 && (run_n[rl] == 0 || e_lo >= run_ord[rl] + run_n[rl])) rl++;
    I.run = rl;
    I.multi = (e_hi > run_ord[rl] + run_n[rl]) ? 1 : 0;
    I.P0 = (int64_t)T.run_start[rl] - (int64_t)run_ord[rl] + (int64_t)I.E0;
    tab[i] = I;
    if (lite) {
        const int64_t left = (int64_t)I.M - (int64_t)I.E0;
        lite[i].g0 = (uint64_t)((int64_t)I.base + I.P0);
        lite[i].hi = I.multi ? 0u : ((uint32_t)(left < (int64_t)strip_elems ? left : (int64_t)strip_elems) | (I.E0 < 0 ? STRIP_FIRST : 0u));
        lite[i].p0 = (uint32_t)I.P0;
    }
}

struct SketchArgs {
    SeqTables T;
    const uint32_t *run_n, *run_ord, *seq_M;
    const struct StripInfo *strip_tab; /* [nstrips] everything a strip needs to start; seq = NTL_NONE past the last strip */
    uint32_t nstrips;            /* entries of strip_tab (an upper bound of the real number of strips) */
    uint32_t *mask;              /* 1 bit per global base index: k-mer starting there is a minimizer */
    SketchGeom G;
    uint64_t roll_tab[16][2];    /* [in<<2|out] = {seed[in]^srol^k(seed[out]), srol^k(seedc[in])^seedc[out]} */
    uint64_t seed_tab[4][2];     /* [c] = {seed[c], seed[3-c]} */
    const uint64_t (*g4)[2];     /* [256] four-base init table (dev_common.h hash_init) */
    const uint64_t (*g8)[2];     /* [65536] eight-base init table */
    const struct StripLite *strip_lite; /* [nstrips + 1] the same for sketch_wave_kernel */
    const uint32_t *redo_list;   /* not NULL: process exactly these strips (flagged by sketch_fast_kernel), looping */
    const uint32_t *redo_count;
    StripLists Ls;               /* Ls.cnt != NULL: the strips' minimizers go to their lists, not to the bitmask */
};

/* 16-bit strip-local index back to 32 bits (0xFFFF -> NTL_NONE) */
__device__ __forceinline__ uint32_t ntl_idx16(uint16_t v) { return v == 0xFFFFu ? NTL_NONE : (uint32_t)v; }

struct NtlTrue { static constexpr bool value = true; };
struct NtlFalse { static constexpr bool value = false; };


/*
 * MULTI = false: strips whose k-mers lie in one ACGT run (positions contiguous): register rolling.
 * MULTI = true : strips that cross non-ACGT runs: per-lane walk over the run table, positions
 *                staged in LDS.  Both instantiations are launched over all strips; each returns
 *                immediately on strips of the other kind.
 * R0 = (w - C) % C is a template parameter so that the window pass unrolls into straight-line code
 * (which remote block and which whole-block minimum a window uses is then known at compile time).
 */
template <int C, int NT, bool MULTI, int R0T>
__device__ __forceinline__ void sketch_mask_strip(const SketchArgs &A, const uint32_t strip)
{
    constexpr int NBW = (C * NT + 31) / 32; /* words of the strip-local emission bitmask */
    __shared__ uint64_t s_h[C * NT];  /* element (L,t) at [t*NT + L] */
    __shared__ uint32_t s_pos[MULTI ? C * NT : 1];
    __shared__ uint64_t s_bm_h[NT], s_pr_h[NT];
    __shared__ uint16_t s_bm_i[NT], s_pr_i[NT], s_last[NT]; /* strip-local indices < 65535; 0xFFFF = none */
    __shared__ uint32_t s_bits[NBW + 64]; /* + one dummy word per lane of a wavefront */
    __shared__ uint64_t s_roll[16][2], s_seed[4][2];

    const int L = threadIdx.x;
    const SketchGeom G = A.G;
    /* R0T >= 0: G.r0 == R0T, the host picks the instantiation (straight-line window pass).  R0T < 0: R0 is read at run time -- the
       16-k-mer form, which is the redo / multi-run / huge-window pass now, is compiled once per strip width instead of sixteen times */
    const int R0 = R0T >= 0 ? R0T : G.r0;

    if (strip >= A.nstrips) return;
    const StripInfo I = A.strip_tab[strip];
    if (I.seq == NTL_NONE) return; /* the grid is an upper bound of the number of strips */
    if (L < 16) { s_roll[L][0] = A.roll_tab[L][0]; s_roll[L][1] = A.roll_tab[L][1]; }
    if (L < 4) { s_seed[L][0] = A.seed_tab[L][0]; s_seed[L][1] = A.seed_tab[L][1]; }
    if (L < NBW) s_bits[L] = 0;
    if ((I.multi != 0) != MULTI) return;
    __syncthreads();

    const int64_t e_lane = (int64_t)I.E0 + (int64_t)L * C; /* ordinal of this lane's element t = 0 */
    uint64_t h[C];

    /* ---- phase 1: h0 of the lane's C k-mers ------------------------------------------- */
    if (!MULTI) {
#pragma unroll
        for (int t = 0; t < C; t++) h[t] = NTL_INF;
        if (e_lane < (int64_t)I.M) {
            /* element -1 of the first strip is virtual: it is hashed like a real k-mer (the bases
               in front of the sequence are padding or the previous sequence) so that rolling out
               of it is exact, then voided. */
            const uint64_t gp = (uint64_t)((int64_t)I.base + I.P0 + (int64_t)L * C);
            uint64_t fwd, rev;
            hash_init(A.T.packed, gp, G.k, A.g8, A.g4, s_seed, fwd, rev);
            h[0] = fwd + rev;
            const uint32_t so = load_bases16(A.T.packed, gp);
            const uint32_t si = load_bases16(A.T.packed, gp + (uint64_t)G.k);
#pragma unroll
            for (int t = 1; t < C; t++) {
                const uint32_t idx = (((si >> (2 * (t - 1))) & 3u) << 2) | ((so >> (2 * (t - 1))) & 3u);
                fwd = srol1(fwd) ^ s_roll[idx][0];
                rev = sror1(rev ^ s_roll[idx][1]);
                h[t] = fwd + rev;
            }
            if (e_lane < 0 || e_lane + C > (int64_t)I.M) { /* strip edges only */
#pragma unroll
                for (int t = 0; t < C; t++) {
                    const int64_t e = e_lane + t;
                    if (e < 0 || e >= (int64_t)I.M) h[t] = NTL_INF;
                }
            }
        }
    } else {
        /* walk the run table; re-initialise the hash at every run crossing */
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
            s_pos[t * NT + L] = pv;
        }
#pragma unroll
        for (int t = 0; t < C; t++) h[t] = s_h[t * NT + L];
    }

    /* ---- phase 2: suffix minima of the own block, block minimum, prefix of length R0 --- */
    uint64_t S_h[C];
    uint32_t S_i[C];
    if (e_lane >= (int64_t)I.M) {
        /* lanes past the end of the sequence (tail strips): nothing to scan; whole wavefronts skip */
        s_bm_h[L] = NTL_INF; s_bm_i[L] = 0xFFFFu;
        s_pr_h[L] = NTL_INF; s_pr_i[L] = 0xFFFFu;
    } else {
        uint64_t rh = NTL_INF;
        uint32_t ri = NTL_NONE;
#pragma unroll
        for (int j = C - 1; j >= 0; j--) {
            if (h[j] < rh) { rh = h[j]; ri = (uint32_t)(L * C + j); } /* ties keep the right one */
            S_h[j] = rh;
            S_i[j] = ri;
        }
        s_bm_h[L] = rh;
        s_bm_i[L] = (uint16_t)ri;
        uint64_t ph = NTL_INF;
        uint32_t pi = NTL_NONE;
#pragma unroll
        for (int t = 0; t < C; t++) {
            if (t < R0 && h[t] <= ph) { ph = h[t]; pi = (uint32_t)(L * C + t); }
        }
        s_pr_h[L] = ph;
        s_pr_i[L] = (uint16_t)pi;
        if (!MULTI) {
#pragma unroll
            for (int t = 0; t < C; t++) s_h[t * NT + L] = h[t];
        }
    }
    __syncthreads();

    /* ---- phase 3: minimum over the whole blocks L+1..L+a -------------------------------- */
    /* lanes that start at least one window lying inside the sequence; in tail strips whole wavefronts
       have none and skip phases 3 and 4 */
    const bool own = L < G.LW && e_lane + G.w <= (int64_t)I.M;
    uint64_t fa_h = NTL_INF;
    uint32_t fa_i = NTL_NONE;
    if (own) {
        for (int d = 1; d <= G.a; d++) {
            const uint64_t hh = s_bm_h[L + d];
            if (hh <= fa_h) { fa_h = hh; fa_i = ntl_idx16(s_bm_i[L + d]); }
        }
    }

    /* ---- phase 4: every window starting in the own block -------------------------------- */
    uint32_t prev_i = NTL_NONE, A0_i = NTL_NONE;
    bool A0_fin = false;
    /* emitted minimizers: single-run strips set bits in the strip-local LDS bitmask without a branch
       (OR of 0 when nothing is emitted); multi-run strips go straight to the global bitmask */
    const bool lists = A.Ls.cnt != nullptr;
    auto emit = [&](uint32_t idx, bool flag) {
        if (!MULTI || lists) { /* (a multi-run strip's list is made from the strip-local bits too) */
            /* lanes with nothing to emit set a bit in a private dummy word behind the bitmask (harmless, never
               read): no branch, no same-address serialisation; a flag implies a real index */
            const uint32_t ii = flag ? idx : (uint32_t)((NBW + (L & 63)) * 32);
            atomicOr(&s_bits[ii >> 5], 1u << (ii & 31u));
        } else if (flag) {
            const uint64_t g = I.base + s_pos[(idx % C) * NT + (idx / C)];
            atomicOr(&A.mask[g >> 5], 1u << ((uint32_t)g & 31u));
        }
    };
    /* CHECK = false: every window of every owning lane of this wavefront lies inside the sequence.
       INFCHK = false: every owning lane of this wavefront has a finite minimum over its whole blocks, so no
       window minimum can be the never-emitted value 2^64-1 and the selected hash need not be tracked. */
    auto window_pass = [&](auto chk, auto infchk) {
        constexpr bool CHECK = decltype(chk)::value;
        constexpr bool INFCHK = decltype(infchk)::value;
        /* Everything right of the own block is one stream: the a whole blocks, then the elements of
           blocks Lr and Lr+1 in order.  Window j sees the stream up to element R0+j-1 of those two
           blocks, so a single running minimum, seeded with the whole-block minimum and the first R0
           elements (phase 2), serves all C windows. */
        const int Lr = L + G.a + 1;
        uint64_t P_h = fa_h;
        uint32_t P_i = fa_i;
        {
            const uint64_t hh = s_pr_h[Lr];
            if (hh <= P_h) { P_h = hh; P_i = ntl_idx16(s_pr_i[Lr]); }
        }
#pragma unroll
        for (int j = 0; j < C; j++) {
            const int rt = R0 + j; /* compile-time after unrolling */
            if (j > 0) {
                const int tp = rt - 1 < C ? rt - 1 : rt - 1 - C;
                const int Lb = rt - 1 < C ? Lr : Lr + 1;
                const uint64_t hh = s_h[tp * NT + Lb];
                if (hh <= P_h) { P_h = hh; P_i = (uint32_t)(Lb * C + tp); }
            }
            const bool right = P_h <= S_h[j];
            const uint32_t x_i = right ? P_i : S_i[j];
            bool finite = true;
            if (INFCHK) finite = (right ? P_h : S_h[j]) != NTL_INF;
            bool valid = true;
            if (CHECK) {
                const int64_t s = e_lane + j; /* window = ordinals [s, s+w) */
                valid = s >= 0 && s + G.w <= (int64_t)I.M;
            }
            const uint32_t a_i = valid ? x_i : NTL_NONE;
            if (j == 0) { A0_i = a_i; A0_fin = finite; }
            else emit(a_i, valid && a_i != prev_i && finite);
            prev_i = a_i;
        }
    };
    {
        const bool inside = e_lane >= 0 && e_lane + (C - 1) + G.w <= (int64_t)I.M;
        const bool all_inside = __ballot(own && !inside) == 0ull;
        const bool all_finite = __ballot(own && fa_h == NTL_INF) == 0ull;
        if (own) {
            if (all_inside && all_finite) window_pass(NtlFalse(), NtlFalse());
            else if (all_inside) window_pass(NtlFalse(), NtlTrue());
            else window_pass(NtlTrue(), NtlTrue());
        }
    }
    s_last[L] = (uint16_t)prev_i;
    __syncthreads();
    /* window (L,0) compares with the last window of lane L-1; (0,0) belongs to the previous strip */
    {
        const bool f0 = own && L > 0 && A0_i != NTL_NONE && A0_i != ntl_idx16(s_last[L - 1]) && A0_fin;
        if (!MULTI) emit(A0_i, f0);
        else if (f0) emit(A0_i, true);
    }
    if (lists) {
        /* the windows above are the strip's OWN windows 1 .. NWO, and an element is emitted where the minimum moves to it: each
           minimizer by the strip that owns the first of its windows (StripLists) */
        __syncthreads();
        strip_bits_to_list<NT, NBW>(A.Ls, strip, s_bits, (uint32_t *)s_bm_h, [&](uint32_t i) -> uint32_t {
            return MULTI ? s_pos[(i % C) * NT + (i / C)] : (uint32_t)(I.P0 + (int64_t)i);
        });
    } else if (!MULTI) {
        __syncthreads();
        /* flush the strip-local bitmask: strip element 32*L.. starts at global bit g0 */
        if (L < NBW) {
            const uint32_t word = s_bits[L];
            if (word) {
                const uint64_t g0 = (uint64_t)((int64_t)I.base + I.P0 + 32 * (int64_t)L);
                const uint32_t sh = (uint32_t)g0 & 31u;
                atomicOr(&A.mask[g0 >> 5], word << sh);
                if (sh && (word >> (32u - sh))) atomicOr(&A.mask[(g0 >> 5) + 1], word >> (32u - sh));
            }
        }
    }
}

template <int C, int NT, bool MULTI, int R0T>
__global__ __launch_bounds__(NT) void sketch_mask_kernel(SketchArgs A)
{
    if (A.redo_list) { /* the exact pass over the strips sketch_fast_kernel could not decide */
        const uint32_t n = *A.redo_count;
        for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
            sketch_mask_strip<C, NT, MULTI, R0T>(A, A.redo_list[i]);
            __syncthreads(); /* the strip's LDS arrays are reused */
        }
        return;
    }
    /* Workgroups are handed to the eight XCDs round-robin, and each XCD has its own L2: consecutive strips
       (which share their halo bases and their strip-table lines) go to one XCD, not to eight. */
    const uint32_t per_xcd = gridDim.x >> 3; /* the grid is a multiple of 8 */
    sketch_mask_strip<C, NT, MULTI, R0T>(A, (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3));
}

/* ---------------------------------------------------------------------------- emit -------- */

struct MxRecord {
    uint64_t hash; /* h1: the value indexlr prints and ntlink_pair looks up */
    uint32_t pos;  /* k-mer start within its sequence */
    uint32_t meta; /* bit 0: strand (1 = '+'), bits 1..31: sequence index */
};

#define EMIT_NT 256
#define EMIT_WPT 8                       /* mask words per thread (two 16-byte loads) */
#define EMIT_TILE (EMIT_NT * EMIT_WPT)   /* mask words per workgroup */
#define EMIT_CAP 2048                    /* positions staged in LDS per round (16-bit offsets inside the tile) */

/* pass 1 of the rank scan: set bits per tile */
__global__ __launch_bounds__(EMIT_NT) void mask_count_kernel(const uint32_t *mask, uint64_t nwords,
                                                             uint32_t *tile_cnt, uint32_t *tile_next)
{
    __shared__ uint32_t s_tmp[EMIT_NT];
    if (tile_next && blockIdx.x == 0 && threadIdx.x < 8) tile_next[16 * threadIdx.x] = 0u; /* the emit kernel's tile counters (EmitArgs) */
    const uint64_t w0 = (uint64_t)blockIdx.x * EMIT_TILE + (uint64_t)threadIdx.x * EMIT_WPT;
    uint32_t c = 0;
    if (w0 + EMIT_WPT <= nwords) { /* the thread's eight words as two 16-byte loads (the array is 16-byte aligned, w0 a multiple of 8) */
        const uint4 a = *reinterpret_cast<const uint4 *>(mask + w0), b = *reinterpret_cast<const uint4 *>(mask + w0 + 4);
        c = (uint32_t)(__popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w));
    } else {
        for (int i = 0; i < EMIT_WPT; i++)
            if (w0 + i < nwords) c += (uint32_t)__popc(mask[w0 + i]);
    }
    uint32_t total;
    block_excl_scan<EMIT_NT>(c, s_tmp, total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

/* tile_seq[b] = the sequence the first position of emit tile b lies in (largest s with seq_base[s] <= b * 32 * EMIT_TILE; 0
 * in front of the first sequence), for b = 0 .. ntiles: one thread per SEQUENCE writes the few tiles that start inside it, so that
 * no emit workgroup has to search the sequence table (three rounds of dependent loads and barriers per workgroup before). */
__global__ void tile_seq_kernel(const uint64_t *__restrict__ seq_base, uint32_t nseq, uint64_t ntiles, uint32_t *__restrict__ tile_seq)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseq) return;
    const uint64_t span = (uint64_t)EMIT_TILE * 32;
    const uint64_t lo = s == 0 ? 0 : seq_base[s];
    uint64_t b = (lo + span - 1) / span;
    const uint64_t end = s + 1 < nseq ? (seq_base[s + 1] + span - 1) / span : ntiles + 1; /* first tile that starts at or behind the next sequence */
    for (; b < end && b <= ntiles; b++) tile_seq[b] = (uint32_t)s;
}

struct EmitArgs {
    const uint32_t *packed;
    const uint64_t *seq_base; /* [nseq+1] */
    uint32_t nseq;
    uint32_t *mask;           /* read, then cleared: this kernel is the bitmask's last reader and hands it back zero-filled */
    uint64_t nwords;
    const uint32_t *tile_off; /* exclusive scan of tile_cnt */
    const uint32_t *tile_seq; /* [ntiles + 1] sequence at the first position of each tile (tile_seq_kernel) */
    uint32_t *mx_off;         /* [nseq+1] minimizers before each sequence start = offsets of the per-sequence lists */
    MxRecord *out;
    uint32_t out_cap;         /* records `out` can hold (it is sized before the count is known) */
    int k;
    uint64_t mult;            /* 1 ^ (k * MULTISEED) */
    uint64_t seed_tab[4][2];
    const uint64_t (*g4)[2];
    const uint64_t (*g8)[2];
    /* PROBE != 0: the sketch is made for one contig index and every minimizer is looked up as it is emitted (what probe_kernel
       would do in a pass of its own over the 16-byte records): candidate i belongs to record i */
    const IndexSlot *slots;
    const uint8_t *tags;
    const IndexSpecial *special;
    int ix_bits;
    Cand *cand;
    unsigned long long *nfound;
    /* PROBE != 0: what the map kernels need of a minimizer besides its candidate -- its position in the read (here, 4 bytes) and its
       strand (bit 31 of the candidate's meta: contig ids stay below 2^29): they read 12 bytes per minimizer and never the 16-byte
       records, which a sketch made only to be mapped (ntl_sketch_run_for_map: out = NULL) does not write at all */
    uint32_t *rpos;
    /* The grid need not hold a workgroup per tile: a workgroup walks over tiles.  tile_next = NULL: tiles blockIdx.x, + gridDim.x, ...;
       else eight counters 16 words apart (zeroed by mask_count_kernel): a workgroup of residue x = blockIdx.x % 8 takes tiles
       8 t + x, t = 0, 1, ... from counter x -- beside the window stage's kernel the emit kernel runs with a bounded number of
       resident workgroups, and where those land (and how fast each one is) is not known in advance */
    uint32_t ntiles;
    uint32_t *tile_next;
};

#define EMIT_SEQ_CAP 512 /* sequence starts of one tile cached in LDS */

/* largest s in [lo, hi) with base[s] <= gp */
__device__ __forceinline__ uint32_t seq_of(const uint64_t *base, uint32_t lo, uint32_t hi, uint64_t gp)
{
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (base[mid] <= gp) lo = mid; else hi = mid;
    }
    return lo;
}

/* the same on a table in LDS (dev_intrin.h, ntl_lds_cu64) */
__device__ __forceinline__ uint32_t seq_of_lds(ntl_lds_cu64 *base, uint32_t lo, uint32_t hi, uint64_t gp)
{
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (base[mid] <= gp) lo = mid; else hi = mid;
    }
    return lo;
}

/* PROBE: 0 = records only, 1 = + index lookup through the slot tags, 2 = + index lookup on the slots directly (index_common.h) */
/* CAP: positions staged per round.  A dense sketch (the small windows: up to a third of a tile's 65536 positions) is emitted in
   rounds of 8192 -- between two rounds stand two barriers, and with 2048 per round a tile at w = 5 took eleven: emit 0.68 -> 0.53 ms
   per 200 Mbp (profiles/HISTORY.md); everything else fits one round of EMIT_CAP and keeps the LDS for more resident workgroups */
template <int PROBE, int U = 1, int CAP = EMIT_CAP>
__global__ __launch_bounds__(EMIT_NT) void emit_kernel(EmitArgs A)
{
    unsigned long long found = 0;
    __shared__ uint32_t s_tmp[EMIT_NT];
    __shared__ uint16_t s_list[CAP];
    __shared__ uint16_t s_wrank[EMIT_TILE]; /* set bits of the tile before each of its words */
    __shared__ uint64_t s_seed[4][2];
    __shared__ uint64_t s_base[EMIT_SEQ_CAP];
    ntl_lds_cu64 *const base_lds = NTL_LDS_CU64(s_base); /* every read of s_base goes through it */
    __shared__ uint32_t s_range[2];
    __shared__ uint64_t s_g4[256][2]; /* four-base init table: LDS copy (EMIT_NT == 256 entries) */
    __shared__ uint64_t s_g4r[256][2]; /* ... and the same rotated by four bases: two groups per rotation (hash_init_g4p) */
    __shared__ uint32_t s_tile;
    const int t = threadIdx.x;
    if (t < 4) { s_seed[t][0] = A.seed_tab[t][0]; s_seed[t][1] = A.seed_tab[t][1]; }
    {
        const uint64_t gf = A.g4[t][0], gu = A.g4[t][1];
        s_g4[t][0] = gf; s_g4[t][1] = gu;
        s_g4r[t][0] = srot_h(gf, 4, 4); s_g4r[t][1] = srot_h(gu, 29, 27);
    }
    uint32_t tile = blockIdx.x;
    for (bool first_tile = true;; first_tile = false) {
    if (A.tile_next) {
        if (t == 0) s_tile = 8u * atomicAdd(&A.tile_next[16u * (blockIdx.x & 7u)], 1u) + (blockIdx.x & 7u);
        __syncthreads(); /* (its next write lies behind the barriers of the tile's work) */
        tile = s_tile;
    } else if (!first_tile) tile += gridDim.x;
    tile = ntl_readfirstlane(tile); /* uniform, and known to be: everything derived from it stays in scalar registers */
    if (tile >= A.ntiles) break;
    const uint64_t tile_w0 = (uint64_t)tile * EMIT_TILE;
    /* Sequences that overlap this tile of 65536 base positions: from the one its first position lies in to the one the next
       tile's first position lies in (tile_seq_kernel); their starts are cached in LDS.  More than EMIT_SEQ_CAP of them (tiny
       sequences) falls back to binary searches in global memory. */
    const uint32_t s_lo = A.nseq ? A.tile_seq[tile] : 0u;
    const uint32_t s_hi = A.nseq ? A.tile_seq[tile + 1] + 1u : 0u; /* candidates [s_lo, s_hi) */
    const uint64_t gp_last = tile_w0 * 32 + (uint64_t)EMIT_TILE * 32 - 1;
    const bool cached = s_hi - s_lo <= EMIT_SEQ_CAP;
    const uint32_t ncache = cached ? s_hi - s_lo : EMIT_SEQ_CAP;
    for (uint32_t i = t; i < ncache; i += EMIT_NT) s_base[i] = A.seq_base[s_lo + i];
    /* (the barriers of the scan below stand between these stores and their readers) */
    const uint64_t w0 = tile_w0 + (uint64_t)t * EMIT_WPT;
    uint32_t words[EMIT_WPT];
    uint32_t c = 0;
    if (w0 + EMIT_WPT <= A.nwords) {
        const uint4 a = *reinterpret_cast<const uint4 *>(A.mask + w0), b = *reinterpret_cast<const uint4 *>(A.mask + w0 + 4);
        words[0] = a.x; words[1] = a.y; words[2] = a.z; words[3] = a.w; words[4] = b.x; words[5] = b.y; words[6] = b.z; words[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < EMIT_WPT; i++) words[i] = w0 + i < A.nwords ? A.mask[w0 + i] : 0u;
    }
#pragma unroll
    for (int i = 0; i < EMIT_WPT; i++) c += (uint32_t)__popc(words[i]);
    uint32_t total;
    const uint32_t excl = block_excl_scan<EMIT_NT>(c, s_tmp, total);
    const uint32_t tile_base = A.tile_off[tile];
    NTL_PRIO_LATENCY_BOUND(); /* (behind the scan: in front of the shared arrays' first use it trips the compiler's address-space lowering) */
    {
        uint32_t r = excl;
#pragma unroll
        for (int i = 0; i < EMIT_WPT; i++) {
            s_wrank[t * EMIT_WPT + i] = (uint16_t)r; /* < 65536: at most 65504 bits precede a word of the tile */
            r += (uint32_t)__popc(words[i]);
        }
    }
    __syncthreads();
    /* offsets of the per-sequence lists: the rank of every sequence start that lies in this tile (the end of the last
       sequence, seq_base[nseq], counts as one: it receives the total) */
    {
        const uint64_t gp0 = tile_w0 * 32;
        for (uint32_t s = s_lo + (uint32_t)t; s <= A.nseq; s += EMIT_NT) {
            const uint64_t g = s - s_lo < ncache ? base_lds[s - s_lo] : A.seq_base[s];
            if (g > gp_last) break; /* starts are sorted: nothing further for this thread */
            if (g < gp0) continue;  /* s_lo itself may start before the tile */
            const uint32_t wl = (uint32_t)((g >> 5) - tile_w0), b = (uint32_t)g & 31u;
            const uint32_t word = tile_w0 + wl < A.nwords ? A.mask[tile_w0 + wl] : 0u;
            A.mx_off[s] = tile_base + s_wrank[wl] + (uint32_t)__popc(word & ((1u << b) - 1u));
        }
    }
    for (uint32_t r0 = 0; r0 < total; r0 += CAP) {
        uint32_t r = excl;
#pragma unroll
        for (int i = 0; i < EMIT_WPT; i++) {
            uint32_t m = words[i];
            const uint32_t pc = (uint32_t)__popc(m);
            /* a word whose ranks all lie in another round is skipped whole: a dense sketch (w = 5: a third of all positions, eleven
               rounds per tile) otherwise walks all of a thread's bits in every round */
            if (r + pc <= r0 || r >= r0 + CAP) { r += pc; continue; }
            while (m) {
                const int b = __ffs(m) - 1;
                m &= m - 1;
                if (r >= r0 && r < r0 + CAP) s_list[r - r0] = (uint16_t)((t * EMIT_WPT + i) * 32 + b);
                r++;
            }
        }
        __syncthreads();
        const uint32_t n = total - r0 < CAP ? total - r0 : CAP;
        /* U minimizers per thread and step, in straight-line code: their base-word loads, and then their first index loads, are
           in flight together (the kernel's time is the latency of these dependent random accesses, not its arithmetic) */
        for (uint32_t i0 = t; i0 < n; i0 += U * EMIT_NT) {
            uint64_t tt[U];
            MxRecord R[U];
            bool ok[U];
            uint64_t gpv[U], fwd[U], rev[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + (uint32_t)u * EMIT_NT;
                const uint32_t ic = i < n ? i : i0; /* past the end: the first one again, result dropped */
                gpv[u] = tile_w0 * 32 + s_list[ic];
                ok[u] = i < n && tile_base + r0 + i < A.out_cap;
            }
            /* the base words of all U k-mers are asked for before any of them is hashed (hash_init_g4p_multi): a dense sketch -- the
               small windows, a minimizer every three bases at w = 5 -- has nothing but these loads' latency between its barriers */
            hash_init_g4p_multi<U>(A.packed, gpv, A.k, s_g4, s_g4r, s_seed, fwd, rev);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint64_t gp = gpv[u];
                const uint32_t sq = cached ? s_lo + seq_of_lds(base_lds, 0, s_hi - s_lo, gp) : seq_of(A.seq_base, s_lo, s_hi, gp);
                const uint64_t sb = cached ? base_lds[sq - s_lo] : A.seq_base[sq];
                uint64_t h = (fwd[u] + rev[u]) * A.mult;
                h ^= h >> 27;
                tt[u] = h;
                R[u].hash = h;
                R[u].pos = (uint32_t)(gp - sb);
                R[u].meta = (sq << 1) | (fwd[u] <= rev[u] ? 1u : 0u);
            }
            if (PROBE) {
                IndexProbe<PROBE == 1> pr[U];
#pragma unroll
                for (int u = 0; u < U; u++) pr[u].start(tt[u], A.slots, A.tags, A.ix_bits);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!ok[u]) continue;
                    const uint32_t at = tile_base + r0 + i0 + (uint32_t)u * EMIT_NT;
                    if (A.out) A.out[at] = R[u];
                    Cand cd = pr[u].finish(tt[u], A.slots, A.tags, A.special, ((uint64_t)1 << A.ix_bits) - 1);
                    A.rpos[at] = R[u].pos;
                    cd.meta |= R[u].meta << 31;
                    A.cand[at] = cd;
                    found += cd.meta & 1u;
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (ok[u]) A.out[tile_base + r0 + i0 + (uint32_t)u * EMIT_NT] = R[u];
            }
        }
        __syncthreads();
    }
    /* every read of this tile's mask words is behind a barrier by now (the offsets loop reads other threads' words): clear
       what was set, so that the next window pass finds the bitmask zero-filled without a fill pass of its own */
    __syncthreads();
    if (c) {
        if (w0 + EMIT_WPT <= A.nwords) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(A.mask + w0) = z;
            *reinterpret_cast<uint4 *>(A.mask + w0 + 4) = z;
        } else {
#pragma unroll
            for (int i = 0; i < EMIT_WPT; i++)
                if (words[i]) A.mask[w0 + i] = 0u;
        }
    }
    } /* tiles */
    if (PROBE) { /* one atomic per workgroup */
        __syncthreads();
        if (t == 0) s_range[0] = 0;
        __syncthreads();
        if (found) atomicAdd(&s_range[0], (uint32_t)found);
        __syncthreads();
        if (t == 0 && s_range[0]) atomicAdd(A.nfound, (unsigned long long)s_range[0]);
    }
}

/* ---------------------------------------------------------------------------- emit from the strips' lists -------- */

#define EL_NT 256
#define EL_STRIPS 128 /* strips per tile: a thread (of the first EL_STRIPS) per strip sets the tile up */
#define EL_CAP 2048   /* minimizers per round */

struct EmitListArgs {
    StripLists Ls;
    const StripInfo *strip_tab;
    const uint32_t *strip_off; /* [nstrips + 1] exclusive scan of Ls.cnt: the rank of a strip's first minimizer */
    uint32_t nstrips;
};

/* behind the scan of the strips' counts: a sketch whose lists ran out of pool carries a total that no array holds (the map kernels
   leave it alone, map_sketch_overflowed) and the flag that tells sketch_finalize why; also zeroes the emit kernel's tile counters */
__global__ void list_fail_kernel(const uint32_t *__restrict__ ctl, uint32_t *__restrict__ total, uint32_t *__restrict__ flag, uint32_t *__restrict__ tile_next)
{
    if (threadIdx.x < 8) tile_next[16 * threadIdx.x] = 0u;
    if (threadIdx.x == 0) {
        const uint32_t f = ctl[1];
        *flag = f;
        if (f) *total = 0xFFFFFFFFu;
    }
}

/* offsets of the per-sequence lists: the rank of the first minimizer of a sequence's first strip (a sequence without strips: of the
   next strip; mx_off[nseq] = the total) */
__global__ void mx_off_from_strips_kernel(const uint32_t *__restrict__ strip_first, const uint32_t *__restrict__ strip_off, uint32_t nseq,
                                          uint32_t *__restrict__ mx_off)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= nseq) mx_off[s] = strip_off[strip_first[s]];
}

/*
 * emit_list_kernel: what emit_kernel does, from the per-strip lists instead of the bitmask (round 5).  No bits to expand and to
 * rank, no sequence to search: a minimizer's rank is its strip's offset (a scan over one count per strip, not over the bitmask)
 * plus its place in the list, its sequence is the strip's.  A workgroup takes a tile of EL_STRIPS consecutive strips; the strips'
 * threads spread their strip's number over the tile's ranks (one byte per minimizer in LDS), then a lane per minimizer: list entry
 * (lanes of one strip: consecutive words), the k-mer's hash from its bases, the record, the index lookup.
 */
template <int PROBE, int U = 1>
__global__ __launch_bounds__(EL_NT) NTL_MAIN_STREAM_SGPRS void emit_list_kernel(EmitArgs A, EmitListArgs Q)
{
    unsigned long long found = 0;
    __shared__ uint8_t s_sid[EL_CAP];
    __shared__ uint32_t s_seq[EL_STRIPS], s_first[EL_STRIPS], s_loc[EL_STRIPS];
    __shared__ uint64_t s_sbase[EL_STRIPS];
    __shared__ uint64_t s_seed[4][2];
    __shared__ uint64_t s_g4[256][2];  /* four-base init table: LDS copy (EL_NT == 256 entries) */
    __shared__ uint64_t s_g4r[256][2]; /* ... and the same rotated by four bases: two groups per rotation (hash_init_g4p) */
    __shared__ uint32_t s_tile, s_found;
    static_assert(EL_NT == 256 && EL_STRIPS <= 256, "one table entry per thread; strip numbers are bytes");
    const int t = threadIdx.x;
    if (t < 4) { s_seed[t][0] = A.seed_tab[t][0]; s_seed[t][1] = A.seed_tab[t][1]; }
    if (t == 0) s_found = 0;
    {
        const uint64_t gf = A.g4[t][0], gu = A.g4[t][1];
        s_g4[t][0] = gf; s_g4[t][1] = gu;
        s_g4r[t][0] = srot_h(gf, 4, 4); s_g4r[t][1] = srot_h(gu, 29, 27);
    }
    if (Q.Ls.ctl[1]) return; /* the lists ran out of pool: nothing here is complete (list_fail_kernel) */
    uint32_t tile = blockIdx.x;
    for (bool first_tile = true;; first_tile = false) {
        if (A.tile_next) { /* a bounded number of resident workgroups that take their tiles from counters (EmitArgs) */
            __syncthreads();
            if (t == 0) s_tile = 8u * atomicAdd(&A.tile_next[16u * (blockIdx.x & 7u)], 1u) + (blockIdx.x & 7u);
            __syncthreads();
            tile = s_tile;
        } else if (!first_tile) tile += gridDim.x;
        tile = ntl_readfirstlane(tile);
        if (tile >= A.ntiles) break;
        const uint32_t strip0 = tile * EL_STRIPS;
        const uint32_t strip1 = strip0 + EL_STRIPS < Q.nstrips ? strip0 + EL_STRIPS : Q.nstrips;
        const uint32_t tile_base = Q.strip_off[strip0], total = Q.strip_off[strip1] - tile_base;
        if (!first_tile && !A.tile_next) __syncthreads(); /* the previous tile's last readers of the strip tables */
        uint32_t my_cnt = 0, my_loc = 0;
        if ((uint32_t)t < strip1 - strip0) {
            const uint32_t strip = strip0 + (uint32_t)t;
            my_cnt = Q.Ls.cnt[strip];
            my_loc = Q.strip_off[strip] - tile_base;
            s_loc[t] = my_loc;
            if (my_cnt) {
                s_seq[t] = Q.strip_tab[strip].seq;
                s_sbase[t] = Q.strip_tab[strip].base;
                s_first[t] = my_cnt <= Q.Ls.slot ? strip * Q.Ls.slot : Q.Ls.ovf_base + Q.Ls.ent[(uint64_t)strip * Q.Ls.slot];
            }
        }
        NTL_PRIO_LATENCY_BOUND();
        for (uint32_t r0 = 0; r0 < total; r0 += EL_CAP) {
            {   /* the strip of every rank of this round */
                const uint32_t lo = my_loc > r0 ? my_loc : r0;
                const uint32_t hi = my_loc + my_cnt < r0 + EL_CAP ? my_loc + my_cnt : r0 + EL_CAP;
                for (uint32_t j = lo; j < hi; j++) s_sid[j - r0] = (uint8_t)t;
            }
            __syncthreads();
            const uint32_t n = total - r0 < EL_CAP ? total - r0 : EL_CAP;
            /* U minimizers per thread and step, in straight-line code: their list entries, then their base words, then their first index
               loads are in flight together -- the kernel's time is the latency of these dependent random accesses, and beside the
               window stage it keeps to two workgroups per CU: what it cannot have in wavefronts it has in loads per wavefront */
            for (uint32_t i0 = t; i0 < n; i0 += U * EL_NT) {
                uint32_t p[U], sq[U], sidv[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t i = i0 + (uint32_t)u * EL_NT;
                    const uint32_t ic = i < n ? i : i0; /* past the end: the first one again, result dropped */
                    sidv[u] = s_sid[ic];
                    p[u] = Q.Ls.ent[s_first[sidv[u]] + (r0 + ic - s_loc[sidv[u]])];
                    ok[u] = i < n && tile_base + r0 + i < A.out_cap;
                }
                uint64_t h[U], gpv[U], fwd[U], rev[U];
                MxRecord R[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    sq[u] = s_seq[sidv[u]];
                    gpv[u] = s_sbase[sidv[u]] + p[u];
                }
                hash_init_g4p_multi<U>(A.packed, gpv, A.k, s_g4, s_g4r, s_seed, fwd, rev);
#pragma unroll
                for (int u = 0; u < U; u++) {
                    uint64_t hh = (fwd[u] + rev[u]) * A.mult;
                    hh ^= hh >> 27;
                    h[u] = hh;
                    R[u].hash = hh;
                    R[u].pos = p[u];
                    R[u].meta = (sq[u] << 1) | (fwd[u] <= rev[u] ? 1u : 0u);
                }
                if (PROBE) {
                    IndexProbe<PROBE == 1> pr[U];
#pragma unroll
                    for (int u = 0; u < U; u++) pr[u].start(h[u], A.slots, A.tags, A.ix_bits);
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (!ok[u]) continue;
                        const uint32_t at = tile_base + r0 + i0 + (uint32_t)u * EL_NT;
                        if (A.out) A.out[at] = R[u];
                        Cand cd = pr[u].finish(h[u], A.slots, A.tags, A.special, ((uint64_t)1 << A.ix_bits) - 1);
                        A.rpos[at] = p[u];
                        cd.meta |= R[u].meta << 31;
                        A.cand[at] = cd;
                        found += cd.meta & 1u;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; u++)
                        if (ok[u]) A.out[tile_base + r0 + i0 + (uint32_t)u * EL_NT] = R[u];
                }
            }
            __syncthreads();
        }
    } /* tiles */
    if (PROBE) { /* one atomic per workgroup */
        __syncthreads();
        if (found) atomicAdd(&s_found, (uint32_t)found);
        __syncthreads();
        if (t == 0 && s_found) atomicAdd(A.nfound, (unsigned long long)s_found);
    }
}
