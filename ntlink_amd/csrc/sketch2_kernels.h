/*
 * sketch_fast_kernel: the window pass of the (k,w) minimizer sketch on 32-bit keys made from the hashes' 31-bit rings.
 *
 * Same contract and strip geometry as sketch_mask_kernel (sketch_kernels.h; btllib `indexlr`, ntLink:199,223,
 * SURVEY.md 8 rows a1-a2): one workgroup = one strip of NT*16 consecutive valid-k-mer ordinals of one
 * sequence, lane L owns 16 of them, one bit per emitted minimizer in the global bitmask.  What differs is the
 * arithmetic between the hash and the bit:
 *
 *   key      ntHash's split rotation never mixes bits 33..63 of a hash (a 31-bit ring) with bits 0..32, so the rolling update
 *            of that ring alone is closed: rotate, XOR the ring part of the seeds.  The window minimum is taken on
 *            key = 2 * ((F + R) mod 2^31) + junk bit, F and R the rings of fwd and rev: one register (v_min_u32) instead of
 *            the 64-bit h0 plus an index, and 8 instructions per k-mer instead of 16 for the roll.  With c = h0 >> 33:
 *            c = (F + R + carry of the low 33 bits) mod 2^31, so  key >> 1 <= c <= (key >> 1) + 1  (mod 2^31), and
 *            key_i + 4 <= key_j  implies  c_i < c_j  implies  h0_i < h0_j.  Positions are NOT tracked.
 *   argmin   a window whose minimum VALUE differs from the previous window's is a "changed" window (2/(w+1) of
 *            all).  Where the value dropped, the element that just entered is the new minimum (nothing else can
 *            be below the old one); where it rose, four lanes search the window for the position of its
 *            minimum over the block minima and two blocks of elements staged in LDS.
 *   exact    the result equals Indexlr's rightmost 64-bit argmin unless two k-mers that share a window have keys within
 *            SK2_NEAR = 3 of each other while one of them is that window's minimum (identical k-mers in low-complexity
 *            sequence, or 2^-29 coincidences).  Every such case is DETECTED and the strip is handed to sketch_mask_kernel
 *            (the exact 64-bit pass) through a redo list, and none of what this pass found for it is written.
 *
 *   Why this is exact.  Invariant: the minimum key v of window s is attained by an element m and every other element of
 *   the window has a key > v + SK2_NEAR; then m is the 64-bit argmin, rightmost or not.
 *   (1) Searched windows (the first owned window of a strip, and every window whose minimum rose): the search counts the
 *       elements with key <= v + SK2_NEAR and flags the strip unless there is exactly one.
 *   (2) Every other window s compares the entering element e with the minimum v' of window s-1, for which the invariant
 *       holds.  |e - v'| <= SK2_NEAR flags the strip.  e < v' - SK2_NEAR: everything else in the window was in window s-1,
 *       so it is >= v' > e + SK2_NEAR: e is the new m.  e > v' + SK2_NEAR and the minimum unchanged: the old m is still
 *       there (had it left, the minimum would have risen past v' + SK2_NEAR) and still alone.
 *   (3) (F + R + carry) mod 2^31 could wrap -- ring sum 2^31 - 1, true c = 0: the largest key for the smallest hash -- but
 *       F + R = 2^31 - 1 means R = ~F, i.e. F ^ R has 31 set bits, and F ^ R always has an EVEN number: it is the XOR, over
 *       the k bases, of a rotation of seed[b]'s ring and a rotation of seed[complement b]'s ring, and for the ntHash seeds
 *       the rings of A and T have 19 and 15 set bits, those of C and G 12 and 12 (static_assert next to the seed table in
 *       ntl_hip.hip; tests/test_host.py checks it on the oracle's hashes).  So no real k-mer has key >= 2^32 - 2, and the
 *       padding value 2^32 - 1 is nobody's key.
 *   (4) A flagged strip writes NONE of its bits (a lane that saw a near tie may have taken the wrong side of it for a
 *       drop); the exact pass redoes the whole strip.  v == 2^32-1 (a window of padding) flags.
 *
 * The rings of a lane's first k-mer are assembled from 16-base partial hashes (rings) that neighbouring lanes compute
 * for their own 16 bases (two 8-base table lookups each) and exchange through LDS:
 *   F = XOR_i rotl^(k-16(i+1))(F16[L+i]) ^ (first k%16 bases of chunk L+k/16),   same for rev with rotr.
 */
#pragma once
#include <utility>
#include "sketch_kernels.h"

#define SK2_JOBCAP 512 /* searched windows per strip; more (pathological sequence) hands the strip to the exact pass */
#define SK2_QMAX 16     /* k <= 16 * SK2_QMAX */
#define SK2_PAD 72      /* whole blocks right of a window's first block: a + 2 <= SK2_PAD, i.e. w <= 1135; larger windows take the exact pass */
#define SK2_INF 0xFFFFFFFFu
#define SK2_NEAR 3u     /* keys closer than this + 1 do not order their k-mers (see "exact") */
/* phase ablation for tools/gpu_ablate.sh (results WRONG): only in builds with -DNTL_SKETCH_ABLATION, so that the product
   kernel carries no run-time switches inside its unrolled loops */
#ifdef NTL_SKETCH_ABLATION
#define SK2_DBG(B, bit) ((B).dbg & (bit))
#else
#define SK2_DBG(B, bit) 0
#endif

struct Sketch2Args {
    SketchArgs A;
    uint32_t *redo_list;   /* strips for the exact pass */
    uint32_t *redo_count;
    uint64_t max_word;     /* last word of `packed` that may be read */
    const uint2 *g8k;      /* k-dependent eight-base ring tables (g8k_build_kernel): [w] first half of a chunk, [65536 + w] second half */
    int q16, r16;          /* k = 16 * q16 + r16 */
    uint32_t rev_a, rev_b; /* (k - 1) % 33, (k - 1) % 31: the rotation that turns the Horner form of the reverse strand into rev */
    int force_redo;        /* tests: flag every strip */
    int dbg;               /* ablation (tools/sketch_bench.py, results WRONG): 1 no search, 2 no window pass, 4 no rolling, 8 no init */
    uint32_t thresh;       /* sketch_thresh_kernel: keys below it are candidates */
    const uint2 *g4k;      /* sketch_thresh_kernel: k-dependent four-base ring tables (g4k_build_kernel), [256 j + byte] for the j-th four bases of a chunk */
    uint32_t *fb_list;     /* sketch_thresh_kernel: the strips it gives up, for sketch_fast_list_kernel */
    uint32_t *fb_count;
    uint32_t *chunk_next;  /* sketch_wave_kernel: [16 x] = the next chunk of strips of XCD x's share that nobody has taken yet (zeroed per launch) */
    uint32_t chunk_budget; /* sketch_wave_kernel: 0 = resident wavefronts that take chunks until none is left; else chunks per wavefront (>= 2) */
};

__device__ __forceinline__ uint32_t sk2_bases16(const uint32_t *__restrict__ packed, uint64_t gp, uint64_t max_word)
{
    uint64_t wi = gp >> 4;
    if (wi >= max_word) wi = max_word - 1; /* over-reads behind the last sequence: values never used */
    const uint32_t a = (uint32_t)gp & 15u;
    return ntl_alignbit(packed[wi + 1], packed[wi], 2u * a);
}

/* a 31-bit ring held in bits 0..30 (bit 31 clear), rotated by a uniform amount 1..30 */
__device__ __forceinline__ uint32_t ring_rotl(uint32_t x, uint32_t n) { return ((x << n) & 0x7FFFFFFFu) | (x >> (31u - n)); }
__device__ __forceinline__ uint32_t ring_rotr(uint32_t x, uint32_t n) { return ring_rotl(x, 31u - n); }
__device__ __forceinline__ uint32_t ring_of(uint64_t h) { return (uint32_t)(h >> 33); }

/* The eight-base table g8 (k-independent: {XOR_j srol^(7-j)(seed[b_j]), XOR_j sror^(7-j)(seed[3-b_j])}) in the two forms a
 * 16-base chunk needs, so that a chunk's partial hash is two loads and two XORs without any rotation -- and only the 31-bit
 * rings (bits 33..63) of each, which is all the window pass's keys are made of (8 B per entry, 1 MB per k):
 *   T0[w] = {srol^8(f), R(sror^8(u))}   the first eight bases of a chunk       (entry w)
 *   T1[w] = {f,         R(u)}           its second eight bases                 (entry 65536 + w)
 * R = srol^(k-1) on the reverse strand: the rotation that turns the Horner form into rev commutes with every other
 * rotation and XOR, so it is applied to the table once per k instead of to every lane's result. */
__global__ __launch_bounds__(256) void g8k_build_kernel(const uint64_t (*__restrict__ g8)[2], uint2 *__restrict__ g8k,
                                                        uint32_t rev_a, uint32_t rev_b)
{
    const uint32_t w = blockIdx.x * 256 + threadIdx.x;
    if (w >= 65536u) return;
    const uint64_t f = g8[w][0], u = g8[w][1];
    g8k[w] = make_uint2(ring_of(srot_h(f, 8, 8)), ring_of(srot_u(srot_h(u, 25, 23), rev_a, rev_b)));
    g8k[65536u + w] = make_uint2(ring_of(f), ring_of(srot_u(u, rev_a, rev_b)));
}

/* The same partial hashes from four lookups in tables small enough for LDS (8 KB): G[j][byte] = the contribution of the j-th four
 * bases of a sixteen-base chunk, rotated into place -- {srol^(4(3-j))(f4), R(sror^(4(3-j))(u4))}, rings only -- so that
 *   T0[w0] ^ T1[w1] = G[0][b0] ^ G[1][b1] ^ G[2][b2] ^ G[3][b3]     and    T1[w0] = G[2][b0] ^ G[3][b1]   (k % 16 == 8).
 * sketch_thresh_kernel copies them into LDS at the start of a strip, beside its load of the base words: the table lookups of the
 * eight-base form are a SECOND dependent global load per strip (a 1-MB table: L2 at best), which costs the kernel 13 % of its time
 * alone and twice that next to the emit kernel's random traffic (profiles/r03y). */
__global__ __launch_bounds__(256) void g4k_build_kernel(const uint64_t (*__restrict__ g4)[2], uint2 *__restrict__ g4k,
                                                        uint32_t rev_a, uint32_t rev_b)
{
    const uint32_t b = threadIdx.x;
    const uint64_t f = g4[b][0], u = g4[b][1];
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t sh = 4u * (3u - j);
        const uint64_t fs = sh ? srot_h(f, sh, sh) : f;
        const uint64_t us = sh ? srot_h(u, 33u - sh, 31u - sh) : u;
        g4k[256u * j + b] = make_uint2(ring_of(fs), ring_of(srot_u(us, rev_a, rev_b)));
    }
}

/* partial hashes (rings) of one 16-base chunk: Horner forms over its 16 bases FU = {F, R(U)} and over its first r bases P = {PF, R(PU)} */
__device__ __forceinline__ void sk2_chunk(uint32_t so, int r, const Sketch2Args &B, uint2 &FU, uint2 &P)
{
    const uint32_t w0 = so & 0xFFFFu, w1 = so >> 16;
    const uint2 a = B.g8k[w0], b = B.g8k[65536u + w1];
    FU = make_uint2(a.x ^ b.x, a.y ^ b.y);
    P = make_uint2(0u, 0u);
    if (r == 8) P = B.g8k[65536u + w0];
    else if (r) { /* k % 16 not in {0, 8}: four-base and single-base steps on the plain 64-bit tables, then R */
        const SketchArgs &A = B.A;
        uint64_t f = 0, u = 0;
        int j = 0;
        if (r >= 8) { f = A.g8[w0][0]; u = A.g8[w0][1]; j = 8; }
        if (r - j >= 4) {
            const uint32_t byte = (so >> (2 * j)) & 255u;
            f = srot_h(f, 4, 4) ^ A.g4[byte][0];
            u = srot_h(u, 29, 27) ^ A.g4[byte][1];
            j += 4;
        }
        for (; j < r; j++) {
            const uint32_t c = (so >> (2 * j)) & 3u;
            f = srol1(f) ^ A.seed_tab[c][0];
            u = sror1(u) ^ A.seed_tab[c][1];
        }
        P = make_uint2(ring_of(f), ring_of(srot_u(u, B.rev_a, B.rev_b)));
    }
}

template <int NT, int R0, bool BIG>
__device__ __forceinline__ void sk2_fast_strip(const Sketch2Args &B, const uint32_t strip)
{
    constexpr int C = 16;
    constexpr int NBW = (C * NT + 31) / 32;
    constexpr int ST = NT;                      /* row stride of s_c: element (L,t) at [t*ST + L]; rows 1 KB apart, so two rows
                                                   of one lane are one ds_read2st64 / ds_write2st64 (the rare search phase
                                                   pays a 4-way conflict for it) */
    constexpr int NX = NT + SK2_QMAX + 1;
    /* BIG = false (a + 2 <= 16, i.e. w <= 255: the windows ntLink runs with): one range-minimum level, a short job list in the
       bytes of the rolling table (dead after phase 1), 16 INF entries -> 20.4 KB: EIGHT workgroups (32 wavefronts) per CU.
       BIG = true (w <= 1135): two levels, 72 INF entries, 512 jobs -> 23.2 KB, seven workgroups. */
    constexpr int PAD = BIG ? SK2_PAD : 16;     /* INF entries behind the block minima: a + 2 <= PAD (ntl_sketch_run checks) */
    constexpr int JOBCAP = BIG ? SK2_JOBCAP : 128;
    /* s_c doubles as the exchange area of phase 1:
       the rings {F16, U16} of chunk L at s_xy[L], {PF, PU} (first k%16 bases) at s_xy[NX + L]; a barrier separates the last
       read of the partial hashes from the first staged element */
    __shared__ uint32_t s_c[C * ST];
    __shared__ uint32_t s_bm[NT + PAD];         /* block minima; INF behind NT */
    __shared__ uint32_t s_pre0[NT + 4];         /* minimum of the first R0 elements of each block */
    __shared__ uint32_t s_bits[NBW];
    __shared__ uint32_t s_njobs, s_flag, s_q0;
    __shared__ uint32_t s_roll[64];             /* [2 * (in<<2|out)] = the 31-bit-ring parts {fwd seed >> 33, rev seed >> 32} of roll_tab */
    __shared__ uint32_t s_t0[NT + PAD];         /* range-minimum levels */
    __shared__ uint32_t s_t1[BIG ? NT + PAD : 1];
    __shared__ uint16_t s_jobs_big[BIG ? SK2_JOBCAP : 1];
    uint16_t *const s_jobs = BIG ? s_jobs_big : (uint16_t *)&s_roll[0]; /* 256 B = 128 jobs: written two barriers after the last roll */
    uint2 *const s_xy = (uint2 *)s_c;
    uint32_t *const s_so = (uint32_t *)&s_xy[2 * NX]; /* [NX + 1] the chunks' base words, behind the partial hashes */
    static_assert(sizeof(uint32_t) * C * ST >= sizeof(uint2) * 2 * NX + sizeof(uint32_t) * (NX + 1), "the exchange area must fit the element array");

    const SketchArgs &A = B.A;
    const int L = threadIdx.x;
    const SketchGeom G = A.G;
    if (strip >= A.nstrips) return;
    const StripInfo I = A.strip_tab[strip];
    if (I.seq == NTL_NONE || I.multi != 0) return; /* strips that cross non-ACGT runs: sketch_mask_kernel<.., MULTI = true> */
    /* nothing written here is read before the first barrier below */
    if (L < 16) { s_roll[2 * L] = (uint32_t)(A.roll_tab[L][0] >> 33); s_roll[2 * L + 1] = (uint32_t)(A.roll_tab[L][1] >> 32); }
    if (L < NBW) s_bits[L] = 0;
    if (L < PAD) s_bm[NT + L] = SK2_INF;
    if (L < 4) s_pre0[NT + L] = SK2_INF;
    if (L == 0) { s_njobs = 0; s_flag = B.force_redo ? 1u : 0u; s_q0 = NTL_NONE; }
    const bool lists = A.Ls.cnt != nullptr;

    const int64_t e_lane = (int64_t)I.E0 + (int64_t)L * C; /* ordinal of this lane's element t = 0 */
    const uint64_t gp = (uint64_t)((int64_t)I.base + I.P0 + (int64_t)L * C);

    /* ---- phase 1a: 16-base partial hashes of the lane's own chunk (and of the chunks behind the strip) ---- */
    const bool live = e_lane < (int64_t)I.M;            /* the lane has at least one k-mer of the sequence */
    const bool feeds = e_lane - 16 * (int64_t)(B.q16 + 1) < (int64_t)I.M; /* its chunk is part of a live lane's first k-mer */
    uint32_t so = 0;
    if (feeds && !SK2_DBG(B, 8)) {
        so = sk2_bases16(A.T.packed, gp, B.max_word);
        uint2 FU, P;
        sk2_chunk(so, B.r16, B, FU, P);
        s_xy[L] = FU;
        s_so[L] = so;
        if (B.r16) s_xy[NX + L] = P;
    }
    if (L <= B.q16 && !SK2_DBG(B, 8)) { /* chunks NT .. NT+q16 feed the last lanes */
        const int64_t ev = (int64_t)I.E0 + (int64_t)(NT + L) * C;
        if (ev - 16 * (int64_t)(B.q16 + 1) < (int64_t)I.M) {
            const uint32_t sv = sk2_bases16(A.T.packed, gp + (uint64_t)NT * C, B.max_word);
            uint2 FU, P;
            sk2_chunk(sv, B.r16, B, FU, P);
            s_xy[NT + L] = FU;
            s_so[NT + L] = sv;
            if (B.r16) s_xy[NX + NT + L] = P;
        }
    }
    __syncthreads();

    /* ---- phase 1b: first k-mer's rings from the partials, then 15 rolling steps; key = 2 * ring sum (see `key` above) ---- */
    uint32_t c[C];
#pragma unroll
    for (int t = 0; t < C; t++) c[t] = SK2_INF;
    uint32_t fx = 0, ry = 0; /* the rings of the first k-mer's fwd and rev (forms: see the rolling loop) */
    uint32_t si = 0; /* the sixteen bases behind the lane's first k-mer: the chunks q16 (and q16 + 1) further on */
    if (live) {
        {
            const uint32_t lo = s_so[L + B.q16];
            si = B.r16 ? ntl_alignbit(s_so[L + B.q16 + 1], lo, 2u * (uint32_t)B.r16) : lo;
        }
        uint32_t f = 0, u = 0;
        for (int i = 0; i < B.q16; i++) {
            if (i) { f = ring_rotl(f, 16); u = ring_rotr(u, 16); } /* srol^16, sror^16 on the rings */
            const uint2 p = s_xy[L + i];
            f ^= p.x;
            u ^= p.y;
        }
        if (B.r16) {
            const uint32_t r = (uint32_t)B.r16;
            if (B.q16) { f = ring_rotl(f, r); u = ring_rotr(u, r); } /* 1 <= r <= 15 */
            const uint2 p = s_xy[NX + L + B.q16];
            f ^= p.x;
            u ^= p.y;
        }
        fx = f;
        ry = u << 1; /* the tables carry the reverse strand's final rotation */
    }
    __syncthreads(); /* the partial hashes have been read: s_c may take the elements */
    if (live) {
        /* Only the 31-bit rings (bits 33..63) are rolled: fx holds fwd's ring in bits 0..30 (bit 31: junk), ry holds rev's in
           bits 1..31 (bit 0: junk), so that a rotation is two instructions and key = 2 * (F + R) + junk is one. */
        c[0] = (fx << 1) + ry;
        /* byte offset into s_roll of step t = b + 1: 8 * (in<<2 | out) for base b of si / so, as byte b/4 of word b%4 */
        uint32_t wz[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t o2 = r < 2 ? so << (3 - 2 * r) : so >> (2 * r - 3);
            const uint32_t i2 = r < 3 ? si << (5 - 2 * r) : si >> (2 * r - 5);
            wz[r] = (o2 & 0x18181818u) | (i2 & 0x60606060u);
        }
#pragma unroll
        for (int t = 1; t < C; t++) {
            if (SK2_DBG(B, 4)) { c[t] = c[0] + (uint32_t)t * wz[0]; continue; }
            const int b = t - 1;
            const uint32_t off = ntl_bfe(wz[b & 3], 8u * (uint32_t)(b >> 2), 8u);
            const uint2 sd = *(const uint2 *)((const char *)s_roll + off);
            fx = ((fx << 1) | ((fx >> 30) & 1u)) ^ sd.x;        /* srol1 on the ring: v_bfe + v_lshl_or */
            const uint32_t a = ry ^ sd.y;
            ry = ntl_alignbit(a >> 1, a, 1);                    /* sror1: ring bit 0 (bit 1 of a) -> bit 31 */
            c[t] = (fx << 1) + ry;                              /* v_lshl_add_u32 */
        }
        if (e_lane < 0 || e_lane + C > (int64_t)I.M) { /* strip edges only */
#pragma unroll
            for (int t = 0; t < C; t++) {
                const int64_t e = e_lane + t;
                if (e < 0 || e >= (int64_t)I.M) c[t] = SK2_INF;
            }
        }
    }

    /* ---- phase 2: stage, minimum of the first R0 elements, suffix minima in place (c[0] = block minimum) ---- */
    uint32_t pre0 = SK2_INF;
#pragma unroll
    for (int t = 0; t < C; t++) {
        s_c[t * ST + L] = c[t];
        if (t < R0) pre0 = c[t] < pre0 ? c[t] : pre0;
    }
#pragma unroll
    for (int j = C - 2; j >= 0; j--) c[j] = c[j] < c[j + 1] ? c[j] : c[j + 1]; /* c[j] = min of elements j..15 */
    s_bm[L] = c[0];
    s_pre0[L] = pre0;
    __syncthreads();

    /* ---- phase 3: minimum over the whole blocks L+1 .. L+a: range minima of 4^p blocks (one barrier per level),
       the range of a covered by at most four of them ---- */
    uint32_t fa = SK2_INF;
    if (G.a >= 1) {
        const uint32_t *cur = s_bm;
        int span = 1;
        for (int lv = 0; span * 4 <= G.a; lv++) {
            uint32_t *nxt = (lv & 1) ? s_t1 : s_t0;
            uint32_t m = cur[L];
#pragma unroll
            for (int q = 1; q < 4; q++) { const uint32_t v = cur[L + q * span]; m = v < m ? v : m; }
            nxt[L] = m;
            if (L < PAD) nxt[NT + L] = SK2_INF;
            __syncthreads();
            cur = nxt;
            span *= 4;
        }
        for (int o = 0; o < G.a; o += span) {
            const int at = o + span <= G.a ? o : G.a - span; /* the last range is pulled back inside */
            const uint32_t v = cur[L + 1 + at];
            fa = v < fa ? v : fa;
        }
    }

    /* ---- phase 4: the 17 windows starting at elements 0..16 of the own block (window 16 = window 0 of the next lane:
       this lane decides whether it changed) ---- */
    const bool own = L < G.LW && e_lane + G.w <= (int64_t)I.M && !SK2_DBG(B, 2);
    uint32_t chg = 0;    /* bit j: window j has another minimum value than window j-1 */
    uint32_t le = 0;     /* bit j: the element that entered at window j is <= the minimum of window j-1 */
    {
        const bool inside = e_lane >= 0 && e_lane + C + G.w <= (int64_t)I.M;  /* all 17 windows lie in the sequence */
        const bool all_inside = __ballot(own && !inside) == 0ull;             /* ... for every owning lane of the wavefront */
        auto window_pass = [&](auto chk) {
            constexpr bool CHECK = decltype(chk)::value;
            const int Lr = L + G.a + 1;
            uint32_t P = fa;
            {
                const uint32_t hh = s_pre0[Lr];
                P = hh < P ? hh : P;
            }
            uint32_t xp = 0, acc = 0, lacc = 0; /* change bits / "entering element < previous minimum" bits, newest in bit 0 */
            uint32_t dlo = SK2_INF, dhi = 0;    /* extremes of (entering element - previous minimum), wrapping */
#pragma unroll
            for (int j = 0; j <= C; j++) {
                const int rt = R0 + j;
                uint32_t hh = 0;
                if (j > 0) {
                    const int tp = rt - 1 < C ? rt - 1 : rt - 1 - C;
                    const int Lb = rt - 1 < C ? Lr : Lr + 1;
                    hh = s_c[tp * ST + Lb];
                    P = hh < P ? hh : P;
                }
                uint32_t x = P;
                if (j < C) x = P < c[j] ? P : c[j];
                if (j > 0) {
                    uint32_t d = hh - xp;
                    if (!CHECK) { acc = ntl_shl1_or_ne(acc, x, xp); lacc = ntl_shl1_or_lt_diff(lacc, hh, xp, d); }
                    else if (e_lane + j + G.w <= (int64_t)I.M && e_lane + j >= 0) {
                        chg |= (x != xp ? 1u : 0u) << j;
                        le |= (hh < xp ? 1u : 0u) << j;
                    } else d = 0x80000000u;
                    dlo = d < dlo ? d : dlo;
                    dhi = d > dhi ? d : dhi;
                }
                xp = x;
            }
            if (!CHECK) { chg = ntl_brev(acc) >> 15; le = ntl_brev(lacc) >> 15; } /* bit 16-j of acc is window j */
            return dlo <= SK2_NEAR || dhi >= 0u - SK2_NEAR;
        };
        if (own) {
            const bool near = all_inside ? window_pass(NtlFalse()) : window_pass(NtlTrue());
            /* an entering element within SK2_NEAR of the previous window's minimum, on either side: their order is open */
            if (near) {
                s_flag = 2u;
#ifdef NTL_SIM
                if (getenv("NTL_SK2_DEBUG")) fprintf(stderr, "strip %u lane %d near tie M=%u E0=%d\n", strip, L, I.M, I.E0);
#endif
            }
        }
    }
    if (own) {
        /* window (0,0) belongs to the previous strip, so the strip's first owned window (0,1) is always searched;
           window (LW-1,16) is the next strip's */
        if (L == 0) {
            if (e_lane + 1 + G.w <= (int64_t)I.M) chg |= 2u;
            le &= ~2u;
            /* Lists: window 0 is searched as well (bit 0 is nobody's: job 0) -- its minimum is the previous strip's to list, not
               this one's, whether window 1 has the same or not (StripLists); the bitmask takes the same bit twice. */
            if (lists && e_lane >= 0) chg |= 1u;
        }
        if (L == G.LW - 1) { chg &= 0xFFFFu; le &= 0xFFFFu; }
        /* entering element < previous minimum (and not near it): the minimum dropped, and only the entering element can be
           below the old minimum: it is the new minimum, alone within SK2_NEAR, no search needed */
        /* ---- phase 5: dropped minima are set right away (the entering element); one job per window whose minimum rose ---- */
        const uint32_t drop = chg & le, rise = chg & ~le;
        if (drop) { /* window j's entering element sits at strip position L*C + j + w - 1: the lane's drop bits, shifted, are its mask bits */
            const uint32_t p0 = (uint32_t)(L * C + G.w - 1);
            const uint64_t v = (uint64_t)drop << (p0 & 31u);
            atomicOr(&s_bits[p0 >> 5], (uint32_t)v);
            if (v >> 32) atomicOr(&s_bits[(p0 >> 5) + 1], (uint32_t)(v >> 32));
        }
        if (rise) {
            uint32_t at = atomicAdd(&s_njobs, (uint32_t)__popc(rise));
            uint32_t m = rise;
            while (m) {
                const int j = __ffs(m) - 1;
                m &= m - 1;
                if (at < (uint32_t)JOBCAP) s_jobs[at] = (uint16_t)(L * C + j);
                at++;
            }
        }
    }
    __syncthreads();

    /* ---- phase 6: position of the minimum of every such window [g, ge) ----
       Four lanes per window: lane q takes elements q, q+4, q+8, q+12 of the window's first block and of its last block
       and every 4th block minimum in between.  v = minimum over the quad (v_min with DPP quad permutes); a carry chain
       collects which of a lane's values equal v; v must occur exactly once in the quad.  A block minimum is resolved by
       the four lanes reading that block's elements, the same way. */
    {
        uint32_t njobs = SK2_DBG(B, 1) ? 0u : s_njobs;
        if (njobs > (uint32_t)JOBCAP) { njobs = JOBCAP; if (L == 0) s_flag = 4u; }
        const uint32_t q = (uint32_t)L & 3u, grp = (uint32_t)L >> 2;
        const uint32_t rounds = (njobs + NT / 4 - 1) / (NT / 4);
        const uint32_t nmid = (uint32_t)G.a + 1u; /* the middle blocks number a or a+1 */
        for (uint32_t it = 0; it < rounds; it++) {
            const uint32_t i = it * (NT / 4) + grp;
            const bool act = i < njobs;
            if (__ballot(act) == 0ull) break; /* jobs fill the wavefronts in order: nothing left for this one */
            const uint32_t g = s_jobs[act ? i : 0u], ge = g + (uint32_t)G.w;
            const uint32_t b0 = g >> 4, b1 = (ge - 1) >> 4, t0 = g & 15u, t1 = (ge - 1) & 15u;
            uint32_t val[12];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t t = q + 4u * r, b = b0 + 1 + q + 4u * r;
                val[r] = s_c[t * ST + b0];
                val[4 + r] = s_c[t * ST + b1];
                val[8 + r] = s_bm[b < (uint32_t)(NT + PAD - 1) ? b : (uint32_t)(NT + PAD - 1)];
            }
            uint32_t v = SK2_INF;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t t = q + 4u * r, b = b0 + 1 + q + 4u * r;
                val[r] = (t >= t0 && (b1 > b0 || t <= t1)) ? val[r] : SK2_INF;
                val[4 + r] = (b1 > b0 && t <= t1) ? val[4 + r] : SK2_INF;
                val[8 + r] = b < b1 ? val[8 + r] : SK2_INF;
                v = val[r] < v ? val[r] : v;
                v = val[4 + r] < v ? val[4 + r] : v;
                v = val[8 + r] < v ? val[8 + r] : v;
            }
            for (uint32_t b = b0 + 17 + q; nmid > 16 && b < b1; b += 4) { /* windows of more than 17 blocks (w > 271): the further middle blocks */
                const uint32_t u = s_bm[b];
                v = u < v ? u : v;
            }
            v = ntl_quad_min(v);
            const uint32_t vn = v + SK2_NEAR; /* a wrap (v within SK2_NEAR of 2^32) leaves no match: flagged */
            uint32_t mk = 0;
#pragma unroll
            for (int r = 0; r < 12; r++) mk = ntl_shl1_or_le(mk, val[r], vn); /* bit 11-r: val[r] within SK2_NEAR of the minimum */
            uint32_t xn = 0, xb = 0;
            for (uint32_t b = b0 + 17 + q; nmid > 16 && b < b1; b += 4)
                if (s_bm[b] <= vn) { xn++; xb = b; }
            uint32_t n = (uint32_t)__popc(mk) + xn;
            const uint32_t r1 = 11u - (uint32_t)(__ffs(mk | 0x1000u) - 1); /* the (last) matching value */
            uint32_t code = r1 < 4 ? b0 * 16 + q + 4u * r1 : (r1 < 8 ? b1 * 16 + q + 4u * (r1 - 4) : (0x10000u | (b0 + 1 + q + 4u * (r1 - 8))));
            if (mk == 0) code = 0x10000u | xb;
            n = ntl_quad_sum(v != SK2_INF ? n : 2u);
            code = ntl_quad_min(n && (mk || xn) ? code : SK2_INF);
            /* a block minimum: its sixteen elements, four per lane */
            const bool blk = code >= 0x10000u && code != SK2_INF;
            const uint32_t bb = blk ? (code & 0xFFFFu) : 0u;
            uint32_t mk2 = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) mk2 = ntl_shl1_or_le(mk2, s_c[(q + 4u * r) * ST + bb], vn);
            const uint32_t n2 = ntl_quad_sum((uint32_t)__popc(mk2));
            const uint32_t p2 = ntl_quad_min(mk2 ? bb * 16 + q + 4u * (3u - (uint32_t)(__ffs(mk2) - 1)) : SK2_INF);
            const bool ok = n == 1 && (!blk || n2 == 1);
            const uint32_t pos = blk ? p2 : code;
            if (act && q == 0) {
                if (ok && g == 0u) s_q0 = pos; /* (lists: the minimum of window 0) */
                else if (ok) atomicOr(&s_bits[pos >> 5], 1u << (pos & 31u));
                else {
                    s_flag = 8u;
#ifdef NTL_SIM
                    if (getenv("NTL_SK2_DEBUG")) fprintf(stderr, "strip %u job g=%u: minimum %x occurs %u/%u times (M=%u E0=%d)\n", strip, g, v, n, n2, I.M, I.E0);
#endif
                }
            }
        }
    }
    __syncthreads();

    /* ---- phase 7: proven minimizers to the global bitmask; flagged strips to the exact pass ---- */
    const uint32_t flagged = s_flag; /* written before the barrier above */
    if (lists) {
        if (!flagged)
            strip_bits_to_list<NT, NBW>(A.Ls, strip, s_bits, s_t0, [&](uint32_t i) -> uint32_t { return (uint32_t)(I.P0 + (int64_t)i); }, s_q0);
    } else if (L < NBW && !flagged) {
        const uint32_t word = s_bits[L];
        if (word) {
            const uint64_t g0 = (uint64_t)((int64_t)I.base + I.P0 + 32 * (int64_t)L);
            const uint32_t sh = (uint32_t)g0 & 31u;
            atomicOr(&A.mask[g0 >> 5], word << sh);
            if (sh && (word >> (32u - sh))) atomicOr(&A.mask[(g0 >> 5) + 1], word >> (32u - sh));
        }
    }
    if (L == 0 && flagged) B.redo_list[atomicAdd(B.redo_count, 1u)] = strip;
}

template <int NT, int R0, bool BIG>
__global__ __launch_bounds__(NT) void sketch_fast_kernel(Sketch2Args B)
{
    const uint32_t per_xcd = gridDim.x >> 3; /* consecutive strips on one XCD (see sketch_mask_kernel) */
    sk2_fast_strip<NT, R0, BIG>(B, (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3));
}

/* The same pass over a list of strips: the ones sketch_thresh_kernel gave up (a window without a candidate, mostly).  What this
   pass cannot decide either goes on to B.redo_list, the exact pass's list.  A fixed grid walks the list, whose length is only
   known on the device. */
template <int NT, int R0, bool BIG = false>
__global__ __launch_bounds__(NT) void sketch_fast_list_kernel(Sketch2Args B, const uint32_t *__restrict__ list, const uint32_t *__restrict__ count)
{
    const uint32_t n = *count;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        sk2_fast_strip<NT, R0, BIG>(B, list[i]); /* (BIG: windows of 256 .. 1135 k-mers, which sketch_wave_kernel takes since round 6) */
        __syncthreads(); /* the next strip's first writes to LDS behind this one's last reads */
    }
}


/*
 * sketch_lanes_kernel: the same window pass for 64 <= w <= 255, run only where a minimum can change.
 *
 * Same strip, same keys, same bits and the same flagged strips as sketch_fast_kernel<NT, R0, false>.  What differs: of the
 * 17 windows that start in a lane's block, the minimum changes in about 16 * 2 / (w + 1) of the lanes (one in eight at
 * w = 250).  Every lane still hashes, rolls and stages its sixteen keys, but then only decides -- from the block minima --
 * whether anything can happen in its windows:
 *     x0  = minimum of window 0   = min(own block, whole blocks L+1..L+a, first R0 elements of block L+a+1)
 *     x16 = minimum of window 16  = min(whole blocks L+1..L+a+1, first R0 elements of block L+a+2)
 *     ent >= a lower bound of the sixteen elements that enter at windows 1..16 = min(block L+a+1, first R0 of block L+a+2)
 * If x0 == x16 and ent > x0 + SK2_NEAR, the element m that attains x0 is still the lone minimum of window 16: everything else
 * in window 16 was in window 0 (so it is above x0 + SK2_NEAR by the invariant) or entered (above it by the test), hence the
 * value x0 in window 16 is m itself, m lies in all seventeen windows, and none of them changes its minimizer: the lane has
 * nothing to do and nothing to flag ("exact" (2) holds for each of its windows).  Every other lane -- and every lane at a
 * strip or sequence edge -- is put on a list, and the lanes of ONE wavefront walk the listed blocks' windows exactly as
 * sketch_fast_kernel's phases 4-5 do (keys read back from LDS).  Which wavefront: the strip index picks it, so that over
 * consecutive strips the four SIMDs of a CU get the same share of this single-wavefront phase (and of the search phase,
 * which another wavefront takes).  About 30 % fewer VALU instructions per strip than walking all 256 lanes' windows.
 */
template <int NT, int R0>
__global__ __launch_bounds__(NT) void sketch_lanes_kernel(Sketch2Args B)
{
    constexpr int C = 16;
    constexpr int NBW = (C * NT + 31) / 32;
    constexpr int ST = NT;
    constexpr int NX = NT + SK2_QMAX + 1;
    constexpr int PAD = 16;
    constexpr int JOBCAP = 128;
    constexpr uint32_t NW = NT / 64;
    __shared__ uint32_t s_c[C * ST];
    __shared__ uint32_t s_bm[NT + PAD];         /* block minima; INF behind NT */
    __shared__ uint32_t s_pre0[NT + 4];         /* minimum of the first R0 elements of each block */
    __shared__ uint32_t s_bits[NBW];
    __shared__ uint32_t s_njobs, s_flag, s_nalive;
    __shared__ uint32_t s_roll[64];             /* the rolling table; after phase 1b the list of lanes with work (uint8[NT]) */
    __shared__ uint32_t s_t0[NT + PAD];         /* range-minimum level; after the lanes' decision the job list (uint16[JOBCAP]) */
    static_assert(NT <= 256 && JOBCAP * 2 <= (NT + PAD) * 4, "the lists alias the rolling table and the range-minimum level");
    uint8_t *const s_list = (uint8_t *)&s_roll[0];
    uint16_t *const s_jobs = (uint16_t *)&s_t0[0];
    uint2 *const s_xy = (uint2 *)s_c;
    uint32_t *const s_so = (uint32_t *)&s_xy[2 * NX];
    static_assert(sizeof(uint32_t) * C * ST >= sizeof(uint2) * 2 * NX + sizeof(uint32_t) * (NX + 1), "the exchange area must fit the element array");

    const SketchArgs &A = B.A;
    const int L = threadIdx.x;
    const SketchGeom G = A.G;
    const uint32_t per_xcd = gridDim.x >> 3; /* consecutive strips on one XCD (see sketch_mask_kernel) */
    const uint32_t strip = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (strip >= A.nstrips) return;
    const StripInfo I = A.strip_tab[strip];
    if (I.seq == NTL_NONE || I.multi != 0) return; /* strips that cross non-ACGT runs: sketch_mask_kernel<.., MULTI = true> */
    if (L < 16) { s_roll[2 * L] = (uint32_t)(A.roll_tab[L][0] >> 33); s_roll[2 * L + 1] = (uint32_t)(A.roll_tab[L][1] >> 32); }
    if (L < NBW) s_bits[L] = 0;
    if (L < PAD) s_bm[NT + L] = SK2_INF;
    if (L < 4) s_pre0[NT + L] = SK2_INF;
    if (L == 0) { s_njobs = 0; s_nalive = 0; s_flag = B.force_redo ? 1u : 0u; }

    const int64_t e_lane = (int64_t)I.E0 + (int64_t)L * C; /* ordinal of this lane's element t = 0 */
    const uint64_t gp = (uint64_t)((int64_t)I.base + I.P0 + (int64_t)L * C);

    /* ---- phase 1a: 16-base partial hashes of the lane's own chunk (and of the chunks behind the strip) ---- */
    const bool live = e_lane < (int64_t)I.M;
    const bool feeds = e_lane - 16 * (int64_t)(B.q16 + 1) < (int64_t)I.M;
    uint32_t so = 0;
    if (feeds) {
        so = sk2_bases16(A.T.packed, gp, B.max_word);
        uint2 FU, P;
        sk2_chunk(so, B.r16, B, FU, P);
        s_xy[L] = FU;
        s_so[L] = so;
        if (B.r16) s_xy[NX + L] = P;
    }
    if (L <= B.q16) {
        const int64_t ev = (int64_t)I.E0 + (int64_t)(NT + L) * C;
        if (ev - 16 * (int64_t)(B.q16 + 1) < (int64_t)I.M) {
            const uint32_t sv = sk2_bases16(A.T.packed, gp + (uint64_t)NT * C, B.max_word);
            uint2 FU, P;
            sk2_chunk(sv, B.r16, B, FU, P);
            s_xy[NT + L] = FU;
            s_so[NT + L] = sv;
            if (B.r16) s_xy[NX + NT + L] = P;
        }
    }
    __syncthreads();

    /* ---- phase 1b: first k-mer's rings from the partials, then 15 rolling steps (as in sketch_fast_kernel) ---- */
    uint32_t c[C];
#pragma unroll
    for (int t = 0; t < C; t++) c[t] = SK2_INF;
    uint32_t fx = 0, ry = 0, si = 0;
    if (live) {
        {
            const uint32_t lo = s_so[L + B.q16];
            si = B.r16 ? ntl_alignbit(s_so[L + B.q16 + 1], lo, 2u * (uint32_t)B.r16) : lo;
        }
        uint32_t f = 0, u = 0;
        for (int i = 0; i < B.q16; i++) {
            if (i) { f = ring_rotl(f, 16); u = ring_rotr(u, 16); }
            const uint2 p = s_xy[L + i];
            f ^= p.x;
            u ^= p.y;
        }
        if (B.r16) {
            const uint32_t r = (uint32_t)B.r16;
            if (B.q16) { f = ring_rotl(f, r); u = ring_rotr(u, r); }
            const uint2 p = s_xy[NX + L + B.q16];
            f ^= p.x;
            u ^= p.y;
        }
        fx = f;
        ry = u << 1;
    }
    __syncthreads(); /* the partial hashes have been read: s_c may take the elements */
    if (live) {
        c[0] = (fx << 1) + ry;
        uint32_t wz[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t o2 = r < 2 ? so << (3 - 2 * r) : so >> (2 * r - 3);
            const uint32_t i2 = r < 3 ? si << (5 - 2 * r) : si >> (2 * r - 5);
            wz[r] = (o2 & 0x18181818u) | (i2 & 0x60606060u);
        }
#pragma unroll
        for (int t = 1; t < C; t++) {
            const int b = t - 1;
            const uint32_t off = ntl_bfe(wz[b & 3], 8u * (uint32_t)(b >> 2), 8u);
            const uint2 sd = *(const uint2 *)((const char *)s_roll + off);
            fx = ((fx << 1) | ((fx >> 30) & 1u)) ^ sd.x;
            const uint32_t a = ry ^ sd.y;
            ry = ntl_alignbit(a >> 1, a, 1);
            c[t] = (fx << 1) + ry;
        }
        if (e_lane < 0 || e_lane + C > (int64_t)I.M) { /* strip edges only */
#pragma unroll
            for (int t = 0; t < C; t++) {
                const int64_t e = e_lane + t;
                if (e < 0 || e >= (int64_t)I.M) c[t] = SK2_INF;
            }
        }
    }

    /* ---- phase 2: stage; minimum of the first R0 elements and of the whole block ---- */
    uint32_t pre0 = SK2_INF, bmv = SK2_INF;
#pragma unroll
    for (int t = 0; t < C; t++) {
        s_c[t * ST + L] = c[t];
        if (t < R0) pre0 = c[t] < pre0 ? c[t] : pre0;
        else bmv = c[t] < bmv ? c[t] : bmv;
    }
    bmv = pre0 < bmv ? pre0 : bmv;
    s_bm[L] = bmv;
    s_pre0[L] = pre0;
    __syncthreads(); /* also: the last read of the rolling table lies behind -- s_roll may take the lane list */

    /* ---- phase 3: minimum over the whole blocks L+1 .. L+a (one radix-4 level: a + 2 <= 16) ---- */
    uint32_t fa = SK2_INF;
    if (G.a >= 1) {
        const uint32_t *cur = s_bm;
        int span = 1;
        if (4 <= G.a) {
            uint32_t m = cur[L];
#pragma unroll
            for (int q = 1; q < 4; q++) { const uint32_t v = cur[L + q]; m = v < m ? v : m; }
            s_t0[L] = m;
            if (L < PAD) s_t0[NT + L] = SK2_INF;
            __syncthreads();
            cur = s_t0;
            span = 4;
        }
        for (int o = 0; o < G.a; o += span) {
            const int at = o + span <= G.a ? o : G.a - span;
            const uint32_t v = cur[L + 1 + at];
            fa = v < fa ? v : fa;
        }
    }

    /* ---- phase 4a: can the minimum change in this lane's windows?  (see the header of this kernel) ---- */
    const bool own = L < G.LW && e_lane + G.w <= (int64_t)I.M;
    {
        bool work = false;
        if (own) {
            const int Lr = L + G.a + 1;
            const uint32_t p1 = s_pre0[Lr], b1 = s_bm[Lr], p2 = s_pre0[Lr + 1];
            uint32_t x0 = bmv < fa ? bmv : fa;
            x0 = p1 < x0 ? p1 : x0;
            uint32_t x16 = fa < b1 ? fa : b1;
            x16 = p2 < x16 ? p2 : x16;
            const uint32_t ent = b1 < p2 ? b1 : p2;
            const bool inside = e_lane >= 0 && e_lane + C + G.w <= (int64_t)I.M; /* all 17 windows lie in the sequence */
            work = !inside || x0 != x16 || ent - x0 <= SK2_NEAR || L == 0 || L == G.LW - 1;
        }
        const unsigned long long bal = __ballot(work);
        __syncthreads(); /* every lane has read the range-minimum level: s_t0 may take the job list (and s_roll is long dead) */
        if (bal != 0ull) {
            uint32_t base = 0;
            const int first = __ffsll((long long)bal) - 1;
            if ((L & 63) == first) base = atomicAdd(&s_nalive, (uint32_t)__popcll(bal));
            base = __shfl(base, first);
            if (work) s_list[base + ntl_mbcnt(bal)] = (uint8_t)L;
        }
    }
    __syncthreads();

    /* ---- phases 4-5 for the listed lanes: wavefront (strip + round) mod NW takes round `round` of 64 list entries ---- */
    {
        const uint32_t nalive = s_nalive;
        /* which wavefront: a hash of the strip index (workgroups reach a CU in an order that correlates with their index:
           the plain index modulo NW would give every workgroup of a CU the same wavefront) */
        const uint32_t rot = (strip * 2654435761u) >> 30;
        const uint32_t round = (((uint32_t)L >> 6) + NW - (rot % NW)) % NW;
        const uint32_t li = round * 64u + ((uint32_t)L & 63u);
        const bool act = li < nalive;
        if (__ballot(act) != 0ull) {
            const int Lx = act ? (int)s_list[li] : 0;
            const int64_t ex = (int64_t)I.E0 + (int64_t)Lx * C;
            /* the block's keys back from LDS, suffix minima in place (c[j] = min of elements j..15) */
#pragma unroll
            for (int t = 0; t < C; t++) c[t] = s_c[t * ST + Lx];
#pragma unroll
            for (int j = C - 2; j >= 0; j--) c[j] = c[j] < c[j + 1] ? c[j] : c[j + 1];
            uint32_t fx2 = SK2_INF; /* whole blocks Lx+1 .. Lx+a */
            for (int o = 1; o <= G.a; o++) { const uint32_t v = s_bm[Lx + o]; fx2 = v < fx2 ? v : fx2; }
            uint32_t chg = 0, le = 0;
            const bool inside = ex >= 0 && ex + C + G.w <= (int64_t)I.M;
            const bool all_inside = __ballot(act && !inside) == 0ull;
            auto window_pass = [&](auto chk) {
                constexpr bool CHECK = decltype(chk)::value;
                const int Lr = Lx + G.a + 1;
                uint32_t P = fx2;
                {
                    const uint32_t hh = s_pre0[Lr];
                    P = hh < P ? hh : P;
                }
                uint32_t xp = 0, acc = 0, lacc = 0;
                uint32_t dlo = SK2_INF, dhi = 0;
#pragma unroll
                for (int j = 0; j <= C; j++) {
                    const int rt = R0 + j;
                    uint32_t hh = 0;
                    if (j > 0) {
                        const int tp = rt - 1 < C ? rt - 1 : rt - 1 - C;
                        const int Lb = rt - 1 < C ? Lr : Lr + 1;
                        hh = s_c[tp * ST + Lb];
                        P = hh < P ? hh : P;
                    }
                    uint32_t x = P;
                    if (j < C) x = P < c[j] ? P : c[j];
                    if (j > 0) {
                        uint32_t d = hh - xp;
                        if (!CHECK) { acc = ntl_shl1_or_ne(acc, x, xp); lacc = ntl_shl1_or_lt_diff(lacc, hh, xp, d); }
                        else if (ex + j + G.w <= (int64_t)I.M && ex + j >= 0) {
                            chg |= (x != xp ? 1u : 0u) << j;
                            le |= (hh < xp ? 1u : 0u) << j;
                        } else d = 0x80000000u;
                        dlo = d < dlo ? d : dlo;
                        dhi = d > dhi ? d : dhi;
                    }
                    xp = x;
                }
                if (!CHECK) { chg = ntl_brev(acc) >> 15; le = ntl_brev(lacc) >> 15; }
                return dlo <= SK2_NEAR || dhi >= 0u - SK2_NEAR;
            };
            if (act) {
                const bool near = all_inside ? window_pass(NtlFalse()) : window_pass(NtlTrue());
                if (near) {
                    s_flag = 2u;
#ifdef NTL_SIM
                    if (getenv("NTL_SK2_DEBUG")) fprintf(stderr, "strip %u lane %d near tie M=%u E0=%d\n", strip, Lx, I.M, I.E0);
#endif
                }
                if (Lx == 0) { if (ex + 1 + G.w <= (int64_t)I.M) chg |= 2u; le &= ~2u; }
                if (Lx == G.LW - 1) { chg &= 0xFFFFu; le &= 0xFFFFu; }
                const uint32_t drop = chg & le, rise = chg & ~le;
                if (drop) {
                    const uint32_t p0 = (uint32_t)(Lx * C + G.w - 1);
                    const uint64_t v = (uint64_t)drop << (p0 & 31u);
                    atomicOr(&s_bits[p0 >> 5], (uint32_t)v);
                    if (v >> 32) atomicOr(&s_bits[(p0 >> 5) + 1], (uint32_t)(v >> 32));
                }
                if (rise) {
                    uint32_t at = atomicAdd(&s_njobs, (uint32_t)__popc(rise));
                    uint32_t m = rise;
                    while (m) {
                        const int j = __ffs(m) - 1;
                        m &= m - 1;
                        if (at < (uint32_t)JOBCAP) s_jobs[at] = (uint16_t)(Lx * C + j);
                        at++;
                    }
                }
            }
        }
    }
    __syncthreads();

    /* ---- phase 6: position of the minimum of every window whose minimum rose (as in sketch_fast_kernel); the jobs start at
       another wavefront than the one that walked the windows ---- */
    {
        uint32_t njobs = s_njobs;
        if (njobs > (uint32_t)JOBCAP) { njobs = JOBCAP; if (L == 0) s_flag = 4u; }
        const uint32_t Lq = ((uint32_t)L + 64u * ((((strip * 2654435761u) >> 28) & 3u) % NW)) % (uint32_t)NT; /* rotated lane number: same quads */
        const uint32_t q = Lq & 3u, grp = Lq >> 2;
        const uint32_t rounds = (njobs + NT / 4 - 1) / (NT / 4);
        for (uint32_t it = 0; it < rounds; it++) {
            const uint32_t i = it * (NT / 4) + grp;
            const bool act = i < njobs;
            if (__ballot(act) == 0ull) break;
            const uint32_t g = s_jobs[act ? i : 0u], ge = g + (uint32_t)G.w;
            const uint32_t b0 = g >> 4, b1 = (ge - 1) >> 4, t0 = g & 15u, t1 = (ge - 1) & 15u;
            uint32_t val[12];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t t = q + 4u * r, b = b0 + 1 + q + 4u * r;
                val[r] = s_c[t * ST + b0];
                val[4 + r] = s_c[t * ST + b1];
                val[8 + r] = s_bm[b < (uint32_t)(NT + PAD - 1) ? b : (uint32_t)(NT + PAD - 1)];
            }
            uint32_t v = SK2_INF;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const uint32_t t = q + 4u * r, b = b0 + 1 + q + 4u * r;
                val[r] = (t >= t0 && (b1 > b0 || t <= t1)) ? val[r] : SK2_INF;
                val[4 + r] = (b1 > b0 && t <= t1) ? val[4 + r] : SK2_INF;
                val[8 + r] = b < b1 ? val[8 + r] : SK2_INF;
                v = val[r] < v ? val[r] : v;
                v = val[4 + r] < v ? val[4 + r] : v;
                v = val[8 + r] < v ? val[8 + r] : v;
            }
            v = ntl_quad_min(v);
            const uint32_t vn = v + SK2_NEAR;
            uint32_t mk = 0;
#pragma unroll
            for (int r = 0; r < 12; r++) mk = ntl_shl1_or_le(mk, val[r], vn);
            uint32_t n = (uint32_t)__popc(mk);
            const uint32_t r1 = 11u - (uint32_t)(__ffs(mk | 0x1000u) - 1);
            uint32_t code = r1 < 4 ? b0 * 16 + q + 4u * r1 : (r1 < 8 ? b1 * 16 + q + 4u * (r1 - 4) : (0x10000u | (b0 + 1 + q + 4u * (r1 - 8))));
            n = ntl_quad_sum(v != SK2_INF ? n : 2u);
            code = ntl_quad_min(n && mk ? code : SK2_INF);
            const bool blk = code >= 0x10000u && code != SK2_INF;
            const uint32_t bb = blk ? (code & 0xFFFFu) : 0u;
            uint32_t mk2 = 0;
#pragma unroll
            for (int r = 0; r < 4; r++) mk2 = ntl_shl1_or_le(mk2, s_c[(q + 4u * r) * ST + bb], vn);
            const uint32_t n2 = ntl_quad_sum((uint32_t)__popc(mk2));
            const uint32_t p2 = ntl_quad_min(mk2 ? bb * 16 + q + 4u * (3u - (uint32_t)(__ffs(mk2) - 1)) : SK2_INF);
            const bool ok = n == 1 && (!blk || n2 == 1);
            const uint32_t pos = blk ? p2 : code;
            if (act && q == 0) {
                if (ok) atomicOr(&s_bits[pos >> 5], 1u << (pos & 31u));
                else {
                    s_flag = 8u;
#ifdef NTL_SIM
                    if (getenv("NTL_SK2_DEBUG")) fprintf(stderr, "strip %u job g=%u: minimum %x occurs %u/%u times (M=%u E0=%d)\n", strip, g, v, n, n2, I.M, I.E0);
#endif
                }
            }
        }
    }
    __syncthreads();

    /* ---- phase 7: proven minimizers to the global bitmask; flagged strips to the exact pass ---- */
    const uint32_t flagged = s_flag;
    if (L < NBW && !flagged) {
        const uint32_t word = s_bits[L];
        if (word) {
            const uint64_t g0 = (uint64_t)((int64_t)I.base + I.P0 + 32 * (int64_t)L);
            const uint32_t sh = (uint32_t)g0 & 31u;
            atomicOr(&A.mask[g0 >> 5], word << sh);
            if (sh && (word >> (32u - sh))) atomicOr(&A.mask[(g0 >> 5) + 1], word >> (32u - sh));
        }
    }
    if (L == 0 && flagged) B.redo_list[atomicAdd(B.redo_count, 1u)] = strip;
}


/*
 * sketch_thresh_kernel: the window pass on threshold-sparsified windows -- the default for 71 <= w <= 255.
 *
 * Same strip, same keys, same contract as sketch_fast_kernel<NT, R0, false>: a bit per k-mer that is the argmin of one of the
 * windows the strip answers for, or the strip on a list for another pass and none of its bits.  What differs is everything behind the
 * rolling: no block minima, no 17-window pass, no search jobs.  Only the k-mers with key < T ("candidates"; T = 2^32 * c / w for
 * about c = 10 candidates per window) are looked at again.  They are compacted, in position order, into a list in LDS, and one
 * lane per candidate decides whether it is the lone minimum of some window:
 *
 *   blocker of candidate i   a candidate with key <= key_i + SK2_NEAR (what could be the 64-bit argmin in i's place).
 *   Rp   position of the nearest blocker to the right, at most pos_i + w  (the list's right sentinel sits at the end `hi` of the
 *        strip's elements with key 0: no window reaches beyond it);
 *   i is the argmin of a window  <=  there is no blocker in [Rp - w, pos_i)  (left sentinel: position 0, key 0 -- the strip's
 *        windows start at element 1).  Then the window [Rp - w, Rp) holds i, no other candidate within SK2_NEAR of it or below,
 *        and non-candidates are >= T > key_i + SK2_NEAR (a candidate with key + SK2_NEAR >= T flags the strip): by "exact" in
 *        the header of this file i is that window's 64-bit argmin.  The window lies in the sequence, so the bit is right whoever
 *        owns the window.
 *
 *   Nothing is missed unless the strip is flagged.  Take a window W of the strip.  (a) No candidate in W: then two consecutive
 *   list entries (sentinels included) are more than w apart -- every candidate checks the distance to its successor, lane 0 also
 *   the first one's to element 0 -- flag.  (b) The smallest candidate i of W has another candidate j of W within SK2_NEAR: the LEFT
 *   one of the two scans to the right until its first blocker b, which lies between them (or is the other one), hence in W,
 *   hence key_b >= key_i; key_b is also <= (the scanning one's key) + SK2_NEAR, and both keys are within SK2_NEAR of key_i: the
 *   scan sees a blocker within SK2_NEAR of its own key -- flag.  That is why the right scan is never cut short, while the left one
 *   only covers what the decision needs.  (c) Otherwise i has no blocker inside W, Rp lies behind W, the start of W is
 *   <= Rp - w, and a blocker in [Rp - w, pos_i) would lie in W: there is none, the bit is set.
 *
 * More candidates than the list holds in a strip: flag.  A flagged strip goes on B.fb_list: sketch_fast_list_kernel decides it with
 * the block-minima pass, and what that cannot decide either (near ties) takes the exact pass.
 */
#define SK2T_CAP 402
#define SK2T_CAP_DIRECT 680

/* index of the first of four keys that is <= lim, 4 if none: straight-line (compare + add-with-carry per key, one v_ffbl), so
   that the four LDS reads behind it are issued together instead of one per taken branch */
__device__ __forceinline__ uint32_t sk2t_first_le(uint32_t k0, uint32_t k1, uint32_t k2, uint32_t k3, uint32_t lim)
{
    return (uint32_t)__ffs(ntl_le4_mask(k0, k1, k2, k3, lim)) - 1u;
}

template <int NT, bool DIRECT>
__global__ __launch_bounds__(NT) void sketch_thresh_kernel(Sketch2Args B)
{
    constexpr int C = 16;
    constexpr int NBW = (C * NT + 31) / 32;
    constexpr int ST = NT;
    constexpr int NX = NT + SK2_QMAX + 1;
    constexpr uint32_t NW = NT / 64;
    /* DIRECT: the candidates go from the registers to the list with sixteen predicated writes, the keys are not staged: the
       element array shrinks to the exchange area of phase 1, one barrier goes, and the list may hold more (smaller w) */
    constexpr uint32_t CAP = DIRECT ? SK2T_CAP_DIRECT : SK2T_CAP;
    constexpr int XW = (int)((sizeof(uint2) * 2 * NX + sizeof(uint32_t) * (NX + 1)) / sizeof(uint32_t));
    /* the four-base tables (2048 words) live behind the exchange area while phase 1a needs them: in the second half of the element
       array, which only takes the staged keys two barriers later, or -- DIRECT -- in words of their own */
    constexpr int TOFF = DIRECT ? ((XW + 3) & ~3) : (C * ST) / 2;
    static_assert(TOFF >= XW && (DIRECT || TOFF + 2048 <= C * ST), "the tables must not overlap the exchange area");
    __shared__ __attribute__((aligned(16))) uint32_t s_c[DIRECT ? TOFF + 2048 : C * ST];
    __shared__ uint2 s_cand[CAP + 8];      /* {key, position}: [3] left sentinel, [4 .. n + 3] the candidates, [n + 4] right sentinel;
                                                   [0 .. 2] and [n + 5 .. n + 7]: copies of the sentinels that the four-entry scan steps read along */
    __shared__ uint32_t s_bits[NBW];
    __shared__ uint32_t s_roll[64];
    __shared__ uint32_t s_wsum[NW];
    __shared__ uint32_t s_flag;
    uint2 *const s_xy = (uint2 *)s_c;
    uint32_t *const s_so = (uint32_t *)&s_xy[2 * NX];
    static_assert(sizeof(uint32_t) * C * ST >= sizeof(uint2) * 2 * NX + sizeof(uint32_t) * (NX + 1), "the exchange area must fit the element array");

    const SketchArgs &A = B.A;
    const int L = threadIdx.x;
    const SketchGeom G = A.G;
    const uint32_t per_xcd = gridDim.x >> 3; /* consecutive strips on one XCD (see sketch_mask_kernel) */
    const uint32_t strip = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (strip >= A.nstrips) return;
    static_assert(NT == 256, "one 32-byte piece of the 8-KB tables per lane");
    const uint4 tb0 = ((const uint4 *)B.g4k)[2 * L], tb1 = ((const uint4 *)B.g4k)[2 * L + 1];
    const StripInfo I = A.strip_tab[strip];
    if (I.seq == NTL_NONE || I.multi != 0) return; /* strips that cross non-ACGT runs: sketch_mask_kernel<.., MULTI = true> */
    if (L < 16) { s_roll[2 * L] = (uint32_t)(A.roll_tab[L][0] >> 33); s_roll[2 * L + 1] = (uint32_t)(A.roll_tab[L][1] >> 32); }
    if (L < NBW) s_bits[L] = 0;
    if (L == 0) s_flag = B.force_redo ? 1u : 0u;
    if (L < 4) s_cand[L] = make_uint2(0u, 0u);

    const int64_t e_lane = (int64_t)I.E0 + (int64_t)L * C; /* ordinal of this lane's element t = 0 */
    const uint64_t gp = (uint64_t)((int64_t)I.base + I.P0 + (int64_t)L * C);

    /* ---- phase 1a: 16-base partial hashes of the lane's own chunk (and of the chunks behind the strip).  The base words and the
       workgroup's copy of the four-base tables are requested together; the lookups then stay inside the CU. ---- */
    const bool live = e_lane < (int64_t)I.M;
    const bool feeds = e_lane - 16 * (int64_t)(B.q16 + 1) < (int64_t)I.M;
    const bool feeds2 = L <= B.q16 && (int64_t)I.E0 + (int64_t)(NT + L) * C - 16 * (int64_t)(B.q16 + 1) < (int64_t)I.M; /* chunks NT .. NT+q16 feed the last lanes */
    uint2 *const s_g4k = (uint2 *)&s_c[TOFF];
    uint32_t so = 0, sv = 0;
    if (feeds && !SK2_DBG(B, 8)) so = sk2_bases16(A.T.packed, gp, B.max_word);
    if (feeds2) sv = sk2_bases16(A.T.packed, gp + (uint64_t)NT * C, B.max_word);
    ((uint4 *)s_g4k)[2 * L] = tb0;
    ((uint4 *)s_g4k)[2 * L + 1] = tb1;
    __syncthreads();
    auto chunk = [&](const uint32_t w, uint2 &FU, uint2 &P) {
        const uint2 g0 = s_g4k[w & 255u], g1 = s_g4k[256u + ((w >> 8) & 255u)], g2 = s_g4k[512u + ((w >> 16) & 255u)], g3 = s_g4k[768u + (w >> 24)];
        FU = make_uint2(g0.x ^ g1.x ^ g2.x ^ g3.x, g0.y ^ g1.y ^ g2.y ^ g3.y);
        P = make_uint2(0u, 0u);
        if (B.r16 == 8) {
            const uint2 p0 = s_g4k[512u + (w & 255u)], p1 = s_g4k[768u + ((w >> 8) & 255u)];
            P = make_uint2(p0.x ^ p1.x, p0.y ^ p1.y);
        } else if (B.r16) {
            uint2 fu;
            sk2_chunk(w, B.r16, B, fu, P); /* k % 16 not in {0, 8}: the first r bases on the plain 64-bit tables */
        }
#ifdef NTL_SIM
        {
            uint2 fu, pp;
            sk2_chunk(w, B.r16, B, fu, pp);
            if (fu.x != FU.x || fu.y != FU.y || (B.r16 && (pp.x != P.x || pp.y != P.y))) { fprintf(stderr, "g4k tables disagree with g8k (k %% 16 = %d)\n", B.r16); abort(); }
        }
#endif
    };
    if (feeds && !SK2_DBG(B, 8)) {
        uint2 FU, P;
        if (SK2_DBG(B, 16)) { FU = make_uint2(so * 2654435761u, so ^ 0x9E3779B9u); P = FU; } /* ablation: no table lookups */
        else chunk(so, FU, P);
        s_xy[L] = FU;
        s_so[L] = so;
        if (B.r16) s_xy[NX + L] = P;
    }
    if (feeds2) {
        uint2 FU, P;
        chunk(sv, FU, P);
        s_xy[NT + L] = FU;
        s_so[NT + L] = sv;
        if (B.r16) s_xy[NX + NT + L] = P;
    }
    __syncthreads();

    /* ---- phase 1b: first k-mer's rings from the partials, then 15 rolling steps (as in sketch_fast_kernel) ---- */
    uint32_t c[C];
#pragma unroll
    for (int t = 0; t < C; t++) c[t] = SK2_INF;
    uint32_t fx = 0, ry = 0, si = 0;
    if (live) {
        {
            const uint32_t lo = s_so[L + B.q16];
            si = B.r16 ? ntl_alignbit(s_so[L + B.q16 + 1], lo, 2u * (uint32_t)B.r16) : lo;
        }
        uint32_t f = 0, u = 0;
        for (int i = 0; i < B.q16; i++) {
            if (i) { f = ring_rotl(f, 16); u = ring_rotr(u, 16); }
            const uint2 p = s_xy[L + i];
            f ^= p.x;
            u ^= p.y;
        }
        if (B.r16) {
            const uint32_t r = (uint32_t)B.r16;
            if (B.q16) { f = ring_rotl(f, r); u = ring_rotr(u, r); }
            const uint2 p = s_xy[NX + L + B.q16];
            f ^= p.x;
            u ^= p.y;
        }
        fx = f;
        ry = u << 1;
    }
    if constexpr (!DIRECT) __syncthreads(); /* the partial hashes have been read: s_c may take the elements */
    if (live) {
        c[0] = (fx << 1) + ry;
        uint32_t wz[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t o2 = r < 2 ? so << (3 - 2 * r) : so >> (2 * r - 3);
            const uint32_t i2 = r < 3 ? si << (5 - 2 * r) : si >> (2 * r - 5);
            wz[r] = (o2 & 0x18181818u) | (i2 & 0x60606060u);
        }
#pragma unroll
        for (int t = 1; t < C; t++) {
            const int b = t - 1;
            const uint32_t off = ntl_bfe(wz[b & 3], 8u * (uint32_t)(b >> 2), 8u);
            const uint2 sd = *(const uint2 *)((const char *)s_roll + off);
            fx = ((fx << 1) | ((fx >> 30) & 1u)) ^ sd.x;
            const uint32_t a = ry ^ sd.y;
            ry = ntl_alignbit(a >> 1, a, 1);
            c[t] = (fx << 1) + ry;
        }
        if (e_lane < 0 || e_lane + C > (int64_t)I.M) { /* strip edges only */
#pragma unroll
            for (int t = 0; t < C; t++) {
                const int64_t e = e_lane + t;
                if (e < 0 || e >= (int64_t)I.M) c[t] = SK2_INF;
            }
        }
    }
    if (L == 0) c[0] = SK2_INF; /* element 0 belongs to the windows of the previous strip only */

    /* ---- phase 2: (stage the keys;) which of them are candidates, and how many (bit 15 - t of acc: element t) ---- */
    const uint32_t tm1 = B.thresh - 1u;
    uint32_t acc = 0;
#pragma unroll
    for (int t = 0; t < C; t++) {
        if constexpr (!DIRECT) s_c[t * ST + L] = c[t];
        acc = ntl_shl1_or_le(acc, c[t], tm1);
    }
    uint32_t m16 = ntl_brev(acc) >> 16; /* bit t: element t */
    const uint32_t ncl = (uint32_t)__popc(m16);
    const uint32_t incl = ntl_wave_incl_scan(ncl);
    if ((L & 63) == 63) s_wsum[L >> 6] = incl;
    __syncthreads();
    uint32_t at = incl - ncl, total = 0;
#pragma unroll
    for (uint32_t q = 0; q < NW; q++) {
        const uint32_t v = s_wsum[q];
        if (q < ((uint32_t)L >> 6)) at += v;
        total += v;
    }
    const uint32_t hi = (uint32_t)((int64_t)I.M - (int64_t)I.E0 < (int64_t)(NT * C) ? (int64_t)I.M - (int64_t)I.E0 : (int64_t)(NT * C));
    const bool over = total > CAP; /* the strip is given up: an empty list (nothing is written, nothing scanned) */
    const uint32_t n = over ? 0u : total;
    if (L >= NT - 4) { /* the right sentinel, and copies of it behind it for the scan steps to read along */
        s_cand[n + 4 + (uint32_t)(NT - 1 - L)] = make_uint2(0u, hi);
        if (L == NT - 1 && over) s_flag = 16u;
    }
    /* ---- phase 3: the lane's candidates into the list ---- */
    if constexpr (DIRECT) {
        if (!over) {
            uint2 *dst = &s_cand[at + 4];
#pragma unroll
            for (int t = 0; t < C; t++)
                if (c[t] <= tm1) *dst++ = make_uint2(c[t], (uint32_t)(L * C + t));
        }
    } else if (m16 && !over) { /* its own staged keys back from LDS by index; the read of the next one is in flight while the previous one is written */
        uint32_t t = (uint32_t)__ffs(m16) - 1u;
        m16 &= m16 - 1u;
        uint32_t v = s_c[t * ST + L];
        for (;;) {
            const uint32_t t0 = t, v0 = v;
            const bool more = m16 != 0u;
            if (more) {
                t = (uint32_t)__ffs(m16) - 1u;
                m16 &= m16 - 1u;
                v = s_c[t * ST + L];
            }
            s_cand[at + 4] = make_uint2(v0, (uint32_t)(L * C) + t0);
            at++;
            if (!more) break;
        }
    }
    __syncthreads();

    /* ---- phase 4: one lane per candidate (see the header); the scans take four list entries per step ---- */
    {
        const uint32_t w = (uint32_t)G.w;
        bool bad = false;
        if (L == 0) bad = s_cand[4].y - 1u >= w; /* elements 1 .. w without a candidate */
        for (uint32_t i = (uint32_t)L; i < n; i += NT) {
            const uint2 me = s_cand[i + 4];
            const uint32_t lim = me.x + SK2_NEAR;
            bad |= lim >= B.thresh;
            uint32_t Rp = me.y + w;
            /* Positions grow along the list, so a scan only asks each entry "blocker?" and the chunk's farthest entry "still in
               range?"; which blocker came first (fb, 4 = none) and whether it lies in range is settled behind the loop. */
            {   /* to the right: the first blocker nearer than w; ends at the right sentinel (key 0) at the latest */
                uint32_t j = i + 5, fb;
                bad |= s_cand[j].y - me.y - 1u >= w; /* a window between two candidates */
                for (;;) {
                    const uint2 q0 = s_cand[j], q1 = s_cand[j + 1], q2 = s_cand[j + 2], q3 = s_cand[j + 3];
                    fb = sk2t_first_le(q0.x, q1.x, q2.x, q3.x, lim);
                    if (fb != 4u || q3.y >= Rp) break;
                    j += 4;
                }
                if (fb != 4u) {
                    const uint2 e = s_cand[j + fb];
                    if (e.y < Rp) {
                        Rp = e.y;
                        bad |= e.x + SK2_NEAR >= me.x; /* within SK2_NEAR (the sentinel's key 0: only for keys <= SK2_NEAR) */
                    }
                }
            }
            const int32_t need = (int32_t)Rp - (int32_t)w; /* a blocker at q < pos matters where q >= need */
            bool blocked = false;
            {   /* to the left: is there a blocker at or behind `need`?  Ends at the left sentinel (position 0, key 0) at the latest:
                   the step that holds the sentinel (index 3) stops the scan, and reads no index below 0 */
                uint32_t j = i + 3, fb;
                for (;;) {
                    const uint2 q0 = s_cand[j], q1 = s_cand[j - 1], q2 = s_cand[j - 2], q3 = s_cand[j - 3];
                    fb = sk2t_first_le(q0.x, q1.x, q2.x, q3.x, lim);
                    if (fb != 4u || (int32_t)q3.y < need) break;
                    j -= 4;
                }
                if (fb != 4u) blocked = (int32_t)s_cand[j - fb].y >= need;
            }
            if (!blocked) atomicOr(&s_bits[me.y >> 5], 1u << (me.y & 31u));
        }
        if (bad) {
            s_flag = 2u;
#ifdef NTL_SIM
            if (getenv("NTL_SK2_DEBUG")) fprintf(stderr, "strip %u lane %d: threshold pass gives up (n=%u M=%u E0=%d)\n", strip, L, n, I.M, I.E0);
#endif
        }
    }
    __syncthreads();

    /* ---- phase 7: proven minimizers to the global bitmask; flagged strips to the block-minima pass ---- */
    const uint32_t flagged = s_flag;
    if (L < NBW && !flagged) {
        const uint32_t word = s_bits[L];
        if (word) {
            const uint64_t g0 = (uint64_t)((int64_t)I.base + I.P0 + 32 * (int64_t)L);
            const uint32_t sh = (uint32_t)g0 & 31u;
            atomicOr(&A.mask[g0 >> 5], word << sh);
            if (sh && (word >> (32u - sh))) atomicOr(&A.mask[(g0 >> 5) + 1], word >> (32u - sh));
        }
    }
    if (L == 0 && flagged) B.fb_list[atomicAdd(B.fb_count, 1u)] = strip;
}


/*
 * sketch_wave_kernel: the threshold pass (see sketch_thresh_kernel above for the algorithm and its proof) with ONE WAVEFRONT PER
 * STRIP, 64 k-mers per lane, and wavefronts that stay resident and walk over their strips -- round 4.
 *
 * Same strips (4096 consecutive valid-k-mer ordinals, strip_table_kernel's geometry for NT = 256, C = 16), same keys, same
 * candidates, same decision per candidate, same lists for the passes behind it; so a strip it gives up is taken over by
 * sketch_fast_list_kernel exactly as before.  What differs is who does the work:
 *
 *   - a lane owns 64 consecutive k-mers instead of 16: the first k-mer's hash (the table lookups, the combination of the 16-base
 *     partial hashes) is paid once per 64 k-mers, and for k <= 64 it is made from the lane's OWN bases -- no exchange of partial
 *     hashes or base words between lanes, no exchange area in LDS;
 *   - the keys are never stored: each key is tested against the threshold as it leaves the rolling step, and a candidate is
 *     written at once into the lane's own staging slots (S per lane: a lane holds 64 p = 2.6 candidates on average at ten per
 *     window of 250) as ONE word: the key with its six low bits replaced by the step t it was made in (SKW_NEAR below); the lanes
 *     then take their staged candidates into registers and write them back, in position order, over the same words -- staging
 *     area and list of keys are one array -- with the positions 64 L + t as words at a constant distance behind them: 4 KB of LDS
 *     per wavefront, so that a CU holds 32 of them;
 *   - the wavefronts of a workgroup share nothing but the read-only tables (the four-base ring tables and the rolling seeds,
 *     copied into LDS once per workgroup LIFETIME), so after the one barrier behind that copy no barrier is left: a wavefront
 *     walks through first k-mers -> rolling -> list -> scans -> bits at its own pace, and the wavefronts of a CU are in
 *     different phases at any time (issue-bound rolling of one beside the LDS-latency-bound scans of another).  The phases of
 *     sketch_thresh_kernel were separated by six workgroup barriers;
 *   - a wavefront TAKES its strips, SKW_CHUNK consecutive ones at a time, from a counter of its XCD's share of the strips
 *     (dealt out in advance the launch ends with the slowest share: 2.55 ms per C3 launch against 2.09 taken), and asks for
 *     the 16-byte strip table entry two strips ahead and for the base words one strip ahead (vector loads: a scalar load would sit in the same counter as the LDS
 *     reads of the rolling loop), so that no strip starts with a chain of dependent global loads;
 *   - hand-offs between lanes (the list, the sentinels) are wave-synchronous LDS traffic (ntl_wave_sync: ordering only).
 *
 * Elements: local position p = 64 L + t of the strip is ordinal E0 + p; p = 0 belongs to the previous strip's windows only
 * and positions >= hi = min(4096, M - E0) lie behind the sequence: candidates found there are dropped when the list is made.
 * Gives the strip up (B.fb_list) when a lane stages more than S candidates, the list would hold more than 64 ROUNDS - 8, or the
 * scans say so (a window without a candidate, a near tie, a key within SKW_NEAR of the threshold).  Round 6: one pass over the list
 * decides which candidates scan at all (below: "who has to scan"), and the window may be anything up to the block-minima pass's 1135
 * k-mers -- it only enters the scans as a distance (the workgroup-per-strip passes' a + 2 <= 16 lanes is not this kernel's limit).
 * Soaked against the oracle on
 * 24 Gbases and 985 random (k, w, candidates per window) configurations (profiles/r04s_*.log), and in every `pytest -m gpu` run
 * on 25 Gbases more (tests/test_gpu_soak.py).
 */
struct SkwWords { uint4 o0, i0; uint32_t o4, i4, ao, ai; };

__device__ __forceinline__ uint4 skw_strip_load(const StripLite *tab, uint32_t strip, uint32_t end)
{
    uint4 v = make_uint4(0u, 0u, 0u, 0u); /* {g0 lo, g0 hi, hi, -}; hi = 0: nothing to do */
    if (strip < end) v = *(const uint4 *)&tab[strip];
    return v;
}

/* k-mers per lane of a strip of `hi` elements: 64, or -- a sequence's last strip -- the multiple of 16 that still covers it: the
   wavefront rolls 16, 32 or 48 steps instead of 64, all lanes at work (round 5; a tail strip used to roll 64 steps with the lanes
   behind the sequence's end idle: 15 % of all lane-steps at C3; blocks of eight steps instead of sixteen: no shorter, 2.114 against
   2.102 ms per C3 launch) */
__device__ __forceinline__ uint32_t skw_per_lane(uint32_t hi_raw)
{
    const uint32_t hi = hi_raw & 0xFFFFu;
    return hi > 3072u ? 64u : ((hi + 1023u) >> 10) << 4;
}

__device__ __forceinline__ SkwWords skw_words_load(const uint32_t *__restrict__ packed, const uint4 I, int L, int k, uint32_t wmax)
{
    SkwWords W;
    const uint64_t gp = (((uint64_t)I.y << 32) | I.x) + (uint64_t)((uint32_t)L * skw_per_lane(I.z));
    const uint64_t gq = gp + (uint64_t)k;
    uint32_t wi = (uint32_t)(gp >> 4), wq = (uint32_t)(gq >> 4);
    wi = wi < wmax ? wi : wmax; /* over-reads behind the last sequence: values never used */
    wq = wq < wmax ? wq : wmax;
    W.o0 = ntl_load4_a4(packed + wi); W.o4 = packed[wi + 4];
    W.i0 = ntl_load4_a4(packed + wq); W.i4 = packed[wq + 4];
    W.ao = 2u * ((uint32_t)gp & 15u); W.ai = 2u * ((uint32_t)gq & 15u);
    return W;
}

/* A staged key carries the step it was made in in its six low bits: k' = (key & ~63) | t.  Two such words order their k-mers' hashes
   when they are more than SKW_NEAR apart: k'_j - k'_i > SKW_NEAR  =>  key_j - key_i > SKW_NEAR - 126 >= SK2_NEAR  =>  h0_i < h0_j
   ("exact" in the header of this file); closer pairs are the near ties that give a strip up -- 2^-23 per window instead of 2^-29. */
#define SKW_CHUNK 4u /* strips a wavefront takes from its XCD's counter at a time */
#define SKW_NEAR 131u
static_assert(SKW_NEAR >= 126u + SK2_NEAR, "see above");

/* rolling step T (the k-mer at position T of the lane's block from the one before it) and its candidate test */
template <int T>
__device__ __forceinline__ void skw_step(uint32_t &fx, uint32_t &ry, uint32_t &fd, const uint2 sd, uint32_t tm1, uint32_t low6, uint32_t *&dst)
{
    fx = ntl_alignbit(fx, fd, 31) ^ sd.x; /* srol1 on the ring: (fx << 1) | ring bit 30, which is the top bit of fd = fx << 1 */
    const uint32_t a = ry ^ sd.y;
    ry = ntl_alignbit(a >> 1, a, 1);      /* sror1: ring bit 0 (bit 1 of a) -> bit 31 */
    fd = ntl_double(fx); /* fx << 1 as fx + fx: v_add_u32 issues in 2.4 cycles, v_lshlrev_b32 in 4.4 (profiles/valu_cycles.json) */
    const uint32_t key = fd + ry;
    if (key <= tm1) ntl_lds_push_tagged<T>(dst, low6, key);
}

template <int T0, int... J>
__device__ __forceinline__ void skw_steps(std::integer_sequence<int, J...>, uint32_t &fx, uint32_t &ry, uint32_t &fd, const uint2 (&sd)[8], uint32_t tm1,
                                          uint32_t low6, uint32_t *&dst)
{
    (skw_step<T0 + J>(fx, ry, fd, sd[J], tm1, low6, dst), ...);
}

/* wavefronts per SIMD a shape is compiled for = what its LDS lets a CU hold (the register budget follows from it): the shapes of the
   large windows 32 wavefronts per CU; <8,15,6> three workgroups of 40 KB, <8,19,8> three of 48 KB: 24; <4,19,8> five of 28 KB: 20 */
template <int WAVES, int S>
constexpr int skw_waves_per_simd() { return S >= 15 ? (WAVES >= 8 ? 6 : 5) : (WAVES >= 8 ? 8 : 7); }

template <int WAVES, int S, int ROUNDS>
__global__ __launch_bounds__(64 * WAVES, (skw_waves_per_simd<WAVES, S>())) void sketch_wave_kernel(Sketch2Args B)
{
    constexpr int C = 64;
    constexpr uint32_t SLOTS = 64u * (uint32_t)S;      /* words of a wavefront's key array: staging slots, then the list's keys in place */
    constexpr uint32_t NLIST = 64u * (uint32_t)ROUNDS; /* list entries: [3] left sentinel, [4 .. n + 3] candidates, [n + 4] right sentinel, copies around them */
    constexpr uint32_t CAP = NLIST - 8u;               /* candidates the list holds */
    static_assert(NLIST <= SLOTS, "the list's keys live in the staging slots");
    static_assert(S & 1, "odd S: the lanes' slots start S words apart, and a ds_write_b32 of 32 lanes is conflict-free only for odd S");
    __shared__ __attribute__((aligned(16))) uint2 s_g4k[1024];
    __shared__ __attribute__((aligned(16))) uint32_t s_roll[32];
    /* the list's positions are words at a constant distance KP from their keys (one address register serves both), in the upper part
       of the same array: once the staged keys are in registers only the list lives there */
    constexpr uint32_t KP = NLIST;
    constexpr uint32_t NSCAN = 2u * NLIST + (CAP + 1u) / 2u; /* the list, and the scanners' indices (16 bits each) behind it */
    constexpr uint32_t NWORDS = SLOTS + (uint32_t)(C - S) > NSCAN ? SLOTS + (uint32_t)(C - S) : NSCAN; /* the slots + what lane 63 may write beyond them */
    __shared__ __attribute__((aligned(16))) uint32_t s_keys[WAVES][NWORDS];

    const SketchArgs &A = B.A;
    const int tid = threadIdx.x, L = tid & 63;
    const uint32_t wv = ntl_readfirstlane((uint32_t)tid >> 6);
    for (int i = tid; i < 512; i += 64 * WAVES) ((uint4 *)s_g4k)[i] = ((const uint4 *)B.g4k)[i];
    if (tid < 16) { s_roll[2 * tid] = (uint32_t)(A.roll_tab[tid][0] >> 33); s_roll[2 * tid + 1] = (uint32_t)(A.roll_tab[tid][1] >> 32); }

    /* The strips of this wavefront: XCD x (workgroups x, x + 8, ...: they share an L2) takes the x-th eighth of the strips, and its
       wavefronts take CHUNKS of SKW_CHUNK consecutive strips from a counter as they go (neighbouring strips share their halo bases
       and strip-table lines).  Taken, not dealt out in advance: beside the lookup and map kernels of the other stream a part of the
       workgroups starts late, and with fixed shares the launch would end when the last of them has worked off a full share. */
    const uint32_t per_xcd = (A.nstrips + 7u) >> 3;
    const uint32_t lo = (blockIdx.x & 7u) * per_xcd;
    const uint32_t end = lo + per_xcd < A.nstrips ? lo + per_xcd : A.nstrips;
    uint32_t *const counter = B.chunk_next + 16u * (blockIdx.x & 7u);
    const uint32_t wmax = (uint32_t)B.max_word - 9u;
    const int k = A.G.k;
    /* s0 is the strip at work, s1 and s2 the two behind it (0xFFFFFFFF: none); `chunk` holds the chunk s2 walks through and
       `ahead` -- in lane 0's register until it is needed -- the number of the one after it */
    uint32_t ahead = 0;
    if (L == 0) ahead = atomicAdd(counter, 2u);
    uint32_t chunk = lo + SKW_CHUNK * ntl_readfirstlane(ahead), sub = 0;
    ahead = chunk + SKW_CHUNK; /* (the first take covered two chunks) */
    /* chunk_budget != 0: a wavefront takes that many chunks (>= 2) and ends -- workgroups that live for a fraction of the launch, in a
       grid of as many as it takes: beside the other stream's kernels (which have the higher priority and a bounded number of resident
       workgroups of their own) they fill whatever a CU has free, all of it while that stream is idle (ntl_hip.hip, launch_fast_r0) */
    uint32_t left = B.chunk_budget ? B.chunk_budget - 2u : 0xFFFFFFFFu;
    auto next_strip = [&]() -> uint32_t {
        if (sub == SKW_CHUNK) {
            chunk = ntl_readfirstlane(ahead);
            sub = 0;
            if (left) {
                left--;
                if (L == 0) ahead = lo + SKW_CHUNK * atomicAdd(counter, 1u); /* asked for a whole chunk before it is needed */
            } else ahead = 0xFFFFFFF0u; /* no strip */
        }
        const uint32_t s = chunk + sub;
        sub++;
        return s < end ? s : 0xFFFFFFFFu;
    };
    /* a strip's table entry is the same in every lane: it is loaded as a vector (see above) and kept as four SCALARS from the iteration
       after the one that asked for it -- one entry in flight in vector registers instead of three held there (round 5: the kernel's
       registers are what the other stream's kernels do not get) */
    auto uniform4 = [](const uint4 v) {
        return make_uint4(ntl_readfirstlane(v.x), ntl_readfirstlane(v.y), ntl_readfirstlane(v.z), ntl_readfirstlane(v.w));
    };
    uint32_t s0 = next_strip(), s1 = next_strip();
    uint4 I = uniform4(skw_strip_load(A.strip_lite, s0, end));
    uint4 I1v = skw_strip_load(A.strip_lite, s1, end);
    SkwWords W = skw_words_load(A.T.packed, I, L, k, wmax);
    __syncthreads(); /* the tables are in LDS; the only workgroup barrier of the kernel */

    uint32_t *const keys = &s_keys[wv][0];
    uint32_t *const pos = keys + KP;
    const uint32_t w = (uint32_t)A.G.w;
    while (s0 != 0xFFFFFFFFu) {
        /* two strips ahead: the table entry; one strip ahead: the base words */
        const uint32_t s2 = next_strip();
        const uint4 I2v = skw_strip_load(A.strip_lite, s2, end);
        const uint4 Ic = I;
        const SkwWords Wc = W;
        const uint32_t strip = s0;
        I = uniform4(I1v); I1v = I2v; /* (I: the next strip's entry, asked for a whole strip ago; its base words are asked for behind the rolling loop, below) */
        s0 = s1; s1 = s2;
        const uint32_t hi_raw = Ic.z;
        const uint32_t hi = hi_raw & 0xFFFFu;
        if (hi == 0u) { /* past the last strip, or a strip that crosses non-ACGT runs: sketch_mask_kernel<.., MULTI = true> */
            W = skw_words_load(A.T.packed, I, L, k, wmax);
            continue;
        }
        const bool has_w0 = (hi_raw & STRIP_FIRST) == 0u; /* element 0 is a k-mer of the sequence, window 0 the previous strip's last */

        /* ---- the lane's bases: 64 that leave (so) and the 64 that enter (si), k bases further on ---- */
        uint32_t so[4], si[4];
        so[0] = ntl_alignbit(Wc.o0.y, Wc.o0.x, Wc.ao); so[1] = ntl_alignbit(Wc.o0.z, Wc.o0.y, Wc.ao); so[2] = ntl_alignbit(Wc.o0.w, Wc.o0.z, Wc.ao); so[3] = ntl_alignbit(Wc.o4, Wc.o0.w, Wc.ao);
        si[0] = ntl_alignbit(Wc.i0.y, Wc.i0.x, Wc.ai); si[1] = ntl_alignbit(Wc.i0.z, Wc.i0.y, Wc.ai); si[2] = ntl_alignbit(Wc.i0.w, Wc.i0.z, Wc.ai); si[3] = ntl_alignbit(Wc.i4, Wc.i0.w, Wc.ai);

        const uint32_t Cs = ntl_readfirstlane(skw_per_lane(hi_raw)); /* k-mers per lane of this strip */
        const uint32_t lane_pos = (uint32_t)L * Cs;
        const uint32_t tm1 = lane_pos < hi ? B.thresh - 1u : 0u; /* lanes behind the sequence's last k-mer roll over whatever follows it: nothing of theirs is a candidate */
        uint32_t low6 = 63u;
        NTL_OPAQUE(low6); /* a register, not a literal per step; and not loop-invariant for the optimiser */

        /* ---- the first k-mer's rings from the lane's own bases: 16-base partial hashes out of the four-base tables ---- */
        uint32_t fx, ry;
        {
            uint32_t f = 0, u = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { /* k <= 64 */
                if (i >= B.q16) break;
                if (i) { f = ring_rotl(f, 16); u = ring_rotr(u, 16); }
                const uint32_t x = so[i];
                const uint2 g0 = s_g4k[x & 255u], g1 = s_g4k[256u + ((x >> 8) & 255u)], g2 = s_g4k[512u + ((x >> 16) & 255u)], g3 = s_g4k[768u + (x >> 24)];
                f ^= g0.x ^ g1.x ^ g2.x ^ g3.x;
                u ^= g0.y ^ g1.y ^ g2.y ^ g3.y;
            }
            if (B.r16) {
                const uint32_t r = (uint32_t)B.r16;
                if (B.q16) { f = ring_rotl(f, r); u = ring_rotr(u, r); }
                const uint32_t x = B.q16 == 0 ? so[0] : (B.q16 == 1 ? so[1] : (B.q16 == 2 ? so[2] : so[3]));
                uint2 P;
                if (B.r16 == 8) {
                    const uint2 p0 = s_g4k[512u + (x & 255u)], p1 = s_g4k[768u + ((x >> 8) & 255u)];
                    P = make_uint2(p0.x ^ p1.x, p0.y ^ p1.y);
                } else {
                    uint2 fu;
                    sk2_chunk(x, B.r16, B, fu, P); /* k % 16 not in {0, 8}: the first r bases on the plain 64-bit tables */
                }
                f ^= P.x;
                u ^= P.y;
            }
            fx = f;
            ry = u << 1; /* the tables carry the reverse strand's final rotation */
        }

        /* ---- rolling; every key is tested as it is made, candidates go to the lane's staging slots ---- */
        uint32_t *const slots = keys + (uint32_t)L * (uint32_t)S;
        uint32_t *dst = slots;
        {
            uint32_t fd = fx << 1;
            if (fd + ry <= tm1) ntl_lds_push_tagged<0>(dst, low6, fd + ry);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if ((uint32_t)(16 * m) >= Cs) break; /* (a short last strip: its lanes hold 16, 32 or 48 k-mers; the step that a block of
                                                        sixteen makes beyond them is the next lane's first k-mer, dropped below) */
                uint32_t wz[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t o2 = r < 2 ? so[m] << (3 - 2 * r) : so[m] >> (2 * r - 3);
                    const uint32_t i2 = r < 3 ? si[m] << (5 - 2 * r) : si[m] >> (2 * r - 5);
                    wz[r] = (o2 & 0x18181818u) | (i2 & 0x60606060u);
                }
                /* the seed pairs of eight steps are requested together, in front of the steps: every step ends in a predicated
                   write (a basic block of its own), and a read issued inside a step would be waited for inside it */
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    uint2 sd[8];
#pragma unroll
                    for (int b = 8 * h; b < 8 * h + 8; b++) {
                        const uint32_t off = ntl_bfe(wz[b & 3], 8u * (uint32_t)(b >> 2), 8u);
                        sd[b & 7] = ntl_lds_load2_ordered((const uint2 *)((const char *)s_roll + off));
                    }
                    if (m == 0 && h == 0) skw_steps<1>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else if (m == 0) skw_steps<9>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else if (m == 1 && h == 0) skw_steps<17>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else if (m == 1) skw_steps<25>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else if (m == 2 && h == 0) skw_steps<33>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else if (m == 2) skw_steps<41>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else if (h == 0) skw_steps<49>(std::make_integer_sequence<int, 8>(), fx, ry, fd, sd, tm1, low6, dst);
                    else skw_steps<57>(std::make_integer_sequence<int, 7>(), fx, ry, fd, sd, tm1, low6, dst);
                }
            }
        }
        /* the next strip's base words: asked for HERE, behind the rolling loop (where the registers are scarcest: twelve of them held
           across it made the compiler spill into scratch memory, five stores and six loads per strip) and in front of the list and the
           scans, which take long enough to cover the loads */
        W = skw_words_load(A.T.packed, I, L, k, wmax);
        uint32_t cnt = (uint32_t)(dst - slots);
        bool bad = cnt > (uint32_t)S; /* (what it wrote beyond its slots lies in the next lane's, or behind the array: the strip is given up) */

        /* ---- the staged candidates that are elements of the strip's windows into registers, then back in position order: the list ---- */
        if (cnt > (uint32_t)S) cnt = (uint32_t)S;
        uint32_t mine[S];
#pragma unroll
        for (int j = 0; j < S; j++) mine[j] = (uint32_t)j < cnt ? slots[j] : SK2_INF;
        uint32_t first = 0, k0 = SK2_INF;
        if (L == 0 && cnt && (mine[0] & 63u) == 0u) { first = 1; cnt--; k0 = mine[0]; } /* element 0 belongs to the windows of the previous strip only */
        k0 = ntl_readfirstlane(k0);
        if (hi < (uint32_t)(64 * C)) { /* the sequence ends inside the strip: positions >= hi are not elements (nor is a lane's step Cs) */
            uint32_t keep = 0;
#pragma unroll
            for (int j = 0; j < S; j++)
                if ((uint32_t)j >= first && (uint32_t)j < first + cnt && (mine[j] & 63u) < Cs && lane_pos + (mine[j] & 63u) < hi) keep++;
            cnt = keep;
        }
        const uint32_t incl = ntl_wave_incl_scan(cnt);
        const uint32_t total = ntl_readfirstlane((uint32_t)__shfl((int)incl, 63));
        const uint32_t at = incl - cnt + 4u - first;
        const bool over = total > CAP;
        const uint32_t n = over ? 0u : total;
        ntl_wave_sync(); /* every lane holds its candidates: the slots may be overwritten */
        if (!over) {
#pragma unroll
            for (int j = 0; j < S; j++)
                if ((uint32_t)j >= first && (uint32_t)j < first + cnt) {
                    keys[at + (uint32_t)j] = mine[j];
                    pos[at + (uint32_t)j] = lane_pos + (mine[j] & 63u);
                }
        }
        if (L < 4) {
            keys[L] = 0u; pos[L] = 0;                                       /* the left sentinel (position 0, key 0) and copies of it in front */
            keys[n + 4u + (uint32_t)L] = 0u; pos[n + 4u + (uint32_t)L] = hi; /* the right sentinel at the end of the strip's elements and copies behind it */
        }
        ntl_wave_sync();

        /* ---- round 6: who has to scan at all.  Four candidates in five are nobody's minimum, and most of them show it at once: a
           candidate with a CERTAINLY smaller key (more than SKW_NEAR below its own: the keys then order the hashes) among its four
           list neighbours on either side, all eight of them less than a window apart, lies in no window without one of the two --
           whatever else the strip holds, whatever the keys above the threshold are, however its ties fall.  One pass over the
           list, a lane per candidate, a dozen LDS reads and no loop, takes those out (71 % of them: P(the least of five on one side)
           = 1/5 each, 1/9 both); what is left -- 50 of a strip's 164 at w = 250 -- is scanned as before, in ONE round where there
           were three (the rounds cost the same whoever sits in them: the wavefront waits for its slowest lane).  The candidates
           that drop out stay in the list: they are still somebody's blockers.  The same pass makes the two checks that belong to the
           list and not to a candidate's scan: a gap of a window between two candidates, and window 0 (whose minimum is the previous
           strip's to list: it does not scan either). ---- */
        uint16_t *const sidx = (uint16_t *)(keys + 2u * NLIST); /* the scanners' list indices, behind the list (the staging slots there are in registers) */
        uint32_t ns = 0;
        if (L == 0) bad |= pos[4] - 1u >= w; /* elements 1 .. w without a candidate */
        for (uint32_t r = 0; r * 64u < n; r++) {
            const uint32_t i0 = (uint32_t)L + 64u * r;
            const bool real = i0 < n;
            const uint32_t i = real ? i0 : n - 1u;
            const uint32_t *kp = keys + i;
            const uint32_t l3 = kp[0], l2 = kp[1], l1 = kp[2], l0 = kp[3], mk = kp[4], r0 = kp[5], r1 = kp[6], r2 = kp[7], r3 = kp[8];
            const uint32_t pl4 = kp[KP], pl2 = kp[KP + 2], mp = kp[KP + 4], pn = kp[KP + 5], pr2 = kp[KP + 6], pr4 = kp[KP + 8];
            bool b = pn - mp - 1u >= w; /* a window between two candidates */
            bool scan = real;
            if (r == 0u && has_w0) {
                /* Window 0 = elements 0 .. w - 1 is the previous strip's last one, and its minimum is that strip's to list, not
                   this one's -- whichever of this strip's windows it is the minimum of as well.  The candidates in it are the
                   first entries of the list (and element 0, k0, if it is one): the least key, alone within the tolerance. */
                const bool inw0 = real && mp < w;
                uint32_t kmin = ntl_wave_min(inw0 ? mk : SK2_INF);
                kmin = kmin < k0 ? kmin : k0;
                const bool near0 = inw0 && mk <= kmin + SKW_NEAR;
                const uint32_t nn = (uint32_t)__popcll(__ballot(near0)) + (k0 != SK2_INF && k0 <= kmin + SKW_NEAR ? 1u : 0u);
                b |= nn != 1u || kmin == SK2_INF || (n > 64u && pos[67] < w); /* a near tie, no candidate, or more than this round sees */
                scan = scan && !near0;
            }
            const uint32_t sure = ntl_sub_sat(mk, SKW_NEAR); /* keys below it are certainly smaller (none, for a key within the tolerance of the sentinels' 0) */
            /* how far apart the two are: the second neighbour's position where one of the nearer two is smaller, else the fourth's (the nearer
               smaller one's own position would be a read more and decide 2 candidates of 164 more; the fourth's alone, 22 fewer) */
            const uint32_t ml2 = l0 < l1 ? l0 : l1, mr2 = r0 < r1 ? r0 : r1;
            const uint32_t pl = ml2 < sure ? pl2 : pl4, pr = mr2 < sure ? pr2 : pr4;
            const bool out = (ntl_min3(ml2, l2, l3) < sure) & (ntl_min3(mr2, r2, r3) < sure) & (pr - pl <= w);
            scan = scan && !out;
            const unsigned long long bal = __ballot(scan);
            if (scan) sidx[ns + ntl_mbcnt(bal)] = (uint16_t)i;
            ns += (uint32_t)__popcll(bal);
            if (real) bad |= b;
        }
        ntl_wave_sync(); /* the scanners' indices are other lanes' to read */

        /* ---- one lane per scanner (sketch_thresh_kernel, phase 4), four list entries per step.  Every branch is uniform over the
           wavefront: a lane whose scan has ended reads its last four entries again until the slowest lane's has (the wavefront
           pays for the slowest lane either way, and without diverging lanes there are no execution masks to juggle); a lane
           without a scanner in the last round works on the last one again and drops the answer. ---- */
        uint32_t found = 0; /* bit r: this lane's scanner of round r is a minimizer */
        /* The strip's own windows start at elements 1 .. NWO (the next strip's at its element 1 = NWO + 1 of this one), so the
           last element of an own window is V - 1: no window reaches V, as if a blocker stood there, and a candidate at or behind
           V is nobody's minimum here.  (The bitmask did not mind a bit set by two strips; the lists do: StripLists.) */
        const uint32_t V = (uint32_t)A.G.NWO + w;
        for (uint32_t r = 0; r * 64u < ns; r++) {
            const uint32_t j0 = (uint32_t)L + 64u * r;
            const bool real = j0 < ns;
            const uint32_t i = sidx[real ? j0 : ns - 1u];
            const uint32_t mk = keys[i + 4u], mp = pos[i + 4u];
            const uint32_t lim = mk + SKW_NEAR;
            bool b = lim >= B.thresh;
            uint32_t Rp = mp + w < V ? mp + w : V;
            uint32_t q0, q1, q2, q3;
            {   /* to the right: the first blocker nearer than w; ends at the right sentinel (key 0) at the latest */
                const uint32_t *kp = keys + i + 5u;
                for (;;) { /* a step only asks "any blocker among the four?" (two minima, one compare); which one is settled once, behind the loop */
                    q0 = kp[0]; q1 = kp[1]; q2 = kp[2]; q3 = kp[3];
                    const uint32_t p3 = kp[KP + 3];
                    const bool go = (ntl_min3(q0, q1, q2 < q3 ? q2 : q3) > lim) & (p3 < Rp); /* (no short circuit: the reads are issued together) */
                    if (__ballot(go) == 0ull) break;
                    kp += go ? 4 : 0;
                }
                const uint32_t fb = sk2t_first_le(q0, q1, q2, q3, lim);
                const uint32_t *ke = kp + (fb != 4u ? fb : 3u);
                const uint32_t ek = ke[0], ep = ke[KP];
                const bool take = fb != 4u && ep < Rp;
                b |= take && ek + SKW_NEAR >= mk; /* within the tolerance of the candidate's own key (the sentinel's key 0: only for keys <= SKW_NEAR) */
                Rp = take ? ep : Rp;
            }
            const int32_t need = (int32_t)Rp - (int32_t)w; /* a blocker at q < pos matters where q >= need */
            bool blocked;
            {   /* to the left: is there a blocker at or behind `need`?  Ends at the left sentinel (index 3: position 0, key 0) at the latest */
                const uint32_t *kp = keys + i; /* the chunk [i .. i + 3], nearest entry last */
                for (;;) {
                    q0 = kp[3]; q1 = kp[2]; q2 = kp[1]; q3 = kp[0];
                    const int32_t p3 = (int32_t)kp[KP];
                    const bool go = (ntl_min3(q0, q1, q2 < q3 ? q2 : q3) > lim) & (p3 >= need);
                    if (__ballot(go) == 0ull) break;
                    kp -= go ? 4 : 0;
                }
                const uint32_t fb = sk2t_first_le(q0, q1, q2, q3, lim);
                const uint32_t *ke = kp + 3u - (fb != 4u ? fb : 3u);
                const uint32_t ek = ke[0];
                blocked = fb != 4u && (int32_t)ke[KP] >= need;
                /* ... and is it certainly smaller?  (Round 6.  While every candidate scanned, a near tie of two candidates was seen from
                   its left one, whose scan to the right ends at the other; now the left one may have dropped out before the scans.) */
                b |= blocked && ek + SKW_NEAR >= mk;
            }
            if (real) {
                bad |= b;
                if (!blocked && mp < V) found |= 1u << r;
            }
        }

        /* ---- proven minimizers to the strip's list (or the global bitmask); a strip that was given up writes none and goes to the block-minima pass ---- */
        const bool flagged = B.force_redo || over || __ballot(bad) != 0ull;
        if (flagged) {
            if (L == 0) B.fb_list[atomicAdd(B.fb_count, 1u)] = strip;
        } else if (A.Ls.cnt) {
            /* the strip's list: its minimizers' positions in the sequence, in order (scanner j = L + 64 r: by rounds, then by lanes) */
            uint32_t total = 0;
            for (uint32_t r = 0; r * 64u < ns; r++) total += (uint32_t)__popcll(__ballot((found >> r) & 1u));
            uint32_t at = 0;
            if (L == 0) at = strip_list_place(A.Ls, strip, total);
            at = ntl_readfirstlane(at);
            if (at != NTL_NONE) {
                const uint32_t p0 = Ic.w;
                for (uint32_t r = 0; r * 64u < ns; r++) {
                    const bool mine_r = (found >> r) & 1u;
                    const unsigned long long bal = __ballot(mine_r);
                    if (mine_r) A.Ls.ent[at + ntl_mbcnt(bal)] = p0 + pos[(uint32_t)sidx[(uint32_t)L + 64u * r] + 4u];
                    at += (uint32_t)__popcll(bal);
                }
            }
        } else {
            const uint64_t g0 = ((uint64_t)Ic.y << 32) | Ic.x;
            while (found) {
                const uint32_t r = (uint32_t)__ffs(found) - 1u;
                found &= found - 1u;
                const uint64_t g = g0 + pos[(uint32_t)sidx[(uint32_t)L + 64u * r] + 4u];
                atomicOr(&A.mask[g >> 5], 1u << ((uint32_t)g & 31u));
            }
        }
        ntl_wave_sync(); /* the list has been read: the next strip may stage over it */
    }
}
