/*
 * Contig index, probe and per-read mapping kernels.
 *
 *   index_*      NtLink.read_minimizers (bin/ntlink_pair.py:189-211): minimizer hash ->
 *                (contig, position, strand); a hash seen twice anywhere is dropped.  Open
 *                addressing in HBM, 16-byte slots, home slot = radix of the (already hashed) key;
 *                duplicates are detected by the insert itself (CAS on the key, OR on the flags),
 *                so the table content does not depend on insertion order.
 *   probe_kernel `mx in target_mxs` (bin/ntlink_pair.py:364-367): one slot lookup per read
 *                minimizer.
 *   map_kernel   the per-read body of find_scaffold_pairs (bin/ntlink_pair.py:359-391) with
 *                get_accepted_anchor_contigs (bin/ntlink_utils.py:200-268), mark_subsumed_*
 *                (:271-294) and print_paf (bin/ntlink_paf_output.py:9-135).  One wavefront per
 *                read; hit lists staged in LDS (global scratch for reads that do not fit);
 *                stream compaction with wave ballot + mbcnt prefix sums; run-level logic
 *                (a handful of runs per read) on lane 0.
 */
#pragma once
#include "dev_common.h"
#include "sketch_kernels.h"
#include "index_common.h"

/* ------------------------------------------------------------------------------ index ---- */

/* also zeroes what the build accumulates into: the duplicate bitmap (one word per 32 slots; nslots is a
 * multiple of 32), the side slot and the kept-key counter */
__global__ void index_clear_kernel(IndexSlot *slots, uint64_t nslots, uint32_t *dup, IndexSpecial *special, unsigned long long *count)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nslots) {
        slots[i].key = NTL_INF; slots[i].pos = 0; slots[i].meta = 0;
        if ((i & 31u) == 0) dup[i >> 5] = 0;
    }
    if (i == 0) { special->cnt = 0; special->pos = 0; special->meta = 0; special->pad = 0; *count = 0; }
}

/* one atomic per workgroup: sum of per-thread counts (all threads must call) */
__device__ __forceinline__ void block_count_add(unsigned long long mine, unsigned long long *counter)
{
    __shared__ unsigned long long s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    if (mine) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(counter, s_cnt);
}

/* After the inserts, one streaming pass over the table: folds the duplicate bits into the slots, writes the
 * tag byte of every slot (four slots, one 32-bit store per thread and round) and counts the keys that are
 * kept.  nslots is a multiple of 32.  Grid-stride: launch a bounded number of workgroups. */
__global__ void index_finish_kernel(IndexSlot *slots, uint64_t nslots, const IndexSpecial *special, const uint32_t *dup,
                                    uint8_t *tags, unsigned long long *count)
{
    unsigned long long u = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nslots / 4; q += stride) {
        const uint32_t d = (dup[q >> 3] >> ((q & 7u) * 4u)) & 0xFu;
        uint32_t four = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t key = slots[4 * q + j].key;
            if (key != NTL_INF) {
                four |= (uint32_t)index_tag(key) << (8 * j);
                if ((d >> j) & 1u) slots[4 * q + j].meta |= 1u;
                else u++;
            }
        }
        reinterpret_cast<uint32_t *>(tags)[q] = four;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && special->cnt == 1) u++;
    block_count_add(u, count);
}



/* One atomic per minimizer: the compare-and-swap that claims the slot.  The winner then stores its payload
 * with a plain 64-bit write (pos and meta are one aligned word); a later arrival of the same key only sets
 * the slot's bit in `dup`, which index_finish_kernel folds into the slot -- so no arrival ever
 * read-modify-writes a payload another one may be writing. */
__global__ void index_insert_kernel(const MxRecord *mx, uint64_t n, IndexSlot *slots, int bits,
                                    IndexSpecial *special, uint32_t *dup)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const MxRecord R = mx[i];
    const uint32_t meta = ((R.meta >> 1) << 2) | ((R.meta & 1u) << 1);
    if (R.hash == NTL_INF) {
        atomicAdd(&special->cnt, 1u);
        atomicOr(&special->pos, R.pos);
        atomicOr(&special->meta, meta);
        return;
    }
    const uint64_t mask = ((uint64_t)1 << bits) - 1;
    uint64_t s = index_home(R.hash, bits);
    for (;;) {
        unsigned long long old = atomicCAS((unsigned long long *)&slots[s].key,
                                           (unsigned long long)NTL_INF, (unsigned long long)R.hash);
        if (old == NTL_INF) { /* first arrival owns the payload */
            *reinterpret_cast<uint64_t *>(&slots[s].pos) = ((uint64_t)meta << 32) | (uint64_t)R.pos;
            return;
        }
        if (old == R.hash) { atomicOr(&dup[s >> 5], 1u << (s & 31u)); return; } /* seen before: duplicate */
        s = (s + 1) & mask;
    }
}


#define PROBE_U 4 /* minimizers per thread and round: their loads are issued together (memory-level parallelism) */

/* TAGS = true: the one-byte tag of the home slot decides first (most lookups end there when few minimizers are in the index:
 * ONT reads, h ~ 0.15).  TAGS = false: the slots are read directly -- when most lookups hit (HiFi reads, h ~ 0.9) the tag is one more
 * random cache line per lookup for nothing.  The host picks by the hit fraction of the previous batch on the same index. */
template <bool TAGS>
__global__ void probe_kernel(const MxRecord *mx, uint64_t n, const IndexSlot *slots, int bits,
                             const IndexSpecial *special, Cand *cand, unsigned long long *nfound, const uint8_t *tags)
{
    unsigned long long found = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * PROBE_U;
    const uint64_t mask = ((uint64_t)1 << bits) - 1;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x * PROBE_U + threadIdx.x; i0 < n; i0 += stride) {
        uint64_t key[PROBE_U];
        IndexProbe<TAGS> pr[PROBE_U];
        bool live[PROBE_U];
#pragma unroll
        for (int u = 0; u < PROBE_U; u++) {
            const uint64_t i = i0 + (uint64_t)u * blockDim.x;
            live[u] = i < n;
            key[u] = live[u] ? ntl_stream_load(&mx[i].hash) : 0; /* streamed once: keep L2 for the tags */
        }
#pragma unroll
        for (int u = 0; u < PROBE_U; u++)
            if (live[u]) pr[u].start(key[u], slots, tags, bits); /* the first (random) loads of all four are in flight together */
#pragma unroll
        for (int u = 0; u < PROBE_U; u++) {
            if (!live[u]) continue;
            const Cand c = pr[u].finish(key[u], slots, tags, special, mask);
            ntl_stream_store((uint64_t *)&cand[i0 + (uint64_t)u * blockDim.x], (uint64_t)c.cpos | ((uint64_t)c.meta << 32));
            found += c.meta & 1u;
        }
    }
    block_count_add(found, nfound);
}

/* -------------------------------------------------------------------------------- map ---- */

struct MapParamsDev {
    int32_t k, z;
    double x;
    int32_t sensitive, repeat_filter;
};

struct MapRec { uint32_t read, ctg, n_hits, pad; uint64_t hit_off; };
struct HitRec { uint32_t ctg_pos, read_pos; uint8_t ctg_strand, read_strand, pad[2]; };
struct PafRec { uint32_t read, ctg, q_start, q_end, t_start, t_end, n_hits, strand; };

#define MAP_NT 64
/* hits / runs per read staged in LDS: template parameters of map_kernel (512/128 for dense sketches,
   256/64 -- twice the resident wavefronts -- when reads carry few minimizers; 1024/128 measured slower); larger reads
   use global scratch */
#define MAP_NHA 6    /* per-hit u32 arrays */
#define MAP_NRA 10   /* per-run u32 arrays */

struct MapArgs {
    const MxRecord *mx;
    const uint32_t *mx_off; /* [nreads+1] */
    const Cand *cand;
    const uint32_t *read_len, *ctg_len;
    uint32_t nreads;
    MapParamsDev P;
    MapRec *maps; HitRec *hits; PafRec *pafs;     /* region of read r starts at mx_off[r] */
    uint32_t *n_maps, *n_hits, *n_pafs;           /* [nreads] */
    uint32_t *scr;                                /* (MAP_NHA + MAP_NRA) arrays of scr_stride u32 */
    uint64_t scr_stride;
    uint32_t *err;
    uint32_t *over_list, *over_count;             /* reads that do not fit the LDS staging: [nreads], [1] */
    /* The sketch may still be in flight when these kernels are queued (nothing waits on the host for its size): when its
       minimizer total turns out larger than the arrays that were sized from the expected density, the records are incomplete
       and every kernel here leaves the batch alone; the host makes the sketch again and queues the map a second time. */
    const uint32_t *mx_total;                     /* NULL: the count was known when the call was made */
    uint32_t mx_cap;
};

__device__ __forceinline__ bool map_sketch_overflowed(const MapArgs &A) { return A.mx_total && *A.mx_total > A.mx_cap; }

struct HitArr { uint32_t *ctg, *cpos, *rpos, *fl, *run, *ord; };
struct RunArr { uint32_t *start, *ctg, *leader, *flag, *cnt, *mn, *mni, *mx, *mxi, *last; };

#define HF_CS 1u   /* contig strand */
#define HF_RS 2u   /* read strand */
#define HF_KEEP 4u
#define RF_NOISY 1u
#define RF_SUB 2u
#define AX_DUP 1u
#define AX_FILT 2u
#define AX_BRK 4u

/* stable in-place compaction of the hits whose HF_KEEP bit is set; returns the new count */
__device__ __forceinline__ uint32_t map_compact(HitArr H, uint32_t n)
{
    const uint32_t lane = threadIdx.x;
    uint32_t m = 0;
    for (uint32_t c = 0; c < n; c += MAP_NT) {
        const uint32_t i = c + lane;
        uint32_t a = 0, b = 0, d = 0, f = 0;
        if (i < n) { a = H.ctg[i]; b = H.cpos[i]; d = H.rpos[i]; f = H.fl[i]; }
        const bool keep = i < n && (f & HF_KEEP);
        const unsigned long long bal = __ballot(keep);
        __syncthreads(); /* every lane holds its element before anything is overwritten */
        if (keep) {
            const uint32_t o = m + ntl_mbcnt(bal);
            H.ctg[o] = a; H.cpos[o] = b; H.rpos[o] = d; H.fl[o] = f & ~HF_KEEP;
        }
        m += (uint32_t)__popcll(bal);
    }
    __syncthreads();
    return m;
}

/* run id of every hit (consecutive hits on one contig, itertools.groupby); returns #runs */
__device__ __forceinline__ uint32_t map_number_runs(HitArr H, uint32_t n)
{
    const uint32_t lane = threadIdx.x;
    uint32_t R = 0;
    for (uint32_t c = 0; c < n; c += MAP_NT) {
        const uint32_t i = c + lane;
        const bool isb = i < n && (i == 0 || H.ctg[i] != H.ctg[i - 1]);
        const unsigned long long bal = __ballot(isb);
        if (i < n) H.run[i] = R + ntl_mbcnt(bal) + (isb ? 1u : 0u) - 1u;
        R += (uint32_t)__popcll(bal);
    }
    __syncthreads();
    return R;
}

__device__ __forceinline__ void map_fill_runs(HitArr H, uint32_t n, RunArr RU, uint32_t R)
{
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < n; i += MAP_NT)
        if (i == 0 || H.ctg[i] != H.ctg[i - 1]) { RU.start[H.run[i]] = i; RU.ctg[H.run[i]] = H.ctg[i]; }
    __syncthreads();
    /* leader = first run on the same contig */
    for (uint32_t r = lane; r < R; r += MAP_NT) {
        uint32_t ld = r;
        const uint32_t c = RU.ctg[r];
        for (uint32_t j = 0; j < r; j++)
            if (RU.ctg[j] == c) { ld = j; break; }
        RU.leader[r] = ld;
        RU.flag[r] = 0;
    }
    __syncthreads();
}

/* One read on one wavefront.  GLOBAL = false: hits and runs live in the workgroup's LDS arrays -- the pointers are set from
 * them unconditionally, so every access is a DS instruction; a read that does not fit (more than MAP_CAPH hits or MAP_CAPR
 * runs) is put on the overflow list and left alone.  GLOBAL = true (map_overflow_kernel): the same code on the read's region
 * of the global scratch arrays. */
template <int MAP_CAPH, int MAP_CAPR, bool GLOBAL>
__device__ __forceinline__ void map_read(const MapArgs &A, const uint32_t r, uint32_t (*s_hit)[MAP_CAPH], uint32_t (*s_run)[MAP_CAPR])
{
    const uint32_t lane = threadIdx.x;
    const uint32_t m0 = A.mx_off[r], m1 = A.mx_off[r + 1], nmx = m1 - m0;
    const MapParamsDev P = A.P;
    uint32_t R = 0, n = 0, np = 0;

    /* candidates found in the index: counted only when the read could overflow the LDS staging
       (hits <= minimizers, so short sketches need no counting pass) */
    uint32_t nc = nmx;
    if (!GLOBAL && nmx > MAP_CAPH) {
        nc = 0;
        for (uint32_t c = 0; c < nmx; c += MAP_NT) {
            const uint32_t i = c + lane;
            const bool v = i < nmx && (A.cand[m0 + i].meta & 1u);
            nc += (uint32_t)__popcll(__ballot(v));
        }
    }
    HitArr H;
    if (!GLOBAL) {
        if (nc > MAP_CAPH) { /* wave-uniform */
            if (lane == 0) A.over_list[atomicAdd(A.over_count, 1u)] = r;
            return;
        }
        H.ctg = s_hit[0]; H.cpos = s_hit[1]; H.rpos = s_hit[2]; H.fl = s_hit[3]; H.run = s_hit[4]; H.ord = s_hit[5];
    } else {
        uint32_t *g = A.scr + m0;
        H.ctg = g; H.cpos = g + A.scr_stride; H.rpos = g + 2 * A.scr_stride; H.fl = g + 3 * A.scr_stride;
        H.run = g + 4 * A.scr_stride; H.ord = g + 5 * A.scr_stride;
    }
    RunArr RU;
    RU.start = nullptr;

    if (nc == 0) goto done;

    /* bin/ntlink_pair.py:364-367 hits in read order; bin/ntlink_utils.py:206 contig length >= z */
    for (uint32_t c = 0; c < nmx; c += MAP_NT) {
        const uint32_t i = c + lane;
        Cand cd;
        cd.cpos = 0; cd.meta = 0;
        MxRecord mr;
        mr.pos = 0; mr.meta = 0; mr.hash = 0;
        if (i < nmx) { cd = A.cand[m0 + i]; mr = A.mx[m0 + i]; }
        bool v = (cd.meta & 1u) != 0;
        if (v && !P.repeat_filter) v = (int64_t)A.ctg_len[cd.meta >> 2] >= (int64_t)P.z;
        const unsigned long long bal = __ballot(v);
        if (v) {
            const uint32_t o = n + ntl_mbcnt(bal);
            H.ctg[o] = cd.meta >> 2; H.cpos[o] = cd.cpos; H.rpos[o] = mr.pos;
            H.fl[o] = ((cd.meta >> 1) & 1u) | ((mr.meta & 1u) << 1);
        }
        n += (uint32_t)__popcll(bal);
    }
    __syncthreads();

    if (P.repeat_filter) {
        /* :368-374 drop minimizers that occur more than once among the read's hits, then z */
        for (uint32_t i = lane; i < n; i += MAP_NT) {
            const uint32_t c = H.ctg[i], p = H.cpos[i];
            bool dup = false;
            for (uint32_t j = 0; j < n; j++)
                if (j != i && H.ctg[j] == c && H.cpos[j] == p) { dup = true; break; }
            const bool keep = !dup && (int64_t)A.ctg_len[c] >= (int64_t)P.z;
            H.fl[i] = (H.fl[i] & ~HF_KEEP) | (keep ? HF_KEEP : 0u);
        }
        __syncthreads();
        n = map_compact(H, n);
    }
    if (n == 0) goto done;

    R = map_number_runs(H, n);
    if (!GLOBAL) {
        if (R > MAP_CAPR) { /* wave-uniform; nothing has been written for this read yet */
            if (lane == 0) A.over_list[atomicAdd(A.over_count, 1u)] = r;
            return;
        }
        RU.start = s_run[0]; RU.ctg = s_run[1]; RU.leader = s_run[2]; RU.flag = s_run[3]; RU.cnt = s_run[4];
        RU.mn = s_run[5]; RU.mni = s_run[6]; RU.mx = s_run[7]; RU.mxi = s_run[8]; RU.last = s_run[9];
    } else {
        uint32_t *g = A.scr + MAP_NHA * A.scr_stride + m0;
        RU.start = g; RU.ctg = g + A.scr_stride; RU.leader = g + 2 * A.scr_stride; RU.flag = g + 3 * A.scr_stride;
        RU.cnt = g + 4 * A.scr_stride; RU.mn = g + 5 * A.scr_stride; RU.mni = g + 6 * A.scr_stride;
        RU.mx = g + 7 * A.scr_stride; RU.mxi = g + 8 * A.scr_stride; RU.last = g + 9 * A.scr_stride;
    }
    map_fill_runs(H, n, RU, R);

    /* bin/ntlink_utils.py:217-234 noisy contigs: span on the contig longer than the read allows */
    {
        if (R <= 8 && n >= 128) {
            /* few, long runs (HiFi: hundreds of hits on one contig): the lanes share each run and
               lane 0 merges the 64 partial results; H.ord is free until the PAF stage and holds >= 256 entries (MAP_CAPH in LDS;
               a read is on the global scratch only with more than MAP_CAPH candidates or more than MAP_CAPR > 8 runs) */
            uint32_t *tmp = H.ord;
            for (uint32_t q = 0; q < R; q++) {
                const uint32_t s = RU.start[q], e = q + 1 < R ? RU.start[q + 1] : n;
                uint32_t mn = 0, mni = NTL_NONE, mx = 0, mxi = NTL_NONE;
                for (uint32_t i = s + lane; i < e; i += MAP_NT) { /* i ascends: strict compares keep the first */
                    const uint32_t p = H.cpos[i];
                    if (mni == NTL_NONE || p < mn) { mn = p; mni = i; }
                    if (mxi == NTL_NONE || p > mx) { mx = p; mxi = i; }
                }
                __syncthreads(); /* tmp of the previous run has been consumed */
                tmp[lane] = mn; tmp[MAP_NT + lane] = mni; tmp[2 * MAP_NT + lane] = mx; tmp[3 * MAP_NT + lane] = mxi;
                __syncthreads();
                if (lane == 0) {
                    uint32_t bmn = 0, bmni = NTL_NONE, bmx = 0, bmxi = NTL_NONE;
                    for (uint32_t l = 0; l < MAP_NT; l++) {
                        const uint32_t a = tmp[l], ai = tmp[MAP_NT + l], b = tmp[2 * MAP_NT + l], bi = tmp[3 * MAP_NT + l];
                        if (ai != NTL_NONE && (bmni == NTL_NONE || a < bmn || (a == bmn && ai < bmni))) { bmn = a; bmni = ai; }
                        if (bi != NTL_NONE && (bmxi == NTL_NONE || b > bmx || (b == bmx && bi < bmxi))) { bmx = b; bmxi = bi; }
                    }
                    RU.cnt[q] = e - s; RU.mn[q] = bmn; RU.mni[q] = bmni; RU.mx[q] = bmx; RU.mxi[q] = bmxi;
                }
            }
        } else
        for (uint32_t q = lane; q < R; q += MAP_NT) {
            const uint32_t s = RU.start[q], e = q + 1 < R ? RU.start[q + 1] : n;
            uint32_t mn = H.cpos[s], mni = s, mx = mn, mxi = s;
            for (uint32_t i = s + 1; i < e; i++) {
                const uint32_t p = H.cpos[i];
                if (p < mn) { mn = p; mni = i; }  /* first argmin */
                if (p > mx) { mx = p; mxi = i; }  /* first argmax */
            }
            RU.cnt[q] = e - s; RU.mn[q] = mn; RU.mni[q] = mni; RU.mx[q] = mx; RU.mxi[q] = mxi;
        }
        __syncthreads();
        if (lane == 0) {
            for (uint32_t q = 0; q < R; q++) {
                const uint32_t ld = RU.leader[q];
                if (ld == q) continue;
                RU.cnt[ld] += RU.cnt[q];
                if (RU.mn[q] < RU.mn[ld]) { RU.mn[ld] = RU.mn[q]; RU.mni[ld] = RU.mni[q]; }
                if (RU.mx[q] > RU.mx[ld]) { RU.mx[ld] = RU.mx[q]; RU.mxi[ld] = RU.mxi[q]; }
            }
        }
        __syncthreads();
        const int64_t rl = (int64_t)A.read_len[r];
        bool my_noisy = false;
        for (uint32_t q = lane; q < R; q += MAP_NT) {
            if (RU.leader[q] != q || RU.cnt[q] < 2) continue;
            const int64_t span = (int64_t)RU.mx[q] - (int64_t)RU.mn[q];
            bool noisy;
            if (P.x == 0.0) {
                noisy = span > rl + P.k;
            } else {
                int64_t rd = (int64_t)H.rpos[RU.mxi[q]] - (int64_t)H.rpos[RU.mni[q]];
                if (rd < 0) rd = -rd;
                const double a = (double)(rl + P.k);
                const double b = ntl_mul_add_rn(P.x, (double)rd, (double)P.k);
                const double thr = b < a ? b : a;
                noisy = (double)span > thr;
            }
            if (noisy) { RU.flag[q] |= RF_NOISY; my_noisy = true; }
        }
        const bool any_noisy = __ballot(my_noisy) != 0ull;
        __syncthreads();
        if (any_noisy) {
            for (uint32_t i = lane; i < n; i += MAP_NT) {
                const bool keep = !(RU.flag[RU.leader[H.run[i]]] & RF_NOISY);
                H.fl[i] = (H.fl[i] & ~HF_KEEP) | (keep ? HF_KEEP : 0u);
            }
            __syncthreads();
            n = map_compact(H, n);
            if (n == 0) { R = 0; goto done; }
            R = map_number_runs(H, n);
            map_fill_runs(H, n, RU, R);
        }
    }

    /* bin/ntlink_utils.py:246-258 subsumed runs */
    {
        if (lane == 0) {
            for (uint32_t q = 0; q < R; q++) RU.last[RU.leader[q]] = q;
            bool any = false;
            if (!P.sensitive) {
                /* mark_subsumed_specific :280-294: a run strictly inside (first, last) of any contig
                   marks its whole contig */
                uint32_t pm = 0;
                for (uint32_t q = 0; q < R; q++) {
                    if (pm > q) { RU.flag[RU.leader[q]] |= RF_SUB; any = true; }
                    if (RU.leader[q] == q && RU.last[q] > pm) pm = RU.last[q];
                }
                if (any)
                    for (uint32_t q = 0; q < R; q++)
                        if (RU.flag[RU.leader[q]] & RF_SUB) RU.flag[q] |= RF_SUB;
            } else {
                /* mark_subsumed_sensitive :271-278: runs strictly between two occurrences of
                   ANOTHER contig */
                uint32_t m1 = 0, l1 = NTL_NONE, m2 = 0;
                for (uint32_t q = 0; q < R; q++) {
                    const uint32_t ld = RU.leader[q];
                    const uint32_t cover = l1 != ld ? m1 : m2;
                    if (cover > q) { RU.flag[q] |= RF_SUB; any = true; }
                    if (ld == q) {
                        const uint32_t v = RU.last[q];
                        if (v > m1) { m2 = m1; m1 = v; l1 = q; }
                        else if (v > m2) m2 = v;
                    }
                }
            }
            RU.cnt[0] = any ? 1u : 0u;
        }
        __syncthreads();
        const bool any_sub = RU.cnt[0] != 0;
        __syncthreads();
        if (any_sub) {
            for (uint32_t i = lane; i < n; i += MAP_NT) {
                const bool keep = !(RU.flag[H.run[i]] & RF_SUB);
                H.fl[i] = (H.fl[i] & ~HF_KEEP) | (keep ? HF_KEEP : 0u);
            }
            __syncthreads();
            n = map_compact(H, n);
            if (n == 0) { R = 0; goto done; }
            R = map_number_runs(H, n);
            map_fill_runs(H, n, RU, R);
        }
        /* bin/ntlink_utils.py:262-266: every accepted contig appears once */
        bool bad = false;
        for (uint32_t q = lane; q < R; q += MAP_NT) bad |= RU.leader[q] != q;
        if (__ballot(bad) != 0ull && lane == 0) atomicOr(A.err, 1u);
    }

    /* bin/ntlink_pair.py:382-388 one record per accepted contig, hits in read order */
    for (uint32_t q = lane; q < R; q += MAP_NT) {
        const uint32_t s = RU.start[q], e = q + 1 < R ? RU.start[q + 1] : n;
        MapRec M;
        M.read = r; M.ctg = RU.ctg[q]; M.n_hits = e - s; M.pad = 0; M.hit_off = s;
        A.maps[m0 + q] = M;
    }
    for (uint32_t i = lane; i < n; i += MAP_NT) {
        HitRec h;
        const uint32_t f = H.fl[i];
        h.ctg_pos = H.cpos[i]; h.read_pos = H.rpos[i];
        h.ctg_strand = (uint8_t)(f & 1u); h.read_strand = (uint8_t)((f >> 1) & 1u);
        h.pad[0] = h.pad[1] = 0;
        A.hits[m0 + i] = h;
    }

    /* bin/ntlink_paf_output.py:103-135 */
    for (uint32_t q = 0; q < R; q++) {
        const uint32_t s = RU.start[q], e = q + 1 < R ? RU.start[q + 1] : n;
        const uint32_t m = e - s;
        const uint32_t ctg = RU.ctg[q];
        /* :95-101 read order == (ctg_pos, read_pos) order, or its exact reverse */
        bool nd = true, sd = true;
        uint32_t same = 0;
        for (uint32_t c = s; c < e; c += MAP_NT) {
            const uint32_t i = c + lane;
            bool x1 = true, x2 = true, sm = false;
            if (i < e) {
                const uint32_t f = H.fl[i];
                sm = (f & 1u) == ((f >> 1) & 1u);
                if (i + 1 < e) { x1 = H.cpos[i] <= H.cpos[i + 1]; x2 = H.cpos[i] > H.cpos[i + 1]; }
            }
            nd = nd && (__ballot(!x1) == 0ull);
            sd = sd && (__ballot(!x2) == 0ull);
            same += (uint32_t)__popcll(__ballot(sm));
        }
        if (m == 1 || nd || sd) {
            if (lane == 0) {
                const uint32_t a = nd ? s : e - 1, b = nd ? e - 1 : s; /* first / last in ctg_pos order */
                PafRec p;
                p.read = r; p.ctg = ctg; p.n_hits = m; p.strand = 2 * same >= m ? 1u : 0u;
                const uint32_t ca = H.cpos[a], cb = H.cpos[b], ra = H.rpos[a], rb = H.rpos[b];
                p.t_start = ca < cb ? ca : cb; p.t_end = (ca > cb ? ca : cb) + (uint32_t)P.k;
                p.q_start = ra < rb ? ra : rb; p.q_end = (ra > rb ? ra : rb) + (uint32_t)P.k;
                A.pafs[m0 + np] = p;
            }
            np++;
            continue;
        }
        /* general case: order by (ctg_pos, read_pos) by rank counting */
        uint32_t *aux = H.run; /* run ids are no longer needed: reuse as per-sorted-index flags */
        __syncthreads();
        for (uint32_t i = s + lane; i < e; i += MAP_NT) {
            const uint32_t cp = H.cpos[i], rp = H.rpos[i];
            uint32_t rk = 0;
            for (uint32_t j = s; j < e; j++) {
                const uint32_t cj = H.cpos[j];
                rk += (cj < cp || (cj == cp && H.rpos[j] < rp)) ? 1u : 0u;
            }
            H.ord[s + rk] = i;
        }
        __syncthreads();
        /* :60-93 transitions between sorted neighbours, duplicate contig positions */
        const uint32_t nt = m - 1;
        uint32_t cnt_incr = 0, cnt_decr = 0;
        for (uint32_t c = 0; c < m; c += MAP_NT) {
            const uint32_t t = c + lane;
            bool inc = false, dec = false;
            if (t < m) {
                const uint32_t cp = H.cpos[H.ord[s + t]];
                const bool dup = (t > 0 && H.cpos[H.ord[s + t - 1]] == cp) || (t + 1 < m && H.cpos[H.ord[s + t + 1]] == cp);
                aux[s + t] = dup ? AX_DUP : 0u;
                if (t < nt) {
                    const uint32_t a = H.rpos[H.ord[s + t]], b = H.rpos[H.ord[s + t + 1]];
                    inc = a <= b; dec = a >= b;
                }
            }
            cnt_incr += (uint32_t)__popcll(__ballot(inc));
            cnt_decr += (uint32_t)__popcll(__ballot(dec));
        }
        __syncthreads();
        int mode; /* 0 = one block, 1 = increasing, 2 = decreasing, 3 = no PAF line */
        if (cnt_incr == nt || cnt_decr == nt) mode = 0;
        else if (4ull * cnt_incr >= 3ull * nt) mode = 1;
        else if (4ull * (nt - cnt_incr) >= 3ull * nt) mode = 2;
        else mode = 3;
        if (mode == 3) continue;
        if (mode != 0) {
            /* :34-58 filter single inconsistent minimizers, break at larger problems */
            const bool incr = mode == 1;
            for (uint32_t t = lane; t < nt; t += MAP_NT) {
                const uint32_t a = H.rpos[H.ord[s + t]], b = H.rpos[H.ord[s + t + 1]];
                const bool tr = incr ? a <= b : a >= b;
                if (tr) continue;
                const bool d0 = aux[s + t] & AX_DUP, d1 = aux[s + t + 1] & AX_DUP;
                if (d0 || d1) continue;
                if (t + 2 >= nt) { atomicOr(&aux[s + t + 1], AX_BRK); continue; }
                const uint32_t c2 = H.rpos[H.ord[s + t + 2]];
                const bool d2 = (aux[s + t + 2] & AX_DUP) != 0;
                if (d2 || (incr ? a <= c2 : a >= c2)) { atomicOr(&aux[s + t + 1], AX_FILT); continue; }
                if (t > 0) {
                    const uint32_t cm = H.rpos[H.ord[s + t - 1]];
                    const bool dm = (aux[s + t - 1] & AX_DUP) != 0;
                    if (dm || (incr ? cm <= b : cm >= b)) { atomicOr(&aux[s + t], AX_FILT); continue; }
                }
                atomicOr(&aux[s + t + 1], AX_BRK);
            }
            __syncthreads();
        }
        /* :18-32 blocks, then :114-135 one line per block */
        if (lane == 0) {
            uint32_t first = 0, last = 0, cnt = 0, sm = 0, k_np = np;
            for (uint32_t t = 0; t <= m; t++) {
                const uint32_t ax = t < m ? aux[s + t] : AX_BRK;
                if (t < m && (ax & AX_FILT)) continue;
                if (t == m || (ax & AX_BRK)) {
                    if (cnt) {
                        const uint32_t ia = H.ord[s + first], ib = H.ord[s + last];
                        PafRec p;
                        p.read = r; p.ctg = ctg; p.n_hits = cnt; p.strand = 2 * sm >= cnt ? 1u : 0u;
                        const uint32_t ca = H.cpos[ia], cb = H.cpos[ib], ra = H.rpos[ia], rb = H.rpos[ib];
                        p.t_start = ca < cb ? ca : cb; p.t_end = (ca > cb ? ca : cb) + (uint32_t)P.k;
                        p.q_start = ra < rb ? ra : rb; p.q_end = (ra > rb ? ra : rb) + (uint32_t)P.k;
                        A.pafs[m0 + k_np++] = p;
                    }
                    cnt = 0; sm = 0;
                    if (t == m) break;
                }
                if (!cnt) first = t;
                last = t; cnt++;
                const uint32_t f = H.fl[H.ord[s + t]];
                sm += (f & 1u) == ((f >> 1) & 1u) ? 1u : 0u;
            }
            aux[s] = k_np - np; /* publish the number of blocks */
        }
        __syncthreads();
        np += aux[s];
        __syncthreads();
    }

done:
    if (lane == 0) { A.n_maps[r] = R; A.n_hits[r] = n; A.n_pafs[r] = np; }
}

template <int MAP_CAPH, int MAP_CAPR>
__global__ __launch_bounds__(MAP_NT) void map_kernel(MapArgs A)
{
    __shared__ uint32_t s_hit[MAP_NHA][MAP_CAPH];
    __shared__ uint32_t s_run[MAP_NRA][MAP_CAPR];
    if (map_sketch_overflowed(A)) return;
    map_read<MAP_CAPH, MAP_CAPR, false>(A, blockIdx.x, s_hit, s_run);
}

/* the reads map_kernel could not stage in LDS, on their regions of the global scratch arrays */
__global__ __launch_bounds__(MAP_NT) void map_overflow_kernel(MapArgs A)
{
    if (map_sketch_overflowed(A)) return;
    const uint32_t n = *A.over_count;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        map_read<1, 1, true>(A, A.over_list[i], nullptr, nullptr);
        __syncthreads();
    }
}

/* ---------------------------------------------------------------------------- gather ------ */

/* dense, read-ordered result arrays from the per-read regions */
__global__ void map_gather_kernel(MapArgs A, const uint32_t *off_maps, const uint32_t *off_hits,
                                  const uint32_t *off_pafs, MapRec *d_maps, HitRec *d_hits, PafRec *d_pafs)
{
    if (map_sketch_overflowed(A)) return;
    const uint32_t r = blockIdx.x;
    const uint32_t m0 = A.mx_off[r];
    const uint32_t nm = A.n_maps[r], nh = A.n_hits[r], npf = A.n_pafs[r];
    const uint32_t om = off_maps[r], oh = off_hits[r], op = off_pafs[r];
    for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) {
        MapRec M = A.maps[m0 + i];
        M.hit_off += oh;
        d_maps[om + i] = M;
    }
    for (uint32_t i = threadIdx.x; i < nh; i += blockDim.x) d_hits[oh + i] = A.hits[m0 + i];
    for (uint32_t i = threadIdx.x; i < npf; i += blockDim.x) d_pafs[op + i] = A.pafs[m0 + i];
}
