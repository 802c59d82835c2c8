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
                             const IndexSpecial *special, Cand *cand, uint32_t *rpos, unsigned long long *nfound, const uint8_t *tags)
{
    unsigned long long found = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * PROBE_U;
    const uint64_t mask = ((uint64_t)1 << bits) - 1;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x * PROBE_U + threadIdx.x; i0 < n; i0 += stride) {
        uint64_t key[PROBE_U], pm[PROBE_U]; /* pm: the record's position (low word) and strand | sequence << 1 (high word) */
        IndexProbe<TAGS> pr[PROBE_U];
        bool live[PROBE_U];
#pragma unroll
        for (int u = 0; u < PROBE_U; u++) {
            const uint64_t i = i0 + (uint64_t)u * blockDim.x;
            live[u] = i < n;
            key[u] = live[u] ? ntl_stream_load(&mx[i].hash) : 0; /* streamed once: keep L2 for the tags */
            pm[u] = live[u] ? ntl_stream_load((const uint64_t *)&mx[i].pos) : 0;
        }
#pragma unroll
        for (int u = 0; u < PROBE_U; u++)
            if (live[u]) pr[u].start(key[u], slots, tags, bits); /* the first (random) loads of all four are in flight together */
#pragma unroll
        for (int u = 0; u < PROBE_U; u++) {
            if (!live[u]) continue;
            const Cand c = pr[u].finish(key[u], slots, tags, special, mask);
            /* position in the read beside the candidate, the read strand in its bit 31: all the map kernels read (EmitArgs::rpos) */
            const uint32_t meta = c.meta | ((uint32_t)(pm[u] >> 32) << 31);
            ntl_stream_store((uint64_t *)&cand[i0 + (uint64_t)u * blockDim.x], (uint64_t)c.cpos | ((uint64_t)meta << 32));
            rpos[i0 + (uint64_t)u * blockDim.x] = (uint32_t)pm[u];
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
/* Hits / runs per read staged in LDS are template parameters of map_kernel; reads are sorted into three size classes by
   their number of minimizers (map_class_of) and each class runs with the staging that fits it: 256/64, 512/128,
   1024/128.  Per hit 16 bytes of LDS (three 32-bit words and two 16-bit indices), per run 26: 5.6 KB, 11.3 KB and 19.3 KB per
   wavefront -- 28, 14 and 8 wavefronts per CU.  (Round 2: six + ten 32-bit arrays, 17 KB for 512 hits, 9 wavefronts per CU,
   and every read above 512 hits -- a fifth of 20-kb HiFi reads -- on the global scratch arrays.)  Reads beyond the largest
   class use global scratch (map_overflow_kernel), where every array is 32 bits wide. */
#define MAP_NHA 5    /* per-hit u32 arrays of the global scratch form */
#define MAP_NRA 10   /* per-run u32 arrays of the global scratch form */
#define MAP_NCLASS 3
#define MAP_GROUP 8  /* consecutive reads a wavefront looks at per step (small: a batch of 50 k reads must still fill the device) */

struct MapArgs {
    const uint32_t *rpos;   /* [minimizers] position in the read; the strand is bit 31 of the candidate's meta (EmitArgs::rpos) */
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
    uint32_t *nmx_out;                            /* [1] the batch's number of read minimizers, for the host's hit fraction */
    /* The sketch may still be in flight when these kernels are queued (nothing waits on the host for its size): when its
       minimizer total turns out larger than the arrays that were sized from the expected density, the records are incomplete
       and every kernel here leaves the batch alone; the host makes the sketch again and queues the map a second time. */
    const uint32_t *mx_total;                     /* NULL: the count was known when the call was made */
    uint32_t mx_cap;
};

__device__ __forceinline__ bool map_sketch_overflowed(const MapArgs &A) { return A.mx_total && *A.mx_total > A.mx_cap; }

/* Per-hit state.  cf = contig << 3 | HF_KEEP | read strand << 1 | contig strand (contig ids stay below 2^29, ntl_index_build);
   IT = uint16_t in LDS (indices below the class's capacity), uint32_t on the global scratch. */
template <typename IT> struct HitArr { uint32_t *cf, *cpos, *rpos; IT *run, *ord; };
template <typename IT> struct RunArr { uint32_t *ctg, *mn, *mx; IT *start, *leader, *flag, *cnt, *mni, *mxi, *last; };

#define HF_CS 1u   /* contig strand */
#define HF_RS 2u   /* read strand */
#define HF_KEEP 4u
#define HF_CTG_SHIFT 3
#define RF_NOISY 1u
#define RF_SUB 2u
#define AX_DUP 1u
#define AX_FILT 2u
#define AX_BRK 4u

/* flags of the PAF stage, set by several lanes on neighbouring entries: a 32-bit atomic on the word that holds the entry */
__device__ __forceinline__ void map_ax_or(uint32_t *aux, uint32_t i, uint32_t bits) { atomicOr(&aux[i], bits); }
__device__ __forceinline__ void map_ax_or(uint16_t *aux, uint32_t i, uint32_t bits)
{
    atomicOr(reinterpret_cast<uint32_t *>(aux) + (i >> 1), bits << (16u * (i & 1u))); /* the arrays start on a word boundary */
}

/* stable in-place compaction of the hits whose HF_KEEP bit is set; returns the new count */
template <typename IT>
__device__ __forceinline__ uint32_t map_compact(HitArr<IT> H, uint32_t n)
{
    const uint32_t lane = threadIdx.x;
    uint32_t m = 0;
    for (uint32_t c = 0; c < n; c += MAP_NT) {
        const uint32_t i = c + lane;
        uint32_t a = 0, b = 0, d = 0;
        if (i < n) { a = H.cf[i]; b = H.cpos[i]; d = H.rpos[i]; }
        const bool keep = i < n && (a & HF_KEEP);
        const unsigned long long bal = __ballot(keep);
        __syncthreads(); /* every lane holds its element before anything is overwritten */
        if (keep) {
            const uint32_t o = m + ntl_mbcnt(bal);
            H.cf[o] = a & ~HF_KEEP; H.cpos[o] = b; H.rpos[o] = d;
        }
        m += (uint32_t)__popcll(bal);
    }
    __syncthreads();
    return m;
}

/* run id of every hit (consecutive hits on one contig, itertools.groupby); returns #runs */
template <typename IT>
__device__ __forceinline__ uint32_t map_number_runs(HitArr<IT> H, uint32_t n)
{
    const uint32_t lane = threadIdx.x;
    uint32_t R = 0;
    for (uint32_t c = 0; c < n; c += MAP_NT) {
        const uint32_t i = c + lane;
        const bool isb = i < n && (i == 0 || (H.cf[i] >> HF_CTG_SHIFT) != (H.cf[i - 1] >> HF_CTG_SHIFT));
        const unsigned long long bal = __ballot(isb);
        if (i < n) H.run[i] = (IT)(R + ntl_mbcnt(bal) + (isb ? 1u : 0u) - 1u);
        R += (uint32_t)__popcll(bal);
    }
    __syncthreads();
    return R;
}

template <typename IT>
__device__ __forceinline__ void map_fill_runs(HitArr<IT> H, uint32_t n, RunArr<IT> RU, uint32_t R)
{
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < n; i += MAP_NT) {
        const uint32_t cg = H.cf[i] >> HF_CTG_SHIFT;
        if (i == 0 || cg != (H.cf[i - 1] >> HF_CTG_SHIFT)) { RU.start[H.run[i]] = (IT)i; RU.ctg[H.run[i]] = cg; }
    }
    __syncthreads();
    /* leader = first run on the same contig */
    for (uint32_t r = lane; r < R; r += MAP_NT) {
        uint32_t ld = r;
        const uint32_t c = RU.ctg[r];
        for (uint32_t j = 0; j < r; j++)
            if (RU.ctg[j] == c) { ld = j; break; }
        RU.leader[r] = (IT)ld;
        RU.flag[r] = 0;
    }
    __syncthreads();
}

/* (value, index) of the first minimum / first maximum over the lanes of the wavefront: butterfly with ties to the smaller
   index; lanes without a candidate pass idx = 0xFFFFFFFF */
__device__ __forceinline__ void map_wave_argmin(uint32_t &v, uint32_t &idx)
{
    const int lane = (int)threadIdx.x;
    for (int d = 1; d < MAP_NT; d <<= 1) {
        const uint32_t ov = __shfl(v, lane ^ d), oi = __shfl(idx, lane ^ d);
        if (oi != NTL_NONE && (idx == NTL_NONE || ov < v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
}
__device__ __forceinline__ void map_wave_argmax(uint32_t &v, uint32_t &idx)
{
    const int lane = (int)threadIdx.x;
    for (int d = 1; d < MAP_NT; d <<= 1) {
        const uint32_t ov = __shfl(v, lane ^ d), oi = __shfl(idx, lane ^ d);
        if (oi != NTL_NONE && (idx == NTL_NONE || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
}

/* One read on one wavefront.  GLOBAL = false: hits and runs live in the workgroup's LDS arrays -- the pointers are set from
 * them unconditionally, so every access is a DS instruction; a read that does not fit (more than MAP_CAPH hits or MAP_CAPR
 * runs) is put on the overflow list and left alone.  GLOBAL = true (map_overflow_kernel): the same code on the read's region
 * of the global scratch arrays. */
template <int MAP_CAPH, int MAP_CAPR, bool GLOBAL, typename IT>
__device__ __forceinline__ void map_read(const MapArgs &A, const uint32_t r, HitArr<IT> HL, RunArr<IT> RL)
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
    HitArr<IT> H = HL;
    RunArr<IT> RU = RL;
    if (!GLOBAL) {
        if (nc > MAP_CAPH) { /* wave-uniform */
            if (lane == 0) A.over_list[atomicAdd(A.over_count, 1u)] = r;
            return;
        }
    } else {
        uint32_t *g = A.scr + m0;
        H.cf = g; H.cpos = g + A.scr_stride; H.rpos = g + 2 * A.scr_stride;
        H.run = (IT *)(g + 3 * A.scr_stride); H.ord = (IT *)(g + 4 * A.scr_stride);
        uint32_t *q = A.scr + MAP_NHA * A.scr_stride + m0;
        RU.ctg = q; RU.mn = q + A.scr_stride; RU.mx = q + 2 * A.scr_stride;
        RU.start = (IT *)(q + 3 * A.scr_stride); RU.leader = (IT *)(q + 4 * A.scr_stride); RU.flag = (IT *)(q + 5 * A.scr_stride);
        RU.cnt = (IT *)(q + 6 * A.scr_stride); RU.mni = (IT *)(q + 7 * A.scr_stride); RU.mxi = (IT *)(q + 8 * A.scr_stride);
        RU.last = (IT *)(q + 9 * A.scr_stride);
    }

    if (nc == 0) goto done;

    /* bin/ntlink_pair.py:364-367 hits in read order; bin/ntlink_utils.py:206 contig length >= z */
    for (uint32_t c = 0; c < nmx; c += MAP_NT) {
        const uint32_t i = c + lane;
        Cand cd;
        cd.cpos = 0; cd.meta = 0;
        uint32_t rp = 0, rs = 0; /* the minimizer's position in the read, its strand */
        if (i < nmx) {
            cd = A.cand[m0 + i];
            rp = A.rpos[m0 + i];
            rs = cd.meta >> 31;
            cd.meta &= 0x7FFFFFFFu;
        }
        bool v = (cd.meta & 1u) != 0;
        if (v && !P.repeat_filter) v = (int64_t)A.ctg_len[cd.meta >> 2] >= (int64_t)P.z;
        const unsigned long long bal = __ballot(v);
        if (v) {
            const uint32_t o = n + ntl_mbcnt(bal);
            H.cf[o] = ((cd.meta >> 2) << HF_CTG_SHIFT) | ((cd.meta >> 1) & 1u) | (rs << 1);
            H.cpos[o] = cd.cpos; H.rpos[o] = rp;
        }
        n += (uint32_t)__popcll(bal);
    }
    __syncthreads();

    if (P.repeat_filter) {
        /* :368-374 drop minimizers that occur more than once among the read's hits, then z */
        for (uint32_t i = lane; i < n; i += MAP_NT) {
            const uint32_t c = H.cf[i] >> HF_CTG_SHIFT, p = H.cpos[i];
            bool dup = false;
            for (uint32_t j = 0; j < n; j++)
                if (j != i && (H.cf[j] >> HF_CTG_SHIFT) == c && H.cpos[j] == p) { dup = true; break; }
            const bool keep = !dup && (int64_t)A.ctg_len[c] >= (int64_t)P.z;
            H.cf[i] = (H.cf[i] & ~HF_KEEP) | (keep ? HF_KEEP : 0u);
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
    }
    map_fill_runs(H, n, RU, R);

    /* bin/ntlink_utils.py:217-234 noisy contigs: span on the contig longer than the read allows */
    {
        if (R <= 8 && n >= 128) {
            /* few, long runs (HiFi: hundreds of hits on one contig): the lanes share each run, a butterfly over the wavefront
               merges their partial results (first argmin / first argmax: ties go to the smaller index) */
            for (uint32_t q = 0; q < R; q++) {
                const uint32_t s = RU.start[q], e = q + 1 < R ? (uint32_t)RU.start[q + 1] : n;
                uint32_t mn = 0, mni = NTL_NONE, mx = 0, mxi = NTL_NONE;
                for (uint32_t i = s + lane; i < e; i += MAP_NT) { /* i ascends: strict compares keep the first */
                    const uint32_t p = H.cpos[i];
                    if (mni == NTL_NONE || p < mn) { mn = p; mni = i; }
                    if (mxi == NTL_NONE || p > mx) { mx = p; mxi = i; }
                }
                map_wave_argmin(mn, mni);
                map_wave_argmax(mx, mxi);
                if (lane == 0) { RU.cnt[q] = (IT)(e - s); RU.mn[q] = mn; RU.mni[q] = (IT)mni; RU.mx[q] = mx; RU.mxi[q] = (IT)mxi; }
            }
        } else
        for (uint32_t q = lane; q < R; q += MAP_NT) {
            const uint32_t s = RU.start[q], e = q + 1 < R ? (uint32_t)RU.start[q + 1] : n;
            uint32_t mn = H.cpos[s], mni = s, mx = mn, mxi = s;
            for (uint32_t i = s + 1; i < e; i++) {
                const uint32_t p = H.cpos[i];
                if (p < mn) { mn = p; mni = i; }  /* first argmin */
                if (p > mx) { mx = p; mxi = i; }  /* first argmax */
            }
            RU.cnt[q] = (IT)(e - s); RU.mn[q] = mn; RU.mni[q] = (IT)mni; RU.mx[q] = mx; RU.mxi[q] = (IT)mxi;
        }
        __syncthreads();
        if (lane == 0) {
            for (uint32_t q = 0; q < R; q++) {
                const uint32_t ld = RU.leader[q];
                if (ld == q) continue;
                RU.cnt[ld] = (IT)(RU.cnt[ld] + RU.cnt[q]);
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
            if (noisy) { RU.flag[q] = (IT)(RU.flag[q] | RF_NOISY); my_noisy = true; }
        }
        const bool any_noisy = __ballot(my_noisy) != 0ull;
        __syncthreads();
        if (any_noisy) {
            for (uint32_t i = lane; i < n; i += MAP_NT) {
                const bool keep = !(RU.flag[RU.leader[H.run[i]]] & RF_NOISY);
                H.cf[i] = (H.cf[i] & ~HF_KEEP) | (keep ? HF_KEEP : 0u);
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
            for (uint32_t q = 0; q < R; q++) RU.last[RU.leader[q]] = (IT)q;
            bool any = false;
            if (!P.sensitive) {
                /* mark_subsumed_specific :280-294: a run strictly inside (first, last) of any contig
                   marks its whole contig */
                uint32_t pm = 0;
                for (uint32_t q = 0; q < R; q++) {
                    if (pm > q) { RU.flag[RU.leader[q]] = (IT)(RU.flag[RU.leader[q]] | RF_SUB); any = true; }
                    if (RU.leader[q] == q && RU.last[q] > pm) pm = RU.last[q];
                }
                if (any)
                    for (uint32_t q = 0; q < R; q++)
                        if (RU.flag[RU.leader[q]] & RF_SUB) RU.flag[q] = (IT)(RU.flag[q] | RF_SUB);
            } else {
                /* mark_subsumed_sensitive :271-278: runs strictly between two occurrences of
                   ANOTHER contig */
                uint32_t m1 = 0, l1 = NTL_NONE, m2 = 0;
                for (uint32_t q = 0; q < R; q++) {
                    const uint32_t ld = RU.leader[q];
                    const uint32_t cover = l1 != ld ? m1 : m2;
                    if (cover > q) { RU.flag[q] = (IT)(RU.flag[q] | RF_SUB); any = true; }
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
                H.cf[i] = (H.cf[i] & ~HF_KEEP) | (keep ? HF_KEEP : 0u);
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
        const uint32_t s = RU.start[q], e = q + 1 < R ? (uint32_t)RU.start[q + 1] : n;
        MapRec M;
        M.read = r; M.ctg = RU.ctg[q]; M.n_hits = e - s; M.pad = 0; M.hit_off = s;
        A.maps[m0 + q] = M;
    }
    for (uint32_t i = lane; i < n; i += MAP_NT) {
        HitRec h;
        const uint32_t f = H.cf[i];
        h.ctg_pos = H.cpos[i]; h.read_pos = H.rpos[i];
        h.ctg_strand = (uint8_t)(f & 1u); h.read_strand = (uint8_t)((f >> 1) & 1u);
        h.pad[0] = h.pad[1] = 0;
        A.hits[m0 + i] = h;
    }

    /* bin/ntlink_paf_output.py:103-135 */
    for (uint32_t q = 0; q < R; q++) {
        const uint32_t s = RU.start[q], e = q + 1 < R ? (uint32_t)RU.start[q + 1] : n;
        const uint32_t m = e - s;
        const uint32_t ctg = RU.ctg[q];
        /* :95-101 read order == (ctg_pos, read_pos) order, or its exact reverse */
        bool nd = true, sd = true;
        uint32_t same = 0;
        for (uint32_t c = s; c < e; c += MAP_NT) {
            const uint32_t i = c + lane;
            bool x1 = true, x2 = true, sm = false;
            if (i < e) {
                const uint32_t f = H.cf[i];
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
        IT *aux = H.run; /* run ids are no longer needed: reuse as per-sorted-index flags */
        __syncthreads();
        for (uint32_t i = s + lane; i < e; i += MAP_NT) {
            const uint32_t cp = H.cpos[i], rp = H.rpos[i];
            uint32_t rk = 0;
            for (uint32_t j = s; j < e; j++) {
                const uint32_t cj = H.cpos[j];
                rk += (cj < cp || (cj == cp && H.rpos[j] < rp)) ? 1u : 0u;
            }
            H.ord[s + rk] = (IT)i;
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
                if (t + 2 >= nt) { map_ax_or(aux, s + t + 1, AX_BRK); continue; }
                const uint32_t c2 = H.rpos[H.ord[s + t + 2]];
                const bool d2 = (aux[s + t + 2] & AX_DUP) != 0;
                if (d2 || (incr ? a <= c2 : a >= c2)) { map_ax_or(aux, s + t + 1, AX_FILT); continue; }
                if (t > 0) {
                    const uint32_t cm = H.rpos[H.ord[s + t - 1]];
                    const bool dm = (aux[s + t - 1] & AX_DUP) != 0;
                    if (dm || (incr ? cm <= b : cm >= b)) { map_ax_or(aux, s + t, AX_FILT); continue; }
                }
                map_ax_or(aux, s + t + 1, AX_BRK);
            }
            __syncthreads();
        }
        /* :18-32 blocks, then :114-135 one line per block */
        if (lane == 0) {
            uint32_t first = 0, last = 0, cnt = 0, sm = 0, k_np = np;
            for (uint32_t t = 0; t <= m; t++) {
                const uint32_t ax = t < m ? (uint32_t)aux[s + t] : AX_BRK;
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
                const uint32_t f = H.cf[H.ord[s + t]];
                sm += (f & 1u) == ((f >> 1) & 1u) ? 1u : 0u;
            }
            aux[s] = (IT)(k_np - np); /* publish the number of blocks */
        }
        __syncthreads();
        np += aux[s];
        __syncthreads();
    }

done:
    if (lane == 0) { A.n_maps[r] = R; A.n_hits[r] = n; A.n_pafs[r] = np; }
}

/* size class of a read by its number of minimizers (an upper bound of its hits) */
__device__ __forceinline__ int map_class_of(uint32_t nmx) { return nmx <= 256u ? 0 : (nmx <= 512u ? 1 : 2); }

/* One launch per size class, each over ALL reads with a resident-size grid: a wavefront looks at MAP_GROUP consecutive reads
   at a time, votes which of them belong to its class and maps those one after the other.
   No list, no atomic; a class without reads costs one pass over the offsets. */
template <int MAP_CAPH, int MAP_CAPR, int CLASS>
__global__ __launch_bounds__(MAP_NT) NTL_MAIN_STREAM_SGPRS void map_kernel(MapArgs A)
{
    NTL_PRIO_LATENCY_BOUND();
    __shared__ uint32_t s_h32[3][MAP_CAPH];
    __shared__ uint32_t s_h16[MAP_CAPH];      /* two uint16 arrays (run, ord), on a word boundary for the flag atomics */
    __shared__ uint32_t s_r32[3][MAP_CAPR];
    __shared__ uint32_t s_r16[(7 * MAP_CAPR + 1) / 2];
    if (map_sketch_overflowed(A)) return;
    HitArr<uint16_t> H;
    H.cf = s_h32[0]; H.cpos = s_h32[1]; H.rpos = s_h32[2];
    H.run = reinterpret_cast<uint16_t *>(s_h16); H.ord = H.run + MAP_CAPH;
    RunArr<uint16_t> RU;
    RU.ctg = s_r32[0]; RU.mn = s_r32[1]; RU.mx = s_r32[2];
    uint16_t *q = reinterpret_cast<uint16_t *>(s_r16);
    RU.start = q; RU.leader = q + MAP_CAPR; RU.flag = q + 2 * MAP_CAPR; RU.cnt = q + 3 * MAP_CAPR; RU.mni = q + 4 * MAP_CAPR;
    RU.mxi = q + 5 * MAP_CAPR; RU.last = q + 6 * MAP_CAPR;
    const uint32_t lane = threadIdx.x;
    for (uint64_t base = (uint64_t)blockIdx.x * MAP_GROUP; base < A.nreads; base += (uint64_t)gridDim.x * MAP_GROUP) {
        const uint64_t r = base + lane;
        bool mine = false;
        if (lane < MAP_GROUP && r < A.nreads) mine = map_class_of(A.mx_off[r + 1] - A.mx_off[r]) == CLASS;
        unsigned long long bal = __ballot(mine);
        while (bal) {
            const uint32_t j = (uint32_t)__ffsll((long long)bal) - 1u;
            bal &= bal - 1ull;
            map_read<MAP_CAPH, MAP_CAPR, false, uint16_t>(A, (uint32_t)(base + j), H, RU);
            __syncthreads(); /* the staging arrays are reused by the next read */
        }
    }
}

/* the reads map_kernel could not stage in LDS, on their regions of the global scratch arrays */
__global__ __launch_bounds__(MAP_NT) NTL_MAIN_STREAM_SGPRS void map_overflow_kernel(MapArgs A)
{
    NTL_PRIO_LATENCY_BOUND();
    if (map_sketch_overflowed(A)) return;
    const uint32_t n = *A.over_count;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        map_read<1, 1, true, uint32_t>(A, A.over_list[i], HitArr<uint32_t>(), RunArr<uint32_t>());
        __syncthreads();
    }
}

/* ---------------------------------------------------------------------------- gather ------ */

/* Dense, read-ordered mappings and PAF records from the per-read regions.  The HITS stay where map_read wrote them -- read r's at
   [mx_off[r], mx_off[r] + n_hits[r]) of the region array -- and a dense mapping's hit_off points there: every consumer on the
   device reaches the hits through hit_off (the text kernels, the tally's first / last hits), and copying 12 bytes per hit a
   second time was most of this kernel (C5: 0.85 GB read + 0.85 GB written per sub-batch, 0.33 ms of it).  ntl_mapres_download
   makes the dense copy it hands to the host when it is asked (map_densify_kernel). */
__global__ void map_gather_kernel(MapArgs A, const uint32_t *off_maps, const uint32_t *off_pafs, MapRec *d_maps, PafRec *d_pafs)
{
    if (map_sketch_overflowed(A)) return;
    const uint32_t r = blockIdx.x;
    const uint32_t m0 = A.mx_off[r];
    if (r == 0 && threadIdx.x == 0) *A.nmx_out = A.mx_off[A.nreads];
    const uint32_t nm = A.n_maps[r], npf = A.n_pafs[r];
    const uint32_t om = off_maps[r], op = off_pafs[r];
    for (uint32_t i = threadIdx.x; i < nm; i += blockDim.x) {
        MapRec M = A.maps[m0 + i];
        M.hit_off += m0;
        d_maps[om + i] = M;
    }
    for (uint32_t i = threadIdx.x; i < npf; i += blockDim.x) d_pafs[op + i] = A.pafs[m0 + i];
}

/* n_hits of every mapping as a u32 array (the input of the scan that numbers the hits densely) */
__global__ void map_nhits_kernel(const MapRec *maps, uint32_t n_maps, uint32_t *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_maps) out[i] = maps[i].n_hits;
}

/* the dense copy for the host: mapping m's hits to [doff[m], doff[m] + n_hits), its hit_off rewritten to match */
__global__ void map_densify_kernel(const MapRec *maps, uint32_t n_maps, const HitRec *hits, const uint32_t *doff, MapRec *d_maps, HitRec *d_hits)
{
    const uint32_t m = blockIdx.x;
    if (m >= n_maps) return;
    MapRec M = maps[m];
    const uint32_t o = doff[m];
    for (uint32_t i = threadIdx.x; i < M.n_hits; i += blockDim.x) d_hits[o + i] = hits[M.hit_off + i];
    if (threadIdx.x == 0) { M.hit_off = o; d_maps[m] = M; }
}
