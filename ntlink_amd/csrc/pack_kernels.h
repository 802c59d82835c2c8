/*
 * Device-side ingest: ASCII bases -> the batch layout the sketch kernels read (SURVEY row f1;
 * replaces what btllib's SeqReader + NtHash do per base inside `indexlr`, ntLink:199,223).
 *
 *   packed        2 bits per base, 16 per u32, NTL_LEAD_PAD bases of zero padding in front; a
 *                 non-ACGT byte packs as 0 and is kept out of every k-mer by the run table
 *   run table     maximal runs of ACGT/acgt inside each sequence: run_start (offset in the
 *                 sequence), run_len, and seq_run_first[s] = runs before sequence s
 *
 * Global positions ("gpos") count bases of the whole batch from the start of the lead pad, so
 * word w of `packed` holds gpos 16w..16w+15 and one thread of the kernels below owns the 32
 * positions 32t..32t+31 (two packed words, one word of the bit arrays).
 *
 *   pack_kernel        raw bytes -> packed words + valid32 (bit j: gpos 32t+j is ACGT)
 *   seq_mark_kernel    ss32: bit set at the first gpos of every sequence (runs never cross it)
 *   run_count_kernel   starts32 (bit: an ACGT run begins here) and their number per thread
 *   run_fill_kernel    rank of each start / end bit (exclusive scan of the counts) -> gpos of the
 *                      first and last base of run r; starts and ends alternate, so the i-th start
 *                      and the i-th end belong to the same run
 *   run_finish_kernel  per run: owning sequence (binary search of seq_base), offset in it, length
 *   seq_runs_kernel    per sequence: seq_run_first = rank of its first gpos; flags a sequence
 *                      with more than one run
 */
#pragma once
#include "dev_common.h"

#define PACK_NT 256

/* A/a -> 0, C/c -> 1, G/g -> 2, T/t -> 3 from bits 1..2 of the ASCII code; ok = one of those eight */
__device__ __forceinline__ uint32_t pack_code(uint32_t c, uint32_t &ok)
{
    const uint32_t u = c & 0xDFu;
    ok = (u == 0x41u) | (u == 0x43u) | (u == 0x47u) | (u == 0x54u);
    const uint32_t x = (c >> 1) & 3u;
    return (x ^ (x >> 1)) & (0u - ok);
}

/* sixteen bases starting at base index i0 (may be negative inside the lead pad, or run past the end) */
__device__ __forceinline__ void pack16(const uint8_t *raw, int64_t i0, uint64_t total, uint32_t &word, uint32_t &valid)
{
    word = 0; valid = 0;
    if (i0 >= 0 && (uint64_t)i0 + 16 <= total) {
        const uint4 v = *reinterpret_cast<const uint4 *>(raw + i0); /* raw is 16-byte aligned, i0 a multiple of 16 */
        const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            uint32_t ok;
            const uint32_t cde = pack_code((q[j >> 2] >> (8 * (j & 3))) & 0xFFu, ok);
            word |= cde << (2 * j);
            valid |= ok << j;
        }
        return;
    }
    for (int j = 0; j < 16; j++) {
        const int64_t i = i0 + j;
        if (i < 0 || (uint64_t)i >= total) continue;
        uint32_t ok;
        const uint32_t cde = pack_code(raw[i], ok);
        word |= cde << (2 * j);
        valid |= ok << j;
    }
}

__global__ __launch_bounds__(PACK_NT) void pack_kernel(const uint8_t *raw, uint64_t total, uint32_t *packed, uint32_t *valid32, uint64_t n32)
{
    const uint64_t t = (uint64_t)blockIdx.x * PACK_NT + threadIdx.x;
    if (t >= n32) return;
    uint32_t w0, v0, w1, v1;
    pack16(raw, (int64_t)(32 * t) - NTL_LEAD_PAD, total, w0, v0);
    pack16(raw, (int64_t)(32 * t) + 16 - NTL_LEAD_PAD, total, w1, v1);
    *reinterpret_cast<uint2 *>(packed + 2 * t) = make_uint2(w0, w1);
    valid32[t] = v0 | (v1 << 16);
}

__global__ __launch_bounds__(PACK_NT) void seq_mark_kernel(const uint64_t *seq_base, uint32_t nseq, uint32_t *ss32)
{
    const uint32_t s = blockIdx.x * PACK_NT + threadIdx.x;
    if (s >= nseq) return;
    const uint64_t g = seq_base[s];
    atomicOr(&ss32[g >> 5], 1u << (g & 31u));
}

/* valid32 / ss32 carry one zero word of padding after their n32 entries */
__global__ __launch_bounds__(PACK_NT) void run_count_kernel(const uint32_t *valid32, const uint32_t *ss32, uint64_t n32,
                                                            uint32_t *starts32, uint32_t *cnt)
{
    const uint64_t t = (uint64_t)blockIdx.x * PACK_NT + threadIdx.x;
    if (t >= n32) return;
    const uint32_t V = valid32[t];
    const uint32_t Vp = t ? valid32[t - 1] >> 31 : 0u;
    const uint32_t st = V & (~((V << 1) | Vp) | ss32[t]);
    starts32[t] = st;
    cnt[t] = (uint32_t)__popc(st);
}

__global__ __launch_bounds__(PACK_NT) void run_fill_kernel(const uint32_t *valid32, const uint32_t *ss32, const uint32_t *starts32,
                                                           const uint32_t *rank, uint64_t n32, uint64_t *start_g, uint64_t *end_g)
{
    const uint64_t t = (uint64_t)blockIdx.x * PACK_NT + threadIdx.x;
    if (t >= n32) return;
    const uint32_t V = valid32[t];
    if (V == 0) return;
    const uint32_t S = ss32[t];
    const uint32_t Vp = t ? valid32[t - 1] >> 31 : 0u;
    const uint32_t Vn = valid32[t + 1] & 1u, Sn = ss32[t + 1] & 1u;
    uint32_t st = starts32[t];
    uint32_t en = V & (~((V >> 1) | (Vn << 31)) | ((S >> 1) | (Sn << 31)));
    uint32_t rs = rank[t];
    uint32_t re = rs - (Vp & V & ~S & 1u); /* a run that is open on entry has its start counted before t */
    while (st) {
        const int j = __ffs((int)st) - 1;
        st &= st - 1;
        start_g[rs++] = 32 * t + (uint32_t)j;
    }
    while (en) {
        const int j = __ffs((int)en) - 1;
        en &= en - 1;
        end_g[re++] = 32 * t + (uint32_t)j;
    }
}

__global__ __launch_bounds__(PACK_NT) void run_finish_kernel(const uint64_t *start_g, const uint64_t *end_g, uint32_t nruns,
                                                             const uint64_t *seq_base, uint32_t nseq, uint32_t *run_start, uint32_t *run_len)
{
    const uint32_t r = blockIdx.x * PACK_NT + threadIdx.x;
    if (r >= nruns) return;
    const uint64_t g = start_g[r];
    uint32_t lo = 0, hi = nseq; /* largest s < nseq with seq_base[s] <= g: the non-empty sequence that holds g */
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seq_base[mid] <= g) lo = mid; else hi = mid;
    }
    run_start[r] = (uint32_t)(g - seq_base[lo]);
    run_len[r] = (uint32_t)(end_g[r] - g + 1);
}

__device__ __forceinline__ uint32_t run_rank_at(const uint32_t *starts32, const uint32_t *rank, uint64_t g)
{
    const uint64_t t = g >> 5;
    const uint32_t j = (uint32_t)(g & 31u);
    return rank[t] + (uint32_t)__popc(starts32[t] & ((1u << j) - 1u));
}

/* s in [0, nseq]; starts32 / rank are defined (zero count) for the padding word after the data */
__global__ __launch_bounds__(PACK_NT) void seq_runs_kernel(const uint32_t *starts32, const uint32_t *rank, const uint64_t *seq_base,
                                                           uint32_t nseq, uint32_t *seq_run_first, uint32_t *any_multi)
{
    const uint32_t s = blockIdx.x * PACK_NT + threadIdx.x;
    if (s > nseq) return;
    const uint32_t a = run_rank_at(starts32, rank, seq_base[s]);
    seq_run_first[s] = a;
    if (s < nseq && run_rank_at(starts32, rank, seq_base[s + 1]) - a > 1u) atomicOr(any_multi, 1u);
}
