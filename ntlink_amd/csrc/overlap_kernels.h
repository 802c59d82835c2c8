/*
 * Consumer of the overlap-stage sketch (`indexlr --long --pos -k 15 -w 5`, ntLink:243-251):
 * read_minimizer_line of bin/ntlink_overlap_sequences.py:170-190 on device-resident minimizer records.
 *
 *   per sequence (= one line of the TSV):  keep the minimizers whose position lies in one of the sequence's valid regions
 *   [start, end] (is_in_valid_region :138-143) and whose hash occurs exactly ONCE among those -- the reference inserts the first
 *   occurrence into a dict, collects every later one in dup_mxs and deletes the collected keys after the line.
 *
 * The duplicate test is exact and needs no sort: every sequence owns a private open-addressing sub-table of
 * 2 x (its minimizer count) slots inside one array -- records of a sequence are contiguous, so the sub-table of sequence s
 * is slots [2*mx_off[s], 2*mx_off[s+1]) -- keyed by the 64-bit hash alone.  Insert = one atomicCAS claiming a slot; a later
 * arrival of the same key sets the slot's bit in a duplicate bitmap (the scheme of the contig index, map_kernels.h).  A second
 * pass looks every valid record up again and keeps it when its slot's bit is clear; an offset scan and a gather leave the
 * kept records dense, in input order.  The key 2^64-1 (the empty marker) is counted per sequence on the side.
 * HBM-bound: 16 B/record read twice + 16 B/record of table written and read + 4 B flag + 16 B/kept record out.
 */
#pragma once
#include "sketch_kernels.h"

struct OvlArgs {
    const MxRecord *rec;
    uint64_t n;
    const uint32_t *mx_off;    /* [nseq+1] */
    const uint32_t *reg_off;   /* [nseq+1] valid regions of sequence s: reg_off[s] .. reg_off[s+1] */
    const uint32_t *reg_start; /* inclusive */
    const uint32_t *reg_end;   /* inclusive */
    unsigned long long *slots; /* [2n] */
    uint32_t *dupbits;         /* [2n/32 + 1] */
    uint32_t *side;            /* [nseq] occurrences of the key 2^64-1 among the valid records */
    uint32_t *keep;            /* [n+1] 1 = valid region (after ovl_insert) / kept (after ovl_resolve) */
};

__device__ __forceinline__ bool ovl_valid(const OvlArgs &A, uint32_t seq, uint32_t pos)
{
    for (uint32_t r = A.reg_off[seq], e = A.reg_off[seq + 1]; r < e; r++)
        if (A.reg_start[r] <= pos && pos <= A.reg_end[r]) return true;
    return false;
}

__device__ __forceinline__ uint64_t ovl_home(uint64_t key, uint64_t size)
{
    return (uint64_t)(((key >> 32) * size) >> 32); /* size < 2^32: top hash bits spread over the sub-table */
}

__global__ __launch_bounds__(256) void ovl_insert_kernel(OvlArgs A)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const MxRecord R = A.rec[i];
    const uint32_t seq = R.meta >> 1;
    const bool valid = ovl_valid(A, seq, R.pos);
    A.keep[i] = valid ? 1u : 0u;
    if (!valid) return;
    if (R.hash == NTL_INF) { atomicAdd(&A.side[seq], 1u); return; }
    const uint64_t base = 2ull * A.mx_off[seq], size = 2ull * (A.mx_off[seq + 1] - A.mx_off[seq]);
    uint64_t j = ovl_home(R.hash, size);
    for (;;) {
        const uint64_t s = base + j;
        const unsigned long long old = atomicCAS(&A.slots[s], (unsigned long long)NTL_INF, (unsigned long long)R.hash);
        if (old == NTL_INF) return;
        if (old == R.hash) { atomicOr(&A.dupbits[s >> 5], 1u << (s & 31u)); return; }
        if (++j == size) j = 0;
    }
}

__global__ __launch_bounds__(256) void ovl_resolve_kernel(OvlArgs A)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n || !A.keep[i]) return;
    const MxRecord R = A.rec[i];
    const uint32_t seq = R.meta >> 1;
    if (R.hash == NTL_INF) { A.keep[i] = A.side[seq] == 1u ? 1u : 0u; return; }
    const uint64_t base = 2ull * A.mx_off[seq], size = 2ull * (A.mx_off[seq + 1] - A.mx_off[seq]);
    uint64_t j = ovl_home(R.hash, size);
    for (;;) {
        const uint64_t s = base + j;
        if (A.slots[s] == R.hash) {
            A.keep[i] = (A.dupbits[s >> 5] >> (s & 31u)) & 1u ? 0u : 1u;
            return;
        }
        if (++j == size) j = 0;
    }
}

/* kept records to their scanned places; offsets of the sequences in the dense array */
__global__ __launch_bounds__(256) void ovl_gather_kernel(const MxRecord *rec, uint64_t n, const uint32_t *keep, const uint32_t *dst,
                                                         MxRecord *out, const uint32_t *mx_off, uint32_t nseq, uint32_t *out_off)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && keep[i]) out[dst[i]] = rec[i];
    if (i <= nseq) out_off[i] = dst[mx_off[i]]; /* dst has n+1 entries: dst[n] = total */
}
