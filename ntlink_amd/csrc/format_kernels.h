/*
 * Text of <prefix>.verbose_mapping.tsv (bin/ntlink_pair.py:308-313,382-388) and <prefix>.paf
 * (bin/ntlink_paf_output.py:131-135) made ON THE DEVICE from the dense records of a map result.
 *
 * A line of the verbose file is   read \t contig \t n_hits \t tok tok ... tok \n   with  tok = ctg_pos:S_read_pos:S  (S = + / -),
 * tokens separated by one space.  Lengths first, offsets by scans, bytes last:
 *
 *   fmt_len_kernel     one thread per hit: tok_len = digits(ctg_pos) + digits(read_pos) + 6 -- the token and the ONE byte behind it
 *                      (a space, or the line's newline behind the last token: which of the two is decided when the bytes are
 *                      written, so that a token's length does not depend on its place in the line);
 *                      one thread per mapping: hdr_len = |read| + |contig| + digits(n_hits) + 3;
 *                      one thread per PAF record: the length of its line.
 *   (scans)            tok_off = exclusive scan of tok_len, hdr_end = INCLUSIVE scan of hdr_len, paf_off = exclusive scan.
 *   fmt_fill_kernel    token h of mapping m starts at hdr_end[m] + tok_off[h]; the header of mapping m at
 *                      hdr_end[m] - hdr_len[m] + tok_off[hit_off[m]].  A hit finds its mapping by a binary search over the
 *                      mappings' hit offsets (they are dense and sorted).
 *   fmt_ends_kernel    first and last hit of every mapping: all the pair tally on the host needs of the hits
 *                      (bin/ntlink_pair.py:394-406), so that the hit records themselves never cross PCIe.
 *
 * Numbers are written back to front, one division by ten per digit (a multiplication by the reciprocal).
 */
#pragma once
#include "map_kernels.h"

struct FmtArgs {
    const MapRec *maps; const HitRec *hits; const PafRec *pafs;
    const uint32_t *hit_doff; /* [n_maps + 1]: hits of the mappings in front of each one = the DENSE number of its first hit; the hits
                                 themselves lie in per-read regions, mapping m's at hits[maps[m].hit_off ..] (map_gather_kernel) */
    uint32_t n_maps, n_hits, n_pafs; /* n_hits = 0 and do_verbose = 0: no verbose text (the ends are made either way) */
    int do_verbose;
    const uint64_t *read_name_off; const char *read_names; /* device copies of the batch's name table */
    const uint64_t *ctg_name_off; const char *ctg_names;
    const uint32_t *read_len, *ctg_len;
    uint32_t *tok_len, *hdr_len, *paf_len; /* n + 1 entries each: lengths, then (scanned in place) offsets with the total behind them */
    char *verbose, *paf;                   /* fill: the text */
    HitRec *ends;                          /* [2 n_maps] */
    MapRec *maps_out;                      /* [n_maps] the mappings with hit_off in the dense numbering, as ntl_mapres_download hands them out */
};

__device__ __forceinline__ uint32_t fmt_digits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}

/* writes v in decimal so that its last digit lands at end[-1]; returns nothing (the caller knows the length) */
__device__ __forceinline__ void fmt_put_u32(char *end, uint32_t v)
{
    do {
        const uint32_t q = v / 10u;
        *--end = (char)('0' + (v - q * 10u));
        v = q;
    } while (v);
}

__device__ __forceinline__ void fmt_put_u64(char *end, uint64_t v)
{
    do {
        const uint64_t q = v / 10u;
        *--end = (char)('0' + (uint32_t)(v - q * 10u));
        v = q;
    } while (v);
}

__device__ __forceinline__ uint32_t fmt_digits64(uint64_t v)
{
    uint32_t d = 1;
    while (v >= 10u) { v /= 10u; d++; }
    return d;
}

__global__ void fmt_len_kernel(FmtArgs A)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n_hits) { /* dense hit i: of the last mapping whose first dense number is <= i */
        uint32_t lo = 0, hi = A.n_maps;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (A.hit_doff[mid] <= i) lo = mid; else hi = mid;
        }
        const HitRec h = A.hits[A.maps[lo].hit_off + (i - A.hit_doff[lo])];
        A.tok_len[i] = fmt_digits(h.ctg_pos) + fmt_digits(h.read_pos) + 6u;
    }
    if (i < A.n_maps && A.do_verbose) {
        const MapRec m = A.maps[i];
        A.hdr_len[i] = (uint32_t)(A.read_name_off[m.read + 1] - A.read_name_off[m.read]) +
                       (uint32_t)(A.ctg_name_off[m.ctg + 1] - A.ctg_name_off[m.ctg]) + fmt_digits(m.n_hits) + 3u;
    }
    if (i < A.n_pafs) {
        const PafRec p = A.pafs[i];
        /* read \t rlen \t qs \t qe \t S \t contig \t clen \t ts \t te \t n \t (te - ts) \t 255 \n : 11 tabs, the strand, "255", the newline */
        A.paf_len[i] = (uint32_t)(A.read_name_off[p.read + 1] - A.read_name_off[p.read]) + fmt_digits(A.read_len[p.read]) + fmt_digits(p.q_start) +
                       fmt_digits(p.q_end) + (uint32_t)(A.ctg_name_off[p.ctg + 1] - A.ctg_name_off[p.ctg]) + fmt_digits(A.ctg_len[p.ctg]) +
                       fmt_digits(p.t_start) + fmt_digits(p.t_end) + fmt_digits(p.n_hits) + fmt_digits64((uint64_t)p.t_end - p.t_start) + 16u;
    }
}

/* hdr_len[0 .. n) exclusive-scanned in place with the total at [n]: hdr_end[m] = that of m + 1 */
__global__ void fmt_fill_kernel(FmtArgs A)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < A.n_hits) {
        /* the mapping of dense hit i */
        uint32_t lo = 0, hi = A.n_maps;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (A.hit_doff[mid] <= i) lo = mid; else hi = mid;
        }
        const MapRec m = A.maps[lo];
        const HitRec h = A.hits[m.hit_off + (i - A.hit_doff[lo])];
        char *p = A.verbose + (uint64_t)A.hdr_len[lo + 1] + A.tok_len[i];
        const uint32_t dc = fmt_digits(h.ctg_pos), dr = fmt_digits(h.read_pos);
        fmt_put_u32(p + dc, h.ctg_pos);
        p += dc;
        p[0] = ':'; p[1] = h.ctg_strand ? '+' : '-'; p[2] = '_';
        fmt_put_u32(p + 3 + dr, h.read_pos);
        p += 3 + dr;
        p[0] = ':'; p[1] = h.read_strand ? '+' : '-';
        p[2] = i + 1 == A.hit_doff[lo] + m.n_hits ? '\n' : ' ';
    }
    if (i < A.n_maps) {
        const MapRec m = A.maps[i];
        A.ends[2 * i] = A.hits[m.hit_off];
        A.ends[2 * i + 1] = A.hits[m.hit_off + m.n_hits - 1];
        MapRec d = m;
        d.hit_off = A.hit_doff[i];
        A.maps_out[i] = d;
    }
    if (i < A.n_maps && A.do_verbose) {
        const MapRec m = A.maps[i];
        char *p = A.verbose + (uint64_t)A.hdr_len[i] + A.tok_len[A.hit_doff[i]];
        const uint64_t r0 = A.read_name_off[m.read], r1 = A.read_name_off[m.read + 1];
        for (uint64_t j = r0; j < r1; j++) *p++ = A.read_names[j];
        *p++ = '\t';
        const uint64_t c0 = A.ctg_name_off[m.ctg], c1 = A.ctg_name_off[m.ctg + 1];
        for (uint64_t j = c0; j < c1; j++) *p++ = A.ctg_names[j];
        *p++ = '\t';
        const uint32_t d = fmt_digits(m.n_hits);
        fmt_put_u32(p + d, m.n_hits);
        p[d] = '\t';
    }
    if (i < A.n_pafs) {
        const PafRec q = A.pafs[i];
        char *p = A.paf + A.paf_len[i];
        const uint64_t r0 = A.read_name_off[q.read], r1 = A.read_name_off[q.read + 1];
        for (uint64_t j = r0; j < r1; j++) *p++ = A.read_names[j];
        auto num = [&](uint32_t v) { *p++ = '\t'; const uint32_t d = fmt_digits(v); fmt_put_u32(p + d, v); p += d; };
        num(A.read_len[q.read]); num(q.q_start); num(q.q_end);
        *p++ = '\t'; *p++ = q.strand ? '+' : '-'; *p++ = '\t';
        const uint64_t c0 = A.ctg_name_off[q.ctg], c1 = A.ctg_name_off[q.ctg + 1];
        for (uint64_t j = c0; j < c1; j++) *p++ = A.ctg_names[j];
        num(A.ctg_len[q.ctg]); num(q.t_start); num(q.t_end); num(q.n_hits);
        {
            const uint64_t span = (uint64_t)q.t_end - q.t_start;
            *p++ = '\t';
            const uint32_t d = fmt_digits64(span);
            fmt_put_u64(p + d, span);
            p += d;
        }
        p[0] = '\t'; p[1] = '2'; p[2] = '5'; p[3] = '5'; p[4] = '\n';
    }
}
