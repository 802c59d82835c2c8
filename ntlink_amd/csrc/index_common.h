/*
 * The contig minimizer index as the lookups see it (bin/ntlink_pair.py:189-211, :364-367): slot layout, home slot, tags and the
 * probe sequence.  Shared by probe_kernel (map_kernels.h) and emit_kernel (sketch_kernels.h: a sketch that is made for one
 * index looks its minimizers up while it emits them).
 */
#pragma once
#include "dev_common.h"

struct IndexSlot {
    uint64_t key;  /* NTL_INF = empty */
    uint32_t pos;
    uint32_t meta; /* bit 0: duplicate, bit 1: strand, bits 2..31: contig */
};

struct IndexSpecial { /* the one key that equals the empty marker */
    uint32_t cnt, pos, meta, pad;
};

__device__ __forceinline__ uint64_t index_home(uint64_t key, int bits)
{
    return (key * 0x9E3779B97F4A7C15ull) >> (64 - bits);
}

/* One-byte tags in front of the 16-byte slots: 0 = empty slot, otherwise 7 bits of the key | 1.  The tag
 * array is 16x smaller than the table, so it stays in L2 / Infinity Cache while the table does not; ~85 %
 * of read minimizers are absent from the index and are rejected on tags alone. */
__device__ __forceinline__ uint8_t index_tag(uint64_t key) { return (uint8_t)(((key >> 20) & 0xFEu) | 1u); }

/* what a lookup leaves for the map kernel */
struct Cand {
    uint32_t cpos;
    uint32_t meta; /* bit 0: found and unique, bit 1: contig strand, bits 2..31: contig */
};

/* One lookup, split so that a thread can start several before it finishes the first: start() issues the first random load
 * (TAGS: the tag byte of the home slot; else the home slot itself), finish() walks the probe sequence.
 * TAGS = true: most lookups end on the tag (few read minimizers are in the index: ONT reads).  TAGS = false: the slots are
 * read directly -- when most lookups hit (HiFi reads) the tag is one more random cache line per lookup for nothing. */
template <bool TAGS>
struct IndexProbe {
    uint64_t s;
    uint8_t t;
    IndexSlot e0;
    __device__ __forceinline__ void start(uint64_t key, const IndexSlot *slots, const uint8_t *tags, int bits)
    {
        s = index_home(key, bits);
        t = 0;
        if (key == NTL_INF) return;
        if (TAGS) t = tags[s];
        else e0 = slots[s];
    }
    __device__ __forceinline__ Cand finish(uint64_t key, const IndexSlot *slots, const uint8_t *tags, const IndexSpecial *special,
                                           uint64_t mask) const
    {
        Cand c;
        c.cpos = 0; c.meta = 0;
        if (key == NTL_INF) {
            if (special->cnt == 1) { c.cpos = special->pos; c.meta = (special->meta & ~1u) | 1u; }
        } else if (TAGS) {
            const uint8_t tg = index_tag(key);
            uint64_t q = s;
            uint8_t tq = t;
            for (;;) {
                if (tq == 0) break; /* empty slot ends the probe sequence */
                if (tq == tg) {
                    const IndexSlot e = slots[q];
                    if (e.key == key) {
                        if (!(e.meta & 1u)) { c.cpos = e.pos; c.meta = e.meta | 1u; }
                        break;
                    }
                }
                q = (q + 1) & mask;
                tq = tags[q];
            }
        } else {
            uint64_t q = s;
            IndexSlot e = e0;
            for (;;) {
                if (e.key == NTL_INF) break; /* empty slot ends the probe sequence */
                if (e.key == key) {
                    if (!(e.meta & 1u)) { c.cpos = e.pos; c.meta = e.meta | 1u; }
                    break;
                }
                q = (q + 1) & mask;
                e = slots[q];
            }
        }
        return c;
    }
};
