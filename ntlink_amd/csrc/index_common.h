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
    uint32_t meta; /* bit 0: found and unique, bit 1: contig strand, bits 2..30: contig; bit 31: the READ minimizer's strand, put there
                      by the kernel that made the lookup (emit_kernel, probe_kernel) for the map kernels; IndexProbe leaves it 0 */
};

/* One lookup, split so that a thread can start several before it finishes the first: start() issues the first random load
 * (TAGS: the tag byte of the home slot; else the home slot itself), finish() walks the probe sequence.
 * TAGS = true: most lookups end on the tag (few read minimizers are in the index: ONT reads).  TAGS = false: the slots are
 * read directly -- when most lookups hit (HiFi reads) the tag is one more random cache line per lookup for nothing. */
template <bool TAGS>
struct IndexProbe {
    uint64_t s;
    uint64_t t8; /* TAGS: the tags of slots s .. s + 7 (the array carries a copy of its first eight bytes behind its end) */
    IndexSlot e0;
    __device__ __forceinline__ void start(uint64_t key, const IndexSlot *slots, const uint8_t *tags, int bits)
    {
        s = index_home(key, bits);
        t8 = 0;
        if (key == NTL_INF) return;
        if (TAGS) t8 = ntl_load_u64_a1(tags + s);
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
            /* Eight tags in one load: the probe sequence of nearly every lookup (an empty slot ends it; the table is at most half
               full) lies inside them, so a lookup that ends on the tags costs ONE memory round trip, not one per probed slot.
               Zero bytes and bytes equal to the key's tag by the subtract-and-mask test; its lowest flagged byte is exact, higher
               ones may be false alarms (borrows), so only the lowest is ever trusted and a refuted match is blanked out. */
            const uint64_t ones = 0x0101010101010101ull, high = 0x8080808080808080ull;
            const uint8_t tg = index_tag(key);
            const uint64_t x = t8;
            const uint64_t z = (x - ones) & ~x & high;
            const uint32_t iz = z ? (uint32_t)(__ffsll((long long)z) - 1) >> 3 : 8u; /* first empty slot among the eight */
            uint64_t y = x ^ (ones * tg);
            bool done = false;
            for (;;) {
                const uint64_t m = (y - ones) & ~y & high;
                const uint32_t im = m ? (uint32_t)(__ffsll((long long)m) - 1) >> 3 : 8u;
                if (im >= iz) break; /* no (further) match in front of the empty slot */
                const IndexSlot e = slots[(s + im) & mask];
                if (e.key == key) {
                    if (!(e.meta & 1u)) { c.cpos = e.pos; c.meta = e.meta | 1u; }
                    done = true;
                    break;
                }
                y |= 0xFFull << (8u * im); /* another key with the same tag */
            }
            if (!done && iz == 8u) { /* eight occupied slots without the key: on, slot by slot */
                uint64_t q = (s + 8) & mask;
                uint8_t tq = tags[q];
                for (;;) {
                    if (tq == 0) break;
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
