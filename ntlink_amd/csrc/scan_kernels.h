/*
 * Exclusive prefix sums of u32 arrays (offsets of variable-length per-sequence / per-read
 * outputs, so that results land in input order without atomics -- the reference emits
 * everything in read order, ntLink:222, bin/ntlink_pair.py:352-408).
 *
 *   scan_reduce_kernel   per tile of SCAN_TILE items: sum -> tile_cnt
 *   scan_tiles_kernel    one workgroup: exclusive scan of tile_cnt in place, grand total
 *   scan_down_kernel     per tile: exclusive scan of the items + tile offset -> out
 */
#pragma once
#include "dev_common.h"

#define SCAN_NT 256
#define SCAN_IPT 8
#define SCAN_TILE (SCAN_NT * SCAN_IPT)

/* All three kernels take a batch index in blockIdx.y: array y lives at in + y*stride (tiles at
 * tile + y*tiles), so that several independent scans of equal length share one launch. */
__global__ __launch_bounds__(SCAN_NT) void scan_reduce_kernel(const uint32_t *in, uint64_t n, uint32_t *tile_cnt,
                                                              uint64_t stride, uint64_t tiles)
{
    __shared__ uint32_t s_tmp[SCAN_NT];
    in += (uint64_t)blockIdx.y * stride;
    tile_cnt += (uint64_t)blockIdx.y * tiles;
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_IPT;
    uint32_t c = 0;
    for (int i = 0; i < SCAN_IPT; i++)
        if (i0 + i < n) c += in[i0 + i];
    uint32_t total;
    block_excl_scan<SCAN_NT>(c, s_tmp, total);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = total;
}

/* single workgroup; data[0..n) becomes its exclusive scan, *total the sum */
__global__ __launch_bounds__(SCAN_NT) void scan_tiles_kernel(uint32_t *data, uint64_t n, uint32_t *total_out,
                                                             uint64_t total_stride, uint32_t *extra = nullptr)
{
    __shared__ uint32_t s_tmp[SCAN_NT];
    data += (uint64_t)blockIdx.y * n;
    total_out += (uint64_t)blockIdx.y * total_stride;
    uint32_t carry = 0;
    for (uint64_t base = 0; base < n; base += SCAN_TILE) {
        const uint64_t i0 = base + (uint64_t)threadIdx.x * SCAN_IPT;
        uint32_t v[SCAN_IPT];
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < SCAN_IPT; i++) {
            v[i] = i0 + i < n ? data[i0 + i] : 0u;
            c += v[i];
        }
        uint32_t total;
        uint32_t r = carry + block_excl_scan<SCAN_NT>(c, s_tmp, total);
#pragma unroll
        for (int i = 0; i < SCAN_IPT; i++) {
            if (i0 + i < n) data[i0 + i] = r;
            r += v[i];
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        *total_out = carry;
        if (extra) extra[blockIdx.y] = carry; /* the sums of a batch side by side, for one read-back */
    }
}

/* out may alias in; out[n] is not written */
__global__ __launch_bounds__(SCAN_NT) void scan_down_kernel(const uint32_t *in, uint32_t *out, uint64_t n,
                                                            const uint32_t *tile_off, uint64_t stride, uint64_t tiles)
{
    __shared__ uint32_t s_tmp[SCAN_NT];
    in += (uint64_t)blockIdx.y * stride;
    out += (uint64_t)blockIdx.y * stride;
    tile_off += (uint64_t)blockIdx.y * tiles;
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_IPT;
    uint32_t v[SCAN_IPT];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        v[i] = i0 + i < n ? in[i0 + i] : 0u;
        c += v[i];
    }
    uint32_t total;
    uint32_t r = tile_off[blockIdx.x] + block_excl_scan<SCAN_NT>(c, s_tmp, total);
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        if (i0 + i < n) out[i0 + i] = r;
        r += v[i];
    }
}

/* scan_tiles_kernel + scan_down_kernel in one launch (round 5: one dependent launch less on the streams' critical paths): every
 * workgroup sums the counts of the tiles in front of its own itself -- tiles^2 / 2 words in all, nothing for the few hundred tiles
 * of a strip or sequence table -- and the last one writes the total.  tile_cnt holds scan_reduce_kernel's sums, untouched. */
__global__ __launch_bounds__(SCAN_NT) void scan_down_self_kernel(const uint32_t *in, uint32_t *out, uint64_t n, const uint32_t *tile_cnt,
                                                                 uint64_t stride, uint64_t tiles, uint32_t *extra)
{
    __shared__ uint32_t s_tmp[SCAN_NT];
    in += (uint64_t)blockIdx.y * stride;
    out += (uint64_t)blockIdx.y * stride;
    tile_cnt += (uint64_t)blockIdx.y * tiles;
    uint32_t part = 0;
    for (uint64_t j = threadIdx.x; j < blockIdx.x; j += SCAN_NT) part += tile_cnt[j];
    uint32_t before;
    block_excl_scan<SCAN_NT>(part, s_tmp, before);
    const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_IPT;
    uint32_t v[SCAN_IPT];
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        v[i] = i0 + i < n ? in[i0 + i] : 0u;
        c += v[i];
    }
    uint32_t total;
    uint32_t r = before + block_excl_scan<SCAN_NT>(c, s_tmp, total);
#pragma unroll
    for (int i = 0; i < SCAN_IPT; i++) {
        if (i0 + i < n) out[i0 + i] = r;
        r += v[i];
    }
    if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {
        out[n] = before + total;
        if (extra) extra[blockIdx.y] = before + total;
    }
}
