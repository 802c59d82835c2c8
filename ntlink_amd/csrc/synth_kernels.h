/*
 * Synthetic workloads built on the device (bench / parity-test support).
 *
 * Not part of the reference's interface: the reference has no generator.  BASELINE.json's configs are
 * synthetic (3 Gbp assembly + 90 Gbases of ONT-like reads); producing them on the host costs minutes of
 * numpy and tens of GB of PCIe traffic per run, so the generator writes the packed 2-bit batch layout of
 * ntl_batch_create directly in HBM:
 *   synth_genome_kernel   i.i.d. uniform ACGT, one thread per packed word (counter-based hash)
 *   synth_slices_kernel   one lane per output sequence: a slice of a source sequence, forward or
 *                         reverse-complemented, with per-base substitution / insertion / deletion
 *                         events from a per-sequence xorshift stream (contigs: no errors)
 *   unpack_kernel         packed -> ASCII (ntl_batch_download: what the oracle is fed in the tests)
 * Sequences made here contain only ACGT, so every sequence is one ACGT run.
 */
#pragma once
#include "dev_common.h"

__device__ __forceinline__ uint64_t synth_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_genome_kernel(uint32_t *packed, uint64_t nwords, uint64_t seed)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; /* two words per thread */
    if (2 * i >= nwords) return;
    const uint64_t r = synth_mix64(seed * 0xD1342543DE82EF95ull + i);
    packed[2 * i] = (uint32_t)r;
    if (2 * i + 1 < nwords) packed[2 * i + 1] = (uint32_t)(r >> 32);
}

struct SliceArgs {
    const uint32_t *src_packed;
    const uint64_t *src_seq_base; /* [n_src+1] */
    uint32_t n_src;
    uint32_t *dst_packed;         /* zero-filled */
    const uint64_t *dst_seq_base; /* [n+1] */
    const uint32_t *src_seq, *src_start; /* [n] */
    const uint8_t *reverse;       /* [n] or NULL */
    uint64_t n;
    uint64_t seed;
    uint32_t t_ins, t_del, t_sub; /* event thresholds on a 24-bit draw: ins < t_ins <= del < t_del; sub on a second draw */
};

/* bases of source consumed at most by an output of `len` bases (the host leaves this much room) */
__host__ __device__ inline uint64_t synth_span(uint32_t len) { return (uint64_t)len + len / 8u + 64u; }

__global__ void synth_slices_kernel(SliceArgs A)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const uint64_t d0 = A.dst_seq_base[i], d1 = A.dst_seq_base[i + 1];
    if (d1 <= d0) return;
    const uint32_t len = (uint32_t)(d1 - d0);
    const uint32_t sq = A.src_seq[i] < A.n_src ? A.src_seq[i] : 0u;
    const uint64_t s_lo = A.src_seq_base[sq], s_hi = A.src_seq_base[sq + 1]; /* [s_lo, s_hi) */
    const bool rev = A.reverse && A.reverse[i];
    const bool errors = (A.t_del | A.t_sub) != 0;
    const uint64_t span = errors ? synth_span(len) : (uint64_t)len;
    /* source cursor: forward from the slice start, or backward from its end (complemented) */
    int64_t sp = (int64_t)(s_lo + A.src_start[i]) + (rev ? (int64_t)span - 1 : 0);
    const int64_t step = rev ? -1 : 1;
    uint32_t x = (uint32_t)synth_mix64(A.seed ^ (i * 0x9E3779B97F4A7C15ull)) | 1u;
    uint64_t gb = d0;
    uint32_t acc = 0;
    bool first = true;
    for (uint32_t o = 0; o < len; o++) {
        uint32_t code;
        bool from_src = true;
        uint32_t r = 0;
        if (errors) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            r = x;
            const uint32_t ev = r >> 8;
            if (ev < A.t_ins) from_src = false;
            else if (ev < A.t_del) sp += step; /* deletion: one source base is skipped */
        }
        if (from_src) {
            int64_t q = sp;
            if (q < (int64_t)s_lo) q = (int64_t)s_lo;
            if (q >= (int64_t)s_hi) q = (int64_t)s_hi - 1;
            code = load_base(A.src_packed, (uint64_t)q);
            if (rev) code = 3u - code;
            sp += step;
            if (errors) {
                x ^= x << 13; x ^= x >> 17; x ^= x << 5;
                if ((x >> 8) < A.t_sub) code = (code + 1u + (x & 0xFFu) % 3u) & 3u;
            }
        } else {
            code = r & 3u;
        }
        acc |= code << (2u * ((uint32_t)gb & 15u));
        if (((uint32_t)gb & 15u) == 15u) {
            /* the first word of a sequence is shared with its predecessor unless it starts on a word boundary */
            if (first && (d0 & 15u)) atomicOr(&A.dst_packed[gb >> 4], acc);
            else A.dst_packed[gb >> 4] = acc;
            first = false;
            acc = 0;
        }
        gb++;
    }
    if ((uint32_t)gb & 15u) atomicOr(&A.dst_packed[gb >> 4], acc);
}

__global__ void unpack_kernel(const uint32_t *packed, uint64_t gp0, uint64_t n, uint8_t *out)
{
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; /* 16 bases per thread */
    if (i >= n) return;
    const uint32_t w = load_bases16(packed, gp0 + i);
    const uint32_t lut = 0x54474341u; /* "ACGT" little-endian */
    const uint64_t m = n - i < 16 ? n - i : 16;
    for (uint64_t j = 0; j < m; j++) out[i + j] = (uint8_t)(lut >> (8u * ((w >> (2u * (uint32_t)j)) & 3u)));
}
