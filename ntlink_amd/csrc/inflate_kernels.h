/*
 * DEFLATE (RFC 1951) on the device, one LANE per BGZF member -- VERDICT r3 item 6, an experiment with a kill criterion (slower than
 * the host's libdeflate on the box's 16 granted cores: 13.6 GB/s of FASTQ text, profiles/r03ah_gz_diag.txt).
 *
 * Reads arrive as `gzip -cd FILES | indexlr ...` (ntLink:113-117,222).  A BGZF (`bgzip`) file is a chain of independent gzip members
 * of at most 64 KB of text each, and a 2-Gbases FASTQ file holds some 60 000 of them: enough for every lane of the device to
 * inflate a member of its own, with no cooperation between lanes at all.  A lane keeps the two canonical-Huffman count tables of the
 * current block in registers (the compare ladder over the code lengths is straight-line code) and the two symbol tables in LDS
 * (640 B per lane: three wavefronts per CU); bits come from a 64-bit buffer refilled four bytes at a time; matches are copied from
 * the member's own output (no member refers to another).
 *
 * ntl_bgzf_inflate (ntl_hip.hip) wraps it for the measurement and the parity test against zlib; nothing in the product path calls
 * it yet.
 */
#pragma once
#include "dev_common.h"

#define INF_MAXBITS 15
#define INF_LSYM 288
#define INF_DSYM 32

struct InflateArgs {
    const uint8_t *comp;        /* the compressed file (padded behind its end) */
    const uint64_t *in_off;     /* [n + 1] first byte of each member's DEFLATE stream (behind the gzip header); [i + 1] bounds it */
    const uint64_t *out_off;    /* [n + 1] where each member's text goes */
    uint8_t *out;
    uint32_t n;
    uint32_t *status;           /* [n] 0 = ok */
    uint8_t *scratch;           /* [n x 320] the code lengths of a dynamic block's header while its two codes are built */
};

struct InfBits {
    const uint8_t *p;
    uint64_t buf;
    uint32_t cnt;
    __device__ __forceinline__ void refill()
    {
        if (cnt <= 32u) {
            const uint32_t w = ntl_load_u32_a1(p); /* four bytes from any address in one load */
            buf |= (uint64_t)w << cnt;
            p += 4;
            cnt += 32u;
        }
    }
    __device__ __forceinline__ uint32_t take(uint32_t nbits) /* nbits <= 16, after a refill */
    {
        const uint32_t v = (uint32_t)buf & ((1u << nbits) - 1u);
        buf >>= nbits;
        cnt -= nbits;
        return v;
    }
};

/* a canonical Huffman code: count[len] codes of each length in registers, the symbols in code order in `sym` */
struct InfCode {
    uint16_t count[INF_MAXBITS + 1];
};

/* puff.c's decode(): walk the lengths, one compare per length; the bits are peeked from the buffer, not taken one by one */
__device__ __forceinline__ int inf_decode(InfBits &B, const InfCode &H, const uint16_t *sym)
{
    B.refill();
    uint32_t bits = (uint32_t)B.buf; /* the next 32 bits, LSB first */
    int code = 0, first = 0, index = 0;
#pragma unroll
    for (int len = 1; len <= INF_MAXBITS; len++) {
        code |= (int)(bits & 1u);
        bits >>= 1;
        const int count = H.count[len];
        if (code - count < first) {
            B.buf >>= len;
            B.cnt -= (uint32_t)len;
            return sym[index + (code - first)];
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

/* count[] and the symbol order from a list of code lengths (puff.c's construct()); returns < 0 for an over-subscribed set */
__device__ __forceinline__ int inf_construct(InfCode &H, uint16_t *sym, const uint8_t *length, int n)
{
#pragma unroll
    for (int len = 0; len <= INF_MAXBITS; len++) H.count[len] = 0;
    for (int s = 0; s < n; s++) {
        const int l = length[s];
#pragma unroll
        for (int len = 0; len <= INF_MAXBITS; len++)
            if (l == len) H.count[len]++;
    }
    int left = 1;
    uint16_t offs[INF_MAXBITS + 1];
    offs[0] = 0; offs[1] = 0;
#pragma unroll
    for (int len = 1; len <= INF_MAXBITS; len++) {
        left <<= 1;
        left -= H.count[len];
        if (len < INF_MAXBITS) offs[len + 1] = (uint16_t)(offs[len] + H.count[len]);
    }
    if (left < 0) return -1;
    for (int s = 0; s < n; s++) {
        const int l = length[s];
        if (l) {
            uint16_t at = 0;
#pragma unroll
            for (int len = 1; len <= INF_MAXBITS; len++)
                if (l == len) { at = offs[len]; offs[len]++; }
            sym[at] = (uint16_t)s;
        }
    }
    return left;
}

#define INF_NT 64
static __device__ const uint16_t inf_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static __device__ const uint8_t inf_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static __device__ const uint16_t inf_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static __device__ const uint8_t inf_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static __device__ const uint8_t inf_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

/* LDS per lane: the symbols of the two codes in code order (rows of an odd number of words: lanes that read the same index do
   not meet in a bank); the code lengths of a dynamic header live in global scratch (touched once per block) */
__global__ __launch_bounds__(INF_NT) void bgzf_inflate_kernel(InflateArgs A)
{
    __shared__ uint16_t s_lsym[INF_NT][INF_LSYM + 2];
    __shared__ uint16_t s_dsym[INF_NT][INF_DSYM + 2];
    const uint32_t i = blockIdx.x * INF_NT + threadIdx.x;
    if (i >= A.n) return;
    uint16_t *lsym = s_lsym[threadIdx.x], *dsym = s_dsym[threadIdx.x];
    uint8_t *lengths = A.scratch + (uint64_t)i * (INF_LSYM + INF_DSYM);
    InfBits B;
    B.p = A.comp + A.in_off[i];
    B.buf = 0; B.cnt = 0;
    uint8_t *const out0 = A.out + A.out_off[i];
    uint8_t *out = out0;
    uint8_t *const out_end = A.out + A.out_off[i + 1];
    uint32_t err = 0;
    InfCode L, D;
    for (int last = 0; !last && !err;) {
        B.refill();
        last = (int)B.take(1);
        const uint32_t type = B.take(2);
        if (type == 0) { /* stored: skip to a byte boundary, LEN, NLEN, bytes */
            B.take(B.cnt & 7u);
            B.refill();
            const uint32_t len = B.take(16);
            B.refill();
            const uint32_t nlen = B.take(16);
            if ((len ^ 0xFFFFu) != nlen) { err = 2; break; }
            /* the bit buffer holds whole bytes now: give them back */
            B.p -= B.cnt >> 3;
            B.buf = 0; B.cnt = 0;
            if (out + len > out_end) { err = 3; break; }
            for (uint32_t j = 0; j < len; j++) out[j] = B.p[j];
            out += len;
            B.p += len;
            continue;
        }
        if (type == 3) { err = 4; break; }
        if (type == 1) { /* fixed codes */
            for (int s = 0; s < 144; s++) lengths[s] = 8;
            for (int s = 144; s < 256; s++) lengths[s] = 9;
            for (int s = 256; s < 280; s++) lengths[s] = 7;
            for (int s = 280; s < INF_LSYM; s++) lengths[s] = 8;
            inf_construct(L, lsym, lengths, INF_LSYM);
            for (int s = 0; s < 30; s++) lengths[s] = 5;
            inf_construct(D, dsym, lengths, 30);
        } else { /* dynamic: the code-length code, then the two sets of lengths */
            B.refill();
            const int nlen = (int)B.take(5) + 257, ndist = (int)B.take(5) + 1, ncode = (int)B.take(4) + 4;
            if (nlen > 286 || ndist > 30) { err = 5; break; }
            for (int s = 0; s < 19; s++) lengths[s] = 0;
            for (int s = 0; s < ncode; s++) { B.refill(); lengths[inf_order[s]] = (uint8_t)B.take(3); }
            InfCode C;
            uint16_t *csym = dsym; /* 19 symbols: the distance table's words are free until it is built */
            if (inf_construct(C, csym, lengths, 19) < 0) { err = 6; break; }
            int idx = 0;
            while (idx < nlen + ndist) {
                int s = inf_decode(B, C, csym);
                if (s < 0) { err = 7; break; }
                if (s < 16) lengths[idx++] = (uint8_t)s;
                else {
                    int rep, val = 0;
                    B.refill();
                    if (s == 16) {
                        if (idx == 0) { err = 8; break; }
                        val = lengths[idx - 1];
                        rep = 3 + (int)B.take(2);
                    } else if (s == 17) rep = 3 + (int)B.take(3);
                    else rep = 11 + (int)B.take(7);
                    if (idx + rep > nlen + ndist) { err = 9; break; }
                    while (rep--) lengths[idx++] = (uint8_t)val;
                }
            }
            if (err) break;
            if (lengths[256] == 0) { err = 10; break; }
            /* the distance lengths sit behind the literal/length ones: build the distance code first from a copy in registers?  They
               do not overlap the symbol arrays (own LDS array), so the order is free */
            if (inf_construct(L, lsym, lengths, nlen) < 0) { err = 11; break; }
            if (inf_construct(D, dsym, lengths + nlen, ndist) < 0) { err = 12; break; }
        }
        /* the symbols of the block */
        for (;;) {
            int s = inf_decode(B, L, lsym);
            if (s < 0) { err = 13; break; }
            if (s < 256) {
                if (out >= out_end) { err = 14; break; }
                *out++ = (uint8_t)s;
            } else if (s == 256) break;
            else {
                s -= 257;
                if (s >= 29) { err = 15; break; }
                B.refill();
                const uint32_t len = inf_lbase[s] + B.take(inf_lext[s]);
                const int ds = inf_decode(B, D, dsym);
                if (ds < 0 || ds >= 30) { err = 16; break; }
                B.refill();
                const uint32_t dist = inf_dbase[ds] + B.take(inf_dext[ds]);
                if (dist > (uint32_t)(out - out0) || out + len > out_end) { err = 17; break; }
                const uint8_t *from = out - dist;
                uint32_t j = 0;
                if (dist >= 8u) /* source and destination of an eight-byte step do not overlap: eight bytes per load / store */
                    for (; j + 8u <= len; j += 8u) ntl_store_u64_a1(out + j, ntl_load_u64_a1(from + j));
                for (; j < len; j++) out[j] = from[j]; /* the tail, and runs (dist < 8) byte by byte */
                out += len;
            }
        }
    }
    if (!err && out != out_end) err = 18;
    A.status[i] = err;
}
