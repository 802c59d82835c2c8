/*
 * Host-side native I/O of the pair stage (no GPU code): FASTA/FASTQ(.gz) ingest and the text
 * emitters, so that the Python host never loops over bases, minimizers or hits.
 *
 *   ntl_fastx_*        gzip -cd + SeqReader of the reference's pipe (ntLink:113-117,222-223); record
 *                      semantics of bin/read_fasta.py:6-46 (id = header up to the first whitespace,
 *                      multi-line sequences joined, FASTQ qualities skipped by length).  Plain files
 *                      are read with parallel preads, gzip files up to 1-16 GiB (by host memory) inflated in
 *                      one go (libdeflate when present),
 *                      and byte ranges cut at record boundaries are parsed by several threads straight
 *                      into the caller's (page-locked) arrays; larger gzip files and stdin stream
 *                      through zlib on one thread.
 *   ntl_write_indexlr  the TSV `indexlr --long --pos --strand [--len]` prints (ntLink:199,223).
 *   ntl_write_verbose  <prefix>.verbose_mapping.tsv lines (bin/ntlink_pair.py:308-313,382-388).
 *   ntl_write_paf      <prefix>.paf lines (bin/ntlink_paf_output.py:131-135).
 *
 * Formatting is split over threads by record ranges; every thread fills its own buffer and the
 * buffers are written in order.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/vfs.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ntlink_amd.h"

/* ------------------------------------------------------------------ reader -------------- */

/* One byte range of the current block: counted by ntl_fastx_next, parsed into place by ntl_fastx_copy. */
struct Range {
    const char *b = nullptr, *e = nullptr;
    bool at_eof = false;
    uint64_t stop_bases = 0;
    uint64_t nrec = 0, bases = 0, name_bytes = 0;
    bool bad_end = false; /* ended inside a FASTQ quality section: e was not a record boundary */
};

/*
 * The reader works on blocks of input bytes that are cut into per-thread ranges.  Where the bytes
 * come from:
 *   whole    the entire input is in memory (`map`): a gzip file inflated in one go
 *   file     a plain regular file: a mapping of the page cache (or, NTL_IO_PREAD=1, block by block with parallel pread()s)
 *   serial   a pipe / stdin, or a gzip stream too large to inflate whole: read() (through zlib's
 *            inflate() when the data starts with the gzip magic) on the calling thread
 */
struct ntl_fastx {
    const char *map = nullptr; /* whole */
    size_t map_size = 0, map_cap = 0;
    int fd = -1;               /* file / serial */
    bool seekable = false;
    size_t file_size = 0;      /* file: end of the bytes this reader covers (the file's size, or the end of its range) */
    size_t range_lo = 0;       /* file: first byte it covers */
    const char *mm = nullptr;  /* file: mapping of the whole file */
    size_t mm_len = 0;
    bool mm_tried = false;
    int on_tmpfs = -1;         /* file: lives in a tmpfs (no read-ahead to ask for); -1 = not looked at yet */
    bool gz = false, z_init = false, src_eof = false; /* serial */
    z_stream zs;
    std::vector<unsigned char> zin;
    size_t zin_pos = 0, zin_have = 0;
    size_t cur = 0;            /* logical offset of the next unread byte of the (inflated) input */
    char *stage = nullptr;     /* from the buffer cache: pages already faulted in by an earlier file */
    size_t stage_cap = 0;
    size_t stage_off = 0, stage_have = 0; /* stage[0 .. stage_have) = input bytes from stage_off on */
    /* bgzf: a BGZF (bgzip) file, or a range of its members: inflated group by group, the members of a group in parallel, from a
       mapping of the compressed file into `stage` (the serial source's buffer and bookkeeping).  Text offsets (cur, stage_off)
       count from the first byte of member z_lo. */
    bool bgzf = false;
    const unsigned char *zmm = nullptr;
    size_t zmm_len = 0;
    size_t z_lo = 0, z_stop = (size_t)-1; /* compressed offsets of the first member and of the first member behind the range */
    size_t z_pos = 0;                     /* next member to inflate */
    size_t stop_text = (size_t)-1;        /* text offset at which member z_stop begins (known once z_pos has reached it) */
    size_t text_end = (size_t)-1;         /* first record start at or behind stop_text: where this reader's records end */
    bool fastq = false;
    std::vector<Range> ranges;
    /* contiguous copies of the current batch for the pointer accessors */
    std::vector<char> m_seqs, m_names;
    std::vector<uint64_t> m_off, m_name_off;
    bool materialized = false;
    std::vector<std::vector<uint32_t>> pk_seq, pk_start, pk_len; /* ACGT runs of the current batch, per range (ntl_fastx_copy_packed) */
    /* one-pass form (ntl_fastx_next_span / _parse_span / _copy_span): per range where its bases go in the packed stream, and
       what the single pass collected */
    std::vector<uint64_t> sp_pos0, sp_bound;                 /* first base position of a range, bases it has room for */
    std::vector<std::vector<uint64_t>> sp_rec_pos;           /* per range: position of every record's first base */
    std::vector<std::vector<uint32_t>> sp_rec_len;
    std::vector<std::string> sp_names;
    std::vector<std::vector<uint32_t>> sp_name_end;          /* per range: end of every record's id in sp_names[range] */
    size_t sp_advance = 0;                                   /* input bytes the span covers (cur moves when the parse succeeded) */
    uint64_t sp_positions = 0;                               /* base positions the packed stream spans (gaps included) */
    std::string err;
};

/* A pipe end of ours gets the largest buffer an unprivileged process may ask for (1 MiB by default
 * instead of 64 KiB): the reference's recipe joins its operators with pipes (ntLink:221-225), and their
 * throughput is set by how often the two sides have to be woken up. */
static void widen_pipe(int fd)
{
#ifdef F_SETPIPE_SZ
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISFIFO(st.st_mode)) (void)fcntl(fd, F_SETPIPE_SZ, 1 << 20);
#else
    (void)fd;
#endif
}

/* CPU time this process is granted, in cores: its cgroup's quota (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1) when
 * there is one below the CPUs it may run on.  A container often sees every CPU of its host (256 on the GPU boxes) and is granted
 * sixteen: thread pools sized by the former spend the latter on being throttled. */
static double cpu_quota_cores()
{
    double q = 0.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[32] = {0};
        double period = 0.0;
        if (fscanf(f, "%31s %lf", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) q = atof(a) / period;
        fclose(f);
    } else {
        double quota = -1.0, period = 0.0;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lf", &quota) != 1) quota = -1.0; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lf", &period) != 1) period = 0.0; fclose(g); }
        if (quota > 0 && period > 0) q = quota / period;
    }
    return q;
}

static unsigned io_threads()
{
    if (const char *e = getenv("NTL_IO_THREADS")) { int v = atoi(e); if (v > 0) return (unsigned)std::min(v, 256); }
    static const unsigned n_default = [] {
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
        n = std::min(n, 32u);
        /* under a quota: one and a half threads per granted core (measured on the 16-core grant of the GPU box, file to file:
           16 threads 24.4, 20 29.7, 24 31.5, 28 30.1, 32 27.0 Gbases/s; profiles/r03ae_e2e_steady_state.jsonl) */
        const double q = cpu_quota_cores();
        if (q > 0 && q * 1.5 < (double)n) n = std::max(4u, (unsigned)(q * 1.5 + 0.5));
        return n;
    }();
    return n_default;
}

/*
 * Parallel regions run on a process-wide pool of worker threads that is created once: starting and joining 32
 * std::threads costs 3-4 ms per region on a 256-core host (measured: tools/io_diag.py), three regions per read batch, which
 * was most of the reader's time per 256-Mbase batch.  Several callers may be inside run_threads at once (the reader thread
 * and the two text emitters of the pair driver); a caller works through the queue itself while it waits, so regions
 * never wait for each other's workers.  A forked child starts with an empty pool of its own.
 */
struct WorkPool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    std::vector<std::thread> workers;
    bool stop = false;

    explicit WorkPool(unsigned n)
    {
        for (unsigned i = 0; i < n; i++) workers.emplace_back([this] { loop(); });
    }
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this] { return stop || !q.empty(); });
                if (stop && q.empty()) return;
                job = std::move(q.front());
                q.pop_front();
            }
            job();
        }
    }
    bool help() /* the caller of a region runs queued jobs too */
    {
        std::function<void()> job;
        {
            std::lock_guard<std::mutex> lk(m);
            if (q.empty()) return false;
            job = std::move(q.front());
            q.pop_front();
        }
        job();
        return true;
    }
};

/* Two pools: the parsers (FASTA/FASTQ reader, TSV parser) and the text emitters each queue their regions on their own, so a
 * batch of long formatting jobs never sits in front of the reader's short passes (the caller of a region helps only its own pool). */
static WorkPool *g_pool[2] = {nullptr, nullptr};
static std::mutex g_pool_mutex;
static thread_local int t_pool_kind = 0; /* 0 parse, 1 emit */
struct PoolKind {
    int saved;
    explicit PoolKind(int k) : saved(t_pool_kind) { t_pool_kind = k; }
    ~PoolKind() { t_pool_kind = saved; }
};

static WorkPool *work_pool()
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    WorkPool *&P = g_pool[t_pool_kind];
    if (!P) {
        static bool atfork_set = false;
        if (!atfork_set) {
            atfork_set = true;
            pthread_atfork(nullptr, nullptr, [] { g_pool[0] = g_pool[1] = nullptr; new (&g_pool_mutex) std::mutex(); }); /* threads do not survive fork() */
        }
        unsigned hc = std::thread::hardware_concurrency();
        if (hc == 0) hc = 4;
        P = new WorkPool(std::min(hc, std::max(io_threads(), 4u))); /* never destroyed: workers idle on the condition variable */
    }
    return P;
}

template <typename F>
static void run_threads(size_t n, F work)
{
    if (n <= 1) { if (n) work((size_t)0); return; }
    WorkPool *P = work_pool();
    struct Region { std::mutex m; std::condition_variable cv; size_t left; } R;
    R.left = n - 1;
    {
        std::lock_guard<std::mutex> lk(P->m);
        for (size_t t = 1; t < n; t++)
            P->q.emplace_back([&work, &R, t] {
                work(t);
                std::lock_guard<std::mutex> lk2(R.m);
                if (--R.left == 0) R.cv.notify_all();
            });
    }
    P->cv.notify_all();
    work((size_t)0);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(R.m);
            if (R.left == 0) break;
        }
        if (!P->help()) { /* everything left of this region is running on workers */
            std::unique_lock<std::mutex> lk(R.m);
            R.cv.wait(lk, [&R] { return R.left == 0; });
            break;
        }
    }
}

static inline bool id_blank(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\f' || ch == '\v'; }

/* id = header up to the first whitespace; str.split(None, 1) skips leading blanks (bin/read_fasta.py:22) */
static inline void id_span(const char *h, size_t hl, size_t &a, size_t &b)
{
    a = 1;
    while (a < hl && (h[a] == ' ' || h[a] == '\t')) a++;
    b = a;
    while (b < hl && !id_blank(h[b])) b++;
}

/* where parse_range puts what it finds: counters only, or the caller's arrays */
struct CountSink {
    uint64_t nrec = 0, nbases = 0, name_bytes = 0;
    void id(const char *, size_t n) { name_bytes += n; }
    void seq(const char *, size_t n) { nbases += n; }
    void end_record() { nrec++; }
    uint64_t bases() const { return nbases; }
};
struct WriteSink {
    char *seqs; uint64_t *off; char *names; uint64_t *name_off; /* off / name_off point at this range's first record */
    uint64_t b0, n0;                                             /* global positions of the range's first base / name byte */
    uint64_t nb = 0, nn = 0, i = 0;
    void id(const char *p, size_t n) { memcpy(names + n0 + nn, p, n); nn += n; name_off[i + 1] = n0 + nn; }
    void seq(const char *p, size_t n) { memcpy(seqs + b0 + nb, p, n); nb += n; }
    void end_record() { off[i + 1] = b0 + nb; i++; }
    uint64_t bases() const { return nb; }
};

/* The second pass with the device's layout as its output: 2 bits per base (A/a 0, C/c 1, G/g 2, T/t 3, anything else 0),
 * sixteen bases per 32-bit word, base b of the batch at global position NTL_PACK_LEAD + b -- and the table of maximal
 * ACGT runs per sequence that keeps every other byte out of the k-mers.  Exactly what pack_kernel / run_*_kernel
 * (pack_kernels.h) derive from the ASCII bytes on the device; done here, a quarter of the bytes cross PCIe.
 * Eight bases per step: validity and codes with byte-parallel arithmetic, then the eight 2-bit codes are gathered into
 * sixteen bits.  A thread owns the words that lie wholly inside its range; the first and the last word of a range may be
 * shared with its neighbours and are OR-ed in atomically (the caller zeroes those words first). */
#define NTL_PACK_LEAD 16u   /* = NTL_LEAD_PAD of the device code (dev_common.h) */
#define NTL_PACK_END 4096u  /* = NTL_END_PAD (ntl_hip.hip) */

/* 32 bases per step where the CPU has AVX2 + BMI2 (every x86 host of the last decade): four byte compares against A C G T (case
 * folded) decide validity, PEXT gathers bits 1-2 of every byte -- the 2-bit code up to the swap of G and T -- sixteen bits per
 * eight bases.  Returns 1 and the 64 packed bits when all 32 bytes are ACGT/acgt, else 0 (the caller's 8-byte / 1-byte steps
 * take it from there).  The parser threads spend most of their time here: on the GPU boxes the process is granted 16 cores. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,bmi2"))) static int ntl_pack32(const char *p, uint64_t *out)
{
    const __m256i x = _mm256_loadu_si256((const __m256i *)p);
    const __m256i u = _mm256_and_si256(x, _mm256_set1_epi8((char)0xDF));
    const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8(0x41)), _mm256_cmpeq_epi8(u, _mm256_set1_epi8(0x43))),
                                       _mm256_or_si256(_mm256_cmpeq_epi8(u, _mm256_set1_epi8(0x47)), _mm256_cmpeq_epi8(u, _mm256_set1_epi8(0x54))));
    if ((uint32_t)_mm256_movemask_epi8(ok) != 0xFFFFFFFFu) return 0;
    const uint64_t M = 0x0606060606060606ull;
    uint64_t v = _pext_u64((uint64_t)_mm256_extract_epi64(x, 0), M) | (_pext_u64((uint64_t)_mm256_extract_epi64(x, 1), M) << 16) |
                 (_pext_u64((uint64_t)_mm256_extract_epi64(x, 2), M) << 32) | (_pext_u64((uint64_t)_mm256_extract_epi64(x, 3), M) << 48);
    v ^= (v >> 1) & 0x5555555555555555ull; /* 0 1 3 2 -> 0 1 2 3 */
    *out = v;
    return 1;
}
/* 64 bases per step where the CPU has AVX-512BW (the GPU boxes' EPYC 9575F; round 6): the four compares give the validity as one
 * 64-bit mask; the codes -- bits 1-2 of every byte, G and T swapped back -- are folded pairwise by two multiply-adds (c0 + 4 c1 in
 * sixteen bits, then w0 + 16 w1 in thirty-two: four bases per byte) and the low bytes of the sixteen dwords are the 128 packed bits. */
__attribute__((target("avx512f,avx512bw"))) static int ntl_pack64(const char *p, uint64_t out[2])
{
    const __m512i x = _mm512_loadu_si512((const void *)p);
    const __m512i u = _mm512_and_si512(x, _mm512_set1_epi8((char)0xDF));
    const __mmask64 ok = _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8(0x41)) | _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8(0x43)) |
                         _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8(0x47)) | _mm512_cmpeq_epi8_mask(u, _mm512_set1_epi8(0x54));
    if (ok != ~(__mmask64)0) return 0;
    __m512i v = _mm512_and_si512(_mm512_srli_epi16(x, 1), _mm512_set1_epi8(3));                  /* 0 1 3 2 */
    v = _mm512_xor_si512(v, _mm512_and_si512(_mm512_srli_epi16(v, 1), _mm512_set1_epi8(1)));      /* 0 1 2 3 */
    const __m512i w16 = _mm512_maddubs_epi16(v, _mm512_set1_epi16(0x0401));                       /* c0 + 4 c1 */
    const __m512i w32 = _mm512_madd_epi16(w16, _mm512_set1_epi32(0x00100001));                    /* w0 + 16 w1 */
    const __m128i r = _mm512_cvtepi32_epi8(w32);
    _mm_storeu_si128((__m128i *)out, r);
    return 1;
}
/* which packer steps this CPU has: 0 the 8-byte arithmetic only, 1 + AVX2 / BMI2 (32 bases), 2 + AVX-512BW (64 bases).  NTL_IO_SIMD=
 * none | avx2 | avx512 caps it (tests, A/B; read where a parser range begins), NTL_IO_NO_SIMD = none. */
static int ntl_pack_level()
{
    static const int hw = [] {
        if (!(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2"))) return 0;
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") ? 2 : 1;
    }();
    if (getenv("NTL_IO_NO_SIMD")) return 0;
    const char *e = getenv("NTL_IO_SIMD");
    if (!e) return hw;
    if (!strcmp(e, "none") || !strcmp(e, "0")) return 0;
    if (!strcmp(e, "avx2")) return hw < 1 ? hw : 1;
    return hw;
}
#else
static int ntl_pack32(const char *, uint64_t *) { return 0; }
static int ntl_pack64(const char *, uint64_t *) { return 0; }
static int ntl_pack_level() { return 0; }
#endif

struct PackSink {
    uint32_t *packed; uint64_t *off; char *names; uint64_t *name_off;
    uint64_t b0, n0;
    std::vector<uint32_t> *run_seq, *run_start, *run_len; /* runs of this range: record (range-local), offset in it, length */
    uint64_t nb = 0, nn = 0, i = 0;
    uint64_t acc = 0, widx = 0, first_widx = 0;
    unsigned fill = 0;
    bool first_partial = false, in_run = false;
    uint64_t seq_nb0 = 0, run_nb0 = 0;
    int simd = 0;

    void begin()
    {
        simd = ntl_pack_level();
        const uint64_t g0 = NTL_PACK_LEAD + b0;
        widx = first_widx = g0 >> 4;
        fill = 2u * (unsigned)(g0 & 15u);
        first_partial = fill != 0;
    }
    inline void push(uint64_t bits, unsigned nbits)
    {
        acc |= bits << fill;
        fill += nbits;
        while (fill >= 32) {
            const uint32_t w = (uint32_t)acc;
            if (widx == first_widx && first_partial) __atomic_fetch_or(&packed[widx], w, __ATOMIC_RELAXED);
            else packed[widx] = w;
            acc >>= 32; fill -= 32; widx++;
        }
    }
    void finish()
    {
        if (fill) __atomic_fetch_or(&packed[widx], (uint32_t)acc, __ATOMIC_RELAXED);
    }
    inline void close_run(uint64_t at)
    {
        if (!in_run) return;
        run_seq->push_back((uint32_t)i);
        run_start->push_back((uint32_t)(run_nb0 - seq_nb0));
        run_len->push_back((uint32_t)(at - run_nb0));
        in_run = false;
    }
    void id(const char *p, size_t n) { memcpy(names + n0 + nn, p, n); nn += n; name_off[i + 1] = n0 + nn; }
    void seq(const char *p, size_t n)
    {
        const uint64_t K1 = 0x0101010101010101ull, K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
        size_t k = 0;
        while (k < n) {
            if (simd >= 2 && n - k >= 64) {
                uint64_t v128[2];
                if (ntl_pack64(p + k, v128)) {
                    if (!in_run) { in_run = true; run_nb0 = nb + k; }
                    push(v128[0] & 0xFFFFFFFFull, 32);
                    push(v128[0] >> 32, 32);
                    push(v128[1] & 0xFFFFFFFFull, 32);
                    push(v128[1] >> 32, 32);
                    k += 64;
                    continue;
                }
            }
            if (simd >= 1 && n - k >= 32) {
                uint64_t v64;
                if (ntl_pack32(p + k, &v64)) {
                    if (!in_run) { in_run = true; run_nb0 = nb + k; }
                    push(v64 & 0xFFFFFFFFull, 32);
                    push(v64 >> 32, 32);
                    k += 32;
                    continue;
                }
            }
            if (n - k >= 8) {
                uint64_t x;
                memcpy(&x, p + k, 8);
                const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull; /* fold the case */
                auto zero_bytes = [&](uint64_t z) { return ~(((z & K7F) + K7F) | z | K7F); }; /* 0x80 in every zero byte, exact */
                const uint64_t ok = zero_bytes(u ^ (0x41 * K1)) | zero_bytes(u ^ (0x43 * K1)) | zero_bytes(u ^ (0x47 * K1)) | zero_bytes(u ^ (0x54 * K1));
                if (ok == K80) {
                    uint64_t v = (x >> 1) & (3 * K1);
                    v ^= (v >> 1) & K1;                                   /* 0 1 3 2 -> 0 1 2 3 */
                    v = (v | (v >> 6)) & 0x000F000F000F000Full;
                    v = (v | (v >> 12)) & 0x000000FF000000FFull;
                    v = (v | (v >> 24)) & 0xFFFFull;
                    if (!in_run) { in_run = true; run_nb0 = nb + k; }
                    push(v, 16);
                    k += 8;
                    continue;
                }
            }
            const uint32_t c = (uint8_t)p[k], uc = c & 0xDFu;
            const bool okc = uc == 0x41u || uc == 0x43u || uc == 0x47u || uc == 0x54u;
            const uint32_t t = (c >> 1) & 3u;
            push(okc ? ((t ^ (t >> 1)) & 3u) : 0u, 2);
            if (okc) { if (!in_run) { in_run = true; run_nb0 = nb + k; } }
            else close_run(nb + k);
            k++;
        }
        nb += n;
    }
    void end_record() { close_run(nb); off[i + 1] = b0 + nb; i++; seq_nb0 = nb; }
    uint64_t bases() const { return nb; }
};

/*
 * The record state machine of bin/read_fasta.py:6-46 over the bytes [p, e): a line starting with
 * '>' or '@' is a header unless it falls inside a quality section; sequence lines run to the next
 * line starting with '>', '@' or '+'; after '+', quality lines are skipped until they are as long as
 * the sequence (at least one line).  stop_bases != 0: stop after the record that brings the range
 * to that many bases.  Returns where the next record (or e) starts; *bad_end is set when the range
 * ends inside a quality section.
 */
template <typename Sink>
static const char *parse_range(const char *p, const char *e, uint64_t stop_bases, bool at_eof, Sink &c, bool *bad_end)
{
    auto line = [&](const char *&q, size_t &len) -> bool {
        if (p >= e) return false;
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        q = p; len = (size_t)(le - p);
        p = nl ? nl + 1 : e;
        if (len && q[len - 1] == '\r') len--;
        return true;
    };
    const char *q = nullptr, *h = nullptr;
    size_t len = 0, hl = 0;
    bool have_hdr = false;
    for (;;) {
        if (!have_hdr) {
            bool found = false;
            while (line(q, len))
                if (len && (q[0] == '>' || q[0] == '@')) { h = q; hl = len; found = true; break; }
            if (!found) break;
        }
        have_hdr = false;
        size_t ia, ib;
        id_span(h, hl, ia, ib);
        c.id(h + ia, ib - ia);
        bool plus = false;
        const uint64_t s0 = c.bases();
        while (line(q, len)) {
            if (len && (q[0] == '>' || q[0] == '@' || q[0] == '+')) {
                if (q[0] == '+') plus = true;
                else { h = q; hl = len; have_hdr = true; }
                break;
            }
            c.seq(q, len);
        }
        const uint64_t slen = c.bases() - s0;
        c.end_record();
        if (plus) {
            uint64_t got = 0;
            bool done = false;
            while (line(q, len)) {
                got += len;
                if (got >= slen) { done = true; break; }
            }
            if (!done && !at_eof && bad_end) *bad_end = true;
        }
        if (stop_bases && c.bases() >= stop_bases) break;
    }
    return have_hdr ? h : p;
}

/* First line start in [p, e) that begins a record: '>' for FASTA; for FASTQ an '@' line followed by
 * sequence-looking lines (they start with a letter, '*', '-' or '.') up to a '+' line.  A quality line
 * may start with '@' too, but what follows it inside a quality section is, sooner or later, the next
 * header -- another '@' -- before any '+' separator.  A wrong guess (possible with wrapped quality
 * lines) is caught by Range::bad_end and the batch is re-read without cuts. */
static const char *find_boundary(const char *p, const char *e, bool fastq)
{
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        if (!nl || nl + 1 >= e) return e;
        p = nl + 1;
        if (!fastq) {
            if (*p == '>') return p;
            const char *g = (const char *)memchr(p, '>', (size_t)(e - p)); /* skip whole lines at memchr speed */
            if (!g) return e;
            if (g[-1] == '\n') return g;
            p = g;
        } else if (*p == '@') {
            const char *q = p;
            for (;;) { /* q: start of a line after the candidate header */
                const char *n1 = (const char *)memchr(q, '\n', (size_t)(e - q));
                if (!n1 || n1 + 1 >= e) return e; /* cannot tell this close to the end */
                q = n1 + 1;
                const char ch = *q;
                if (ch == '+') return p;
                const bool seq_like = (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') || ch == '*' || ch == '-' || ch == '.' ||
                                      ch == '\r' || ch == '\n';
                if (!seq_like) break;
            }
        }
    }
    return e;
}

/* libdeflate (whole-buffer inflate, about three times zlib's speed) is used when the image has it;
 * there is no header for it here, so the three entry points are bound by hand. */
struct LibDeflate {
    void *(*alloc)(void) = nullptr;
    int (*gzip_ex)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    bool ok = false;
    LibDeflate()
    {
        if (getenv("NTL_IO_NO_LIBDEFLATE")) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        alloc = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
        gzip_ex = (int (*)(void *, const void *, size_t, void *, size_t, size_t *, size_t *))dlsym(h, "libdeflate_gzip_decompress_ex");
        release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        ok = alloc && gzip_ex && release;
    }
};
static const LibDeflate &libdeflate() { static LibDeflate L; return L; }

/* Inflate buffers are kept for the next file: fresh pages cost more than the inflate itself. */
struct BufCache {
    std::mutex mu;
    std::vector<std::pair<char *, size_t>> free_list;
    char *take(size_t want, size_t *cap)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            int pick = -1;
            for (size_t i = 0; i < free_list.size(); i++)
                if (free_list[i].second >= want && (pick < 0 || free_list[i].second < free_list[(size_t)pick].second)) pick = (int)i;
            if (pick >= 0) {
                char *p = free_list[(size_t)pick].first;
                *cap = free_list[(size_t)pick].second;
                free_list.erase(free_list.begin() + pick);
                return p;
            }
        }
        *cap = want;
        return (char *)malloc(want);
    }
    void give(char *p, size_t cap)
    {
        std::lock_guard<std::mutex> g(mu);
        free_list.emplace_back(p, cap);
        while (free_list.size() > 8) { /* drop the smallest */
            size_t m = 0;
            for (size_t i = 1; i < free_list.size(); i++) if (free_list[i].second < free_list[m].second) m = i;
            free(free_list[m].first);
            free_list.erase(free_list.begin() + (long)m);
        }
    }
};
static BufCache &buf_cache() { static BufCache c; return c; }

/* BGZF (bgzip, htslib): every member carries its own compressed size in a 'BC' extra field and its inflated size
 * in its trailer, so the member table is read by hopping over the file and the members inflate independently.
 * Fills the offsets of every member in the file and in the output; false when the data is not BGZF throughout. */
static bool bgzf_table(const unsigned char *in, size_t n, std::vector<size_t> &moff, std::vector<size_t> &ooff)
{
    size_t pos = 0, opos = 0;
    while (pos < n) {
        if (n - pos < 18 + 8 || in[pos] != 0x1f || in[pos + 1] != 0x8b || in[pos + 2] != 8 || !(in[pos + 3] & 4)) return false;
        const size_t xlen = (size_t)in[pos + 10] | ((size_t)in[pos + 11] << 8);
        if (pos + 12 + xlen > n) return false;
        size_t bsize = 0;
        for (size_t q = pos + 12; q + 4 <= pos + 12 + xlen;) { /* extra subfields: SI1 SI2 SLEN data */
            const size_t slen = (size_t)in[q + 2] | ((size_t)in[q + 3] << 8);
            if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2 && q + 6 <= pos + 12 + xlen) bsize = ((size_t)in[q + 4] | ((size_t)in[q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || pos + bsize > n) return false;
        uint32_t isize;
        memcpy(&isize, in + pos + bsize - 4, 4);
        moff.push_back(pos); ooff.push_back(opos);
        pos += bsize; opos += isize;
    }
    moff.push_back(pos); ooff.push_back(opos);
    return moff.size() > 2;
}

/* header of the BGZF member at `pos`: its size in the file and the size of its text; false when there is no valid member */
static bool bgzf_member(const unsigned char *in, size_t n, size_t pos, size_t *bsize_out, size_t *isize_out)
{
    if (pos + 18 + 8 > n || in[pos] != 0x1f || in[pos + 1] != 0x8b || in[pos + 2] != 8 || !(in[pos + 3] & 4)) return false;
    const size_t xlen = (size_t)in[pos + 10] | ((size_t)in[pos + 11] << 8);
    if (pos + 12 + xlen > n) return false;
    size_t bsize = 0;
    for (size_t q = pos + 12; q + 4 <= pos + 12 + xlen;) {
        const size_t slen = (size_t)in[q + 2] | ((size_t)in[q + 3] << 8);
        if (in[q] == 'B' && in[q + 1] == 'C' && slen == 2 && q + 6 <= pos + 12 + xlen) bsize = ((size_t)in[q + 4] | ((size_t)in[q + 5] << 8)) + 1;
        q += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || pos + bsize > n) return false;
    uint32_t isize;
    memcpy(&isize, in + pos + bsize - 4, 4);
    if (isize > (1u << 16)) return false; /* BGZF members hold at most 64 KiB of text */
    *bsize_out = bsize; *isize_out = isize;
    return true;
}

/* First member start at or behind `pos` (n when there is none): the four magic bytes, a well-formed header, and two more
 * well-formed members (or the end of the file) behind it -- compressed data that happens to look like one header does not
 * look like three in a row. */
static size_t bgzf_member_at_or_after(const unsigned char *in, size_t n, size_t pos)
{
    if (pos == 0) return 0;
    for (size_t q = pos; q + 26 <= n; q++) {
        if (in[q] != 0x1f) {
            const unsigned char *f = (const unsigned char *)memchr(in + q, 0x1f, n - q);
            if (!f) return n;
            q = (size_t)(f - in);
            if (q + 26 > n) return n;
        }
        size_t at = q, ok = 0;
        for (; ok < 3 && at < n; ok++) {
            size_t bs, is;
            if (!bgzf_member(in, n, at, &bs, &is)) break;
            at += bs;
        }
        if (ok == 3 || (ok > 0 && at == n)) return q;
    }
    return n;
}

struct BgzfMember { size_t pos, bsize, isize, out; };

/* inflates the listed members (each an independent gzip member) to dst + out, several at a time */
static bool bgzf_inflate(const unsigned char *in, const std::vector<BgzfMember> &ms, char *dst)
{
    const size_t nm = ms.size();
    if (!nm) return true;
    const LibDeflate &L = libdeflate();
    const size_t T = std::min<size_t>(io_threads(), std::max<size_t>(1, nm / 4));
    std::vector<int> bad(T, 0);
    run_threads(T, [&](size_t t) {
        void *d = L.ok ? L.alloc() : nullptr;
        z_stream zs;
        bool z_on = false;
        for (size_t i = nm * t / T; i < nm * (t + 1) / T; i++) {
            const BgzfMember &m = ms[i];
            if (d) {
                size_t ain = 0, aout = 0;
                if (L.gzip_ex(d, in + m.pos, m.bsize, dst + m.out, m.isize, &ain, &aout) != 0 || aout != m.isize) { bad[t] = 1; break; }
                continue;
            }
            if (!z_on) { /* no libdeflate in this image: zlib, one stream object per thread */
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) { bad[t] = 1; break; }
                z_on = true;
            } else inflateReset(&zs);
            zs.next_in = (Bytef *)(in + m.pos); zs.avail_in = (uInt)m.bsize;
            zs.next_out = (Bytef *)(dst + m.out); zs.avail_out = (uInt)m.isize;
            const int rc = inflate(&zs, Z_FINISH);
            if ((rc != Z_STREAM_END && !(rc == Z_OK && m.isize == 0)) || zs.avail_out != 0) { bad[t] = 1; break; }
        }
        if (d) L.release(d);
        if (z_on) inflateEnd(&zs);
    });
    for (int x : bad) if (x) return false;
    return true;
}

/* Inflates every gzip member of [in, in + n) into one malloc'd buffer (`gzip -cd` of a whole file held in
 * memory); BGZF members in parallel.  False on corrupt data or when libdeflate is missing; the caller then
 * streams through zlib. */
static bool inflate_whole(const unsigned char *in, size_t n, char **out, size_t *out_n, size_t *out_cap)
{
    const LibDeflate &L = libdeflate();
    if (!L.ok || n < 18) return false;
    {
        std::vector<size_t> moff, ooff;
        if (bgzf_table(in, n, moff, ooff)) {
            const size_t nm = moff.size() - 1, total = ooff[nm];
            size_t cap = 0;
            char *buf = buf_cache().take(total + 64, &cap);
            if (!buf) return false;
            const size_t T = std::min<size_t>(io_threads(), std::max<size_t>(1, nm / 16));
            std::vector<int> bad(T, 0);
            run_threads(T, [&](size_t t) {
                void *d = L.alloc();
                if (!d) { bad[t] = 1; return; }
                for (size_t i = nm * t / T; i < nm * (t + 1) / T; i++) {
                    size_t ain = 0, aout = 0;
                    const size_t want = ooff[i + 1] - ooff[i];
                    if (L.gzip_ex(d, in + moff[i], moff[i + 1] - moff[i], buf + ooff[i], want, &ain, &aout) != 0 || aout != want) { bad[t] = 1; break; }
                }
                L.release(d);
            });
            bool ok = true;
            for (int x : bad) ok &= !x;
            if (ok) {
                if (getenv("NTL_IO_TRACE")) fprintf(stderr, "ntl_fastx: BGZF, %zu members inflated on %zu threads\n", nm, T);
                *out = buf; *out_n = total; *out_cap = cap;
                return true;
            }
            buf_cache().give(buf, cap); /* not what the table promised: take the general path */
        }
    }
    void *d = L.alloc();
    if (!d) return false;
    uint32_t isize;
    memcpy(&isize, in + n - 4, 4); /* exact for a single member below 4 GiB; a starting point otherwise */
    size_t cap = 0;
    char *buf = buf_cache().take(std::max<size_t>((size_t)isize + 64, 3 * n + (1u << 16)), &cap);
    size_t ipos = 0, opos = 0;
    bool good = buf != nullptr;
    while (good && ipos < n) {
        if (in[ipos] == 0) { ipos++; continue; } /* zero padding between / after members, as gzip tolerates */
        size_t ain = 0, aout = 0;
        const int rc = L.gzip_ex(d, in + ipos, n - ipos, buf + opos, cap - opos, &ain, &aout);
        if (rc == 3) { /* LIBDEFLATE_INSUFFICIENT_SPACE: grow and retry this member */
            cap *= 2;
            char *nb = (char *)realloc(buf, cap);
            if (!nb) { good = false; break; }
            buf = nb;
            continue;
        }
        if (rc != 0) { good = false; break; }
        ipos += ain; opos += aout;
    }
    L.release(d);
    if (!good) { free(buf); return false; }
    *out = buf; *out_n = opos; *out_cap = cap;
    return true;
}

static void bgzf_fill(ntl_fastx *r, size_t need);

static void set_format(ntl_fastx *r, const char *p, const char *e)
{
    while (p < e && (*p == '\n' || *p == '\r')) p++; /* format = first header character */
    r->fastq = p < e && *p == '@';
}

extern "C" int ntl_fastx_open(const char *path, ntl_fastx **out)
{
    if (!path || !out) return NTL_EINVAL;
    *out = nullptr;
    const bool is_stdin = strcmp(path, "-") == 0;
    const int fd = is_stdin ? dup(0) : open(path, O_RDONLY);
    if (fd < 0) return NTL_EINVAL;
    ntl_fastx *r = new ntl_fastx();
    r->fd = fd;
    widen_pipe(fd);
    struct stat st;
    const bool regular = !is_stdin && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    const bool serial_only = getenv("NTL_IO_NO_MMAP") != nullptr; /* tests: every input through the serial source */
    unsigned char head[64];
    ssize_t hn = 0;
    if (regular && st.st_size > 0 && !serial_only && (hn = pread(fd, head, sizeof head, 0)) >= 2) {
        const bool gz = head[0] == 0x1f && head[1] == 0x8b;
        /* compressed bytes up to which a gzip file is inflated in one go (the inflated text, ~4x, is then held in
           memory): 1/40 of the physical memory, between 1 and 16 GiB */
        size_t whole_max = (size_t)1 << 30;
        {
            const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
            if (pages > 0 && psz > 0) whole_max = std::min<size_t>(std::max<size_t>((size_t)pages / 40 * (size_t)psz, (size_t)1 << 30), (size_t)16 << 30);
        }
        if (const char *e = getenv("NTL_IO_GZ_WHOLE_MAX")) whole_max = (size_t)atoll(e);
        if (!gz) { /* file source: parallel pread */
            r->seekable = true;
            r->file_size = (size_t)st.st_size;
            set_format(r, (const char *)head, (const char *)head + hn);
            *out = r;
            return NTL_OK;
        }
        {   /* BGZF: members inflate independently -- any size, bounded memory, several threads, and ranges of members can be
               read on their own (ntl_fastx_open_range) */
            size_t bs = 0, is = 0;
            void *m = (hn >= 28 && (head[3] & 4)) ? mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
            if (m != MAP_FAILED) (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
            if (m != MAP_FAILED && bgzf_member((const unsigned char *)m, (size_t)st.st_size, 0, &bs, &is) && !getenv("NTL_IO_NO_BGZF")) {
                r->bgzf = true;
                r->zmm = (const unsigned char *)m; r->zmm_len = (size_t)st.st_size;
                /* the format is that of the file's first record, wherever a range of it starts */
                std::vector<char> first(is + 64);
                std::vector<BgzfMember> one{{0, bs, is, 0}};
                if (!bgzf_inflate(r->zmm, one, first.data())) { ntl_fastx_close(r); return NTL_EINVAL; }
                set_format(r, first.data(), first.data() + std::min<size_t>(is, 64));
                *out = r;
                return NTL_OK;
            }
            if (m != MAP_FAILED) munmap(m, (size_t)st.st_size);
        }
        if ((size_t)st.st_size <= whole_max) {
            void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                char *buf = nullptr; size_t bn = 0, bcap = 0;
                const bool ok = inflate_whole((const unsigned char *)m, (size_t)st.st_size, &buf, &bn, &bcap);
                munmap(m, (size_t)st.st_size);
                if (ok) { /* whole source */
                    r->map = buf; r->map_size = bn; r->map_cap = bcap;
                    set_format(r, buf, buf + std::min<size_t>(bn, 64));
                    close(fd);
                    r->fd = -1;
                    *out = r;
                    return NTL_OK;
                }
            }
        }
    }
    /* serial source: the format and the gzip magic are looked at when the first bytes arrive */
    r->zin.resize(1u << 20);
    *out = r;
    return NTL_OK;
}

/* First record start at or behind byte `at` of a plain file (a record belongs to the range that holds its first byte:
 * two readers opened on [a, b) and [b, c) see every record exactly once); the file size when there is none. */
static bool record_start_at(int fd, size_t file_size, bool fastq, size_t at, size_t *out)
{
    if (at == 0) { *out = 0; return true; }
    if (at >= file_size) { *out = file_size; return true; }
    size_t win = (size_t)1 << 20;
    std::vector<char> buf;
    for (;;) {
        const size_t from = at - 1; /* the byte before: `at` itself is a record start when it follows a newline */
        const size_t len = std::min(win, file_size - from);
        buf.resize(len);
        size_t have = 0;
        while (have < len) {
            const ssize_t n = pread(fd, buf.data() + have, len - have, (off_t)(from + have));
            if (n <= 0) return false;
            have += (size_t)n;
        }
        const char *b = buf.data(), *e = b + len;
        const char *p = find_boundary(b, e, fastq);
        if (p < e) { *out = from + (size_t)(p - b); return true; }
        if (from + len >= file_size) { *out = file_size; return true; }
        win *= 4; /* a record (or the look-ahead of the FASTQ test) longer than the window */
    }
}

/* A reader over the records of a plain (uncompressed, regular) file whose first byte lies in [lo, hi); hi = 0 or beyond
 * the end means the end of the file.  NTL_EINVAL for anything that cannot be cut (gzip, pipes): such inputs are read whole
 * by one reader.  The multi-GPU driver gives every rank its own byte range of the read files. */
extern "C" int ntl_fastx_open_range(const char *path, uint64_t lo, uint64_t hi, ntl_fastx **out)
{
    if (!path || !out) return NTL_EINVAL;
    *out = nullptr;
    if (strcmp(path, "-") == 0) return NTL_EINVAL;
    ntl_fastx *r = nullptr;
    int rc = ntl_fastx_open(path, &r);
    if (rc != NTL_OK) return rc;
    if (r->bgzf) { /* lo / hi are offsets in the COMPRESSED file: the members that start in [lo, hi) */
        const size_t n = r->zmm_len;
        if (hi == 0 || hi > n) hi = n;
        if (lo > hi) lo = hi;
        r->z_lo = bgzf_member_at_or_after(r->zmm, n, (size_t)lo);
        r->z_stop = hi >= n ? n : bgzf_member_at_or_after(r->zmm, n, (size_t)hi);
        if (r->z_stop < r->z_lo) r->z_stop = r->z_lo;
        r->z_pos = r->z_lo;
        if (r->z_lo > 0) { /* the first record of the range: the first record start behind the first line end of its text */
            size_t need = (size_t)1 << 20;
            for (;;) {
                bgzf_fill(r, need);
                if (!r->err.empty()) { ntl_fastx_close(r); return NTL_EINVAL; }
                const char *b = r->stage, *e = r->stage + r->stage_have;
                const char *q = b < e ? find_boundary(b, e, r->fastq) : e;
                if (q < e || r->src_eof) { r->cur = (size_t)(q - b); break; }
                need *= 4;
            }
            if (r->text_end != (size_t)-1 && r->cur > r->text_end) r->cur = r->text_end;
        }
        *out = r;
        return NTL_OK;
    }
    if (!r->seekable) { ntl_fastx_close(r); return NTL_EINVAL; }
    const size_t size = r->file_size;
    if (hi == 0 || hi > size) hi = size;
    if (lo > hi) lo = hi;
    size_t a = 0, b = 0;
    if (!record_start_at(r->fd, size, r->fastq, (size_t)lo, &a) || !record_start_at(r->fd, size, r->fastq, (size_t)hi, &b)) {
        ntl_fastx_close(r);
        return NTL_EINVAL;
    }
    if (b < a) b = a;
    r->range_lo = a; r->cur = a; r->stage_off = a;
    r->file_size = b;
    *out = r;
    return NTL_OK;
}

/* The byte range [*lo, *hi) of the file this reader covers (0, 0 for sources that are not plain files). */
extern "C" void ntl_fastx_range(const ntl_fastx *r, uint64_t *lo, uint64_t *hi)
{
    if (r && r->bgzf) { /* compressed bytes of the members this reader owns */
        if (lo) *lo = r->z_lo;
        if (hi) *hi = r->z_stop == (size_t)-1 ? r->zmm_len : r->z_stop;
        return;
    }
    if (lo) *lo = r && r->seekable ? r->range_lo : 0;
    if (hi) *hi = r && r->seekable ? r->file_size : 0;
}

extern "C" void ntl_fastx_close(ntl_fastx *r)
{
    if (!r) return;
    if (r->z_init) inflateEnd(&r->zs);
    if (r->map) buf_cache().give((char *)r->map, r->map_cap);
    /* The page-table entries of a multi-GB mapping are dropped under the mapping lock held for READING first (MADV_DONTNEED; the page
       cache keeps its pages): munmap, which holds it for writing -- every page fault of the process waits, the other reader's
       among them -- then finds nothing to drop. */
    if (r->mm) { (void)madvise((void *)r->mm, r->mm_len, MADV_DONTNEED); munmap((void *)r->mm, r->mm_len); }
    if (r->zmm) { (void)madvise((void *)r->zmm, r->zmm_len, MADV_DONTNEED); munmap((void *)r->zmm, r->zmm_len); }
    if (r->fd >= 0) close(r->fd);
    if (r->stage) buf_cache().give(r->stage, r->stage_cap);
    delete r;
}

extern "C" const char *ntl_fastx_error(const ntl_fastx *r) { return r ? r->err.c_str() : "no reader"; }

/* serial source: up to n bytes of the (inflated) input into dst; 0 at the end or on an error (r->err) */
static size_t serial_read(ntl_fastx *r, char *dst, size_t n)
{
    if (r->src_eof || n == 0) return 0;
    auto refill = [&]() -> bool { /* compressed / raw bytes from the descriptor into zin */
        const ssize_t got = read(r->fd, r->zin.data(), r->zin.size());
        if (got < 0) { r->err = "read error"; r->src_eof = true; return false; }
        r->zin_pos = 0; r->zin_have = (size_t)got;
        return got > 0;
    };
    if (r->cur == 0 && r->stage_have == 0 && !r->z_init && r->zin_have == 0) { /* very first call: sniff */
        if (!refill()) { r->src_eof = true; return 0; }
        r->gz = r->zin_have >= 2 && r->zin[0] == 0x1f && r->zin[1] == 0x8b;
        if (r->gz) {
            memset(&r->zs, 0, sizeof r->zs);
            if (inflateInit2(&r->zs, 16 + MAX_WBITS) != Z_OK) { r->err = "zlib init failed"; r->src_eof = true; return 0; }
            r->z_init = true;
        }
    }
    if (!r->gz) {
        if (r->zin_pos < r->zin_have) { /* bytes left from the sniff */
            const size_t m = std::min(n, r->zin_have - r->zin_pos);
            memcpy(dst, r->zin.data() + r->zin_pos, m);
            r->zin_pos += m;
            return m;
        }
        const ssize_t got = read(r->fd, dst, n);
        if (got < 0) { r->err = "read error"; r->src_eof = true; return 0; }
        if (got == 0) r->src_eof = true;
        return (size_t)got;
    }
    size_t done = 0;
    while (done < n) {
        if (r->zin_pos == r->zin_have && !refill()) {
            if (r->err.empty()) r->err = "unexpected end of gzip data"; /* a member was cut short */
            r->src_eof = true;
            break;
        }
        r->zs.next_in = r->zin.data() + r->zin_pos;
        r->zs.avail_in = (uInt)(r->zin_have - r->zin_pos);
        r->zs.next_out = (Bytef *)dst + done;
        r->zs.avail_out = (uInt)std::min<size_t>(n - done, (size_t)1 << 30);
        const size_t out_before = r->zs.avail_out;
        const int rc = inflate(&r->zs, Z_NO_FLUSH);
        r->zin_pos = r->zin_have - r->zs.avail_in;
        done += out_before - r->zs.avail_out;
        if (rc == Z_STREAM_END) { /* next member, if any (gzip -cd concatenates them) */
            while (r->zin_pos < r->zin_have && r->zin[r->zin_pos] == 0) r->zin_pos++; /* zero padding */
            if (r->zin_pos == r->zin_have && !refill()) { r->src_eof = true; break; }
            inflateReset(&r->zs);
        } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
            r->err = r->zs.msg ? r->zs.msg : "corrupt gzip data";
            r->src_eof = true;
            break;
        }
    }
    return done;
}

static bool stage_reserve(ntl_fastx *r, size_t want)
{
    if (r->stage_cap >= want) return true;
    size_t cap = 0;
    char *nb = buf_cache().take(want + want / 8, &cap);
    if (!nb) { r->err = "out of memory"; return false; }
    if (r->stage_have) memcpy(nb, r->stage, r->stage_have);
    if (r->stage) buf_cache().give(r->stage, r->stage_cap);
    r->stage = nb; r->stage_cap = cap;
    return true;
}

/* BGZF source: inflates members behind z_pos into the stage buffer until it holds `need` bytes of text (or the source ends).
 * A ranged reader (z_stop set) ends at the first record start at or behind the text of member z_stop -- `find_boundary` from
 * there, the same function of the same bytes that gives the NEXT range its first record, so that neighbouring ranges see every
 * record once; the members needed to find it are inflated as far as it takes. */
static void bgzf_fill(ntl_fastx *r, size_t need)
{
    const size_t NONE = (size_t)-1;
    for (;;) {
        if (r->src_eof) return;
        if (r->text_end != NONE) { /* the closing record start is known: nothing behind it belongs to this reader */
            if (r->cur > r->text_end) r->cur = r->text_end; /* the range's first record starts behind its end: it owns nothing */
            const size_t keep = r->text_end > r->stage_off ? r->text_end - r->stage_off : 0;
            if (r->stage_have > keep) r->stage_have = keep;
            r->src_eof = true;
            return;
        }
        const bool tail = r->stop_text != NONE; /* behind the range: small steps, only to find the closing record start */
        if (!tail && r->stage_have >= need) return;
        if (!tail && r->z_stop != NONE && r->z_pos >= r->z_stop) { r->stop_text = r->stage_off + r->stage_have; continue; }
        if (r->z_pos >= r->zmm_len) { /* end of the file */
            if (r->z_stop != NONE) { r->text_end = r->stage_off + r->stage_have; continue; }
            r->src_eof = true;
            return;
        }
        std::vector<BgzfMember> ms;
        size_t total = 0, pos = r->z_pos;
        const size_t limit = !tail && r->z_stop != NONE ? std::min(r->z_stop, r->zmm_len) : r->zmm_len;
        const size_t want = tail ? (size_t)1 << 20 : std::max<size_t>(need - r->stage_have, (size_t)4 << 20);
        bool foreign = false;
        while (pos < limit && total < want) {
            size_t bs, is;
            if (!bgzf_member(r->zmm, r->zmm_len, pos, &bs, &is)) { foreign = true; break; }
            ms.push_back({pos, bs, is, total});
            total += is; pos += bs;
        }
        if (foreign && ms.empty()) {
            /* not a BGZF member: zero padding, or an ordinary gzip member somebody concatenated (`cat a.bgz b.gz`; gzip -cd reads
               that too) -- inflated as a stream, on this thread, to wherever it ends */
            if (r->zmm[pos] == 0) { r->z_pos = pos + 1; continue; }
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (r->zmm[pos] != 0x1f || inflateInit2(&zs, 16 + MAX_WBITS) != Z_OK) { r->err = "corrupt gzip data"; r->src_eof = true; return; }
            zs.next_in = (Bytef *)(r->zmm + pos);
            size_t in_left = r->zmm_len - pos;
            int rc = Z_OK;
            while (rc != Z_STREAM_END) {
                if (r->stage_cap - r->stage_have < (1u << 16) && !stage_reserve(r, std::max<size_t>(r->stage_cap * 2, (size_t)8 << 20))) { inflateEnd(&zs); r->src_eof = true; return; }
                zs.avail_in = (uInt)std::min<size_t>(in_left, (size_t)1 << 30);
                const uInt in0 = zs.avail_in;
                zs.next_out = (Bytef *)(r->stage + r->stage_have);
                zs.avail_out = (uInt)std::min<size_t>(r->stage_cap - r->stage_have, (size_t)1 << 30);
                const uInt out0 = zs.avail_out;
                rc = inflate(&zs, Z_NO_FLUSH);
                in_left -= in0 - zs.avail_in;
                r->stage_have += out0 - zs.avail_out;
                if (rc != Z_OK && rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && zs.avail_out == 0)) {
                    inflateEnd(&zs);
                    r->err = zs.msg ? zs.msg : "corrupt gzip data"; r->src_eof = true;
                    return;
                }
                if (rc != Z_STREAM_END && in_left == 0 && zs.avail_in == 0) { inflateEnd(&zs); r->err = "unexpected end of gzip data"; r->src_eof = true; return; }
            }
            inflateEnd(&zs);
            r->z_pos = r->zmm_len - in_left;
            continue;
        }
        if (!stage_reserve(r, r->stage_have + total + 64)) { r->src_eof = true; return; }
        if (!bgzf_inflate(r->zmm, ms, r->stage + r->stage_have)) { r->err = "corrupt BGZF data"; r->src_eof = true; return; }
        r->stage_have += total;
        r->z_pos = pos;
        if (tail) {
            const char *b0 = r->stage + (r->stop_text - r->stage_off), *e = r->stage + r->stage_have;
            const char *q = b0 < e ? find_boundary(b0, e, r->fastq) : e;
            if (q < e) r->text_end = r->stage_off + (size_t)(q - r->stage);
        }
    }
}

/* Makes the input bytes [cur, cur + need) (clipped to the end) addressable; returns their start, *avail = how
 * many there are, *at_eof = they reach the end of the input. */
static const char *view(ntl_fastx *r, size_t need, size_t *avail, bool *at_eof)
{
    *avail = 0; *at_eof = true;
    if (r->map) {
        *avail = r->map_size - r->cur;
        return r->map + r->cur;
    }
    if (r->stage_off != r->cur) { /* keep what was read beyond the previous batch */
        const size_t end = r->stage_off + r->stage_have;
        if (r->cur >= r->stage_off && r->cur < end) {
            memmove(r->stage, r->stage + (r->cur - r->stage_off), end - r->cur);
            r->stage_have = end - r->cur;
        } else r->stage_have = 0;
        r->stage_off = r->cur;
    }
    if (r->seekable) {
        const size_t remain = r->file_size - r->cur;
        /* A plain file is parsed straight from a mapping of the page cache: no staging copy (pread of a 260 MB batch by 32
           threads took 9 of the reader's 15 ms per batch on tmpfs; faulting the mapping in, 64 KB per fault, is part of the
           2.5 ms the counting pass now takes).  The bytes behind the batch are requested ahead for files that are not
           cached.  NTL_IO_PREAD=1 keeps the staged pread path (tests, file systems without mmap). */
        if (!r->mm && !r->mm_tried) {
            r->mm_tried = true;
            if (!getenv("NTL_IO_PREAD")) {
                struct stat st;
                const size_t len = fstat(r->fd, &st) == 0 ? (size_t)st.st_size : r->file_size;
                void *m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, r->fd, 0);
                if (m != MAP_FAILED) {
                    r->mm = (const char *)m; r->mm_len = len;
                    /* read once, front to back.  Also: pages of a mapping that is NOT marked so are marked accessed when it is
                       unmapped, which for pages a process has just written (a freshly copied input in tmpfs) means 8.6 M moves to the
                       active list for 35 GB, under the LRU lock and inside munmap (profiles/r04_first_pass.txt: pgactivate) */
                    (void)madvise(m, len, MADV_SEQUENTIAL);
                }
            }
        }
        if (r->mm) {
            const size_t target = std::min(need, remain);
            *avail = target;
            *at_eof = target == remain;
            /* Ask for the bytes behind the batch ahead of time -- unless the file lives in memory already (tmpfs: there the call
               only walks the page cache of the range, 3-4 ms per 256-MB batch on the reader's critical path, measured) */
            if (target < remain) {
                if (r->on_tmpfs < 0) {
                    struct statfs sf;
                    r->on_tmpfs = fstatfs(r->fd, &sf) == 0 && (unsigned long)sf.f_type == 0x01021994ul ? 1 : 0;
                }
                if (!r->on_tmpfs) (void)posix_fadvise(r->fd, (off_t)(r->cur + target), (off_t)std::min(remain - target, target), POSIX_FADV_WILLNEED);
            }
            return r->mm + r->cur;
        }
        const size_t target = std::min(need, remain);
        if (r->stage_have < target) {
            if (!stage_reserve(r, target)) return r->stage;
            const size_t from = r->stage_have, len = target - from;
            const size_t T = std::min<size_t>(io_threads(), std::max<size_t>(1, len / (4u << 20)));
            std::vector<int> bad(T, 0);
            run_threads(T, [&](size_t t) {
                size_t a = from + len / T * t, b = t + 1 == T ? from + len : from + len / T * (t + 1);
                while (a < b) {
                    const ssize_t n = pread(r->fd, r->stage + a, b - a, (off_t)(r->stage_off + a));
                    if (n <= 0) { bad[t] = 1; return; }
                    a += (size_t)n;
                }
            });
            for (int x : bad) if (x) r->err = "read error";
            r->stage_have = target;
        }
        *avail = target;
        *at_eof = target == remain;
        return r->stage;
    }
    if (r->bgzf) {
        bgzf_fill(r, need);
        if (r->cur < r->stage_off) { r->stage_off = r->cur; r->stage_have = 0; } /* the range ends in front of its first record start: it owns nothing */
        *avail = std::min(r->stage_have, need);
        *at_eof = r->src_eof && *avail == r->stage_have;
        return r->stage;
    }
    /* serial: read on until `need` bytes are there or the source ends */
    while (r->stage_have < need && !r->src_eof) {
        if (r->stage_cap == r->stage_have) { /* full (or not there yet): a bounded need is reserved at once */
            size_t want = need != (size_t)-1 ? need : std::max<size_t>(r->stage_cap * 2, (size_t)64 << 20);
            if (want <= r->stage_cap) want = r->stage_cap * 2;
            if (!stage_reserve(r, want)) break;
        }
        const size_t room = std::min(r->stage_cap, need) - r->stage_have;
        const bool first = r->cur == 0 && r->stage_have == 0;
        const size_t got = serial_read(r, r->stage + r->stage_have, room);
        if (first && got) set_format(r, r->stage, r->stage + std::min<size_t>(got, 64));
        r->stage_have += got;
        if (got == 0) break;
    }
    *avail = std::min(r->stage_have, need);
    *at_eof = r->src_eof && *avail == r->stage_have;
    return r->stage;
}

/* Cuts the byte range of the next batch into per-thread ranges and counts their records. */
static void next_blocks(ntl_fastx *r, uint64_t max_bases)
{
    r->ranges.clear();
    const uint64_t want0 = max_bases ? max_bases * (r->fastq ? 2u : 1u) + (max_bases >> 6) : 0;
    size_t need = max_bases ? (size_t)want0 + (1u << 20) : (size_t)-1;
    size_t avail; bool at_eof;
    const char *p0 = view(r, need, &avail, &at_eof);
    if (!r->err.empty() || avail == 0) return;
    /* the format is known once the first bytes are there (serial source): FASTQ takes two bytes per base */
    const uint64_t want = max_bases ? max_bases * (r->fastq ? 2u : 1u) + (max_bases >> 6) : 0;
    if (max_bases && want > want0 && !at_eof) {
        need = (size_t)want + (1u << 20);
        p0 = view(r, need, &avail, &at_eof);
        if (!r->err.empty()) return;
    }
    size_t end = avail; /* about max_bases bases, cut at a record boundary */
    if (max_bases && want < avail) {
        for (;;) {
            end = (size_t)(find_boundary(p0 + want, p0 + avail, r->fastq) - p0);
            if (end < avail || at_eof) break;
            need = avail * 2; /* a record longer than what was read beyond the cut: read on */
            p0 = view(r, need, &avail, &at_eof);
            if (!r->err.empty()) return;
        }
    }
    const char *pe = p0 + end, *fe = p0 + avail;
    const size_t span = end;
    size_t min_chunk = 2u << 20; /* bytes per thread below which more threads do not pay */
    if (const char *e = getenv("NTL_IO_MIN_CHUNK")) { long v = atol(e); if (v > 0) min_chunk = (size_t)v; }
    const unsigned T = (unsigned)std::min<size_t>(io_threads(), std::max<size_t>(1, span / min_chunk));
    r->ranges.resize(T);
    const char *prev = p0;
    for (unsigned t = 0; t < T; t++) {
        const char *nxt = t + 1 == T ? pe : std::max(prev, find_boundary(p0 + span / T * (t + 1), pe, r->fastq));
        r->ranges[t].b = prev; r->ranges[t].e = nxt; r->ranges[t].at_eof = nxt == fe && at_eof;
        prev = nxt;
    }
    auto count = [&](size_t t) {
        Range &g = r->ranges[t];
        CountSink cs;
        parse_range(g.b, g.e, g.stop_bases, g.at_eof, cs, &g.bad_end);
        g.nrec = cs.nrec; g.bases = cs.nbases; g.name_bytes = cs.name_bytes;
    };
    run_threads(T, count);
    bool bad = false;
    for (auto &g : r->ranges) bad |= g.bad_end;
    if (bad) { /* a cut fell inside a quality section (wrapped FASTQ): one range, no cuts */
        if (getenv("NTL_IO_TRACE")) fprintf(stderr, "ntl_fastx: range cut inside a quality section, batch re-read on one thread\n");
        for (;;) {
            r->ranges.resize(1);
            Range &g = r->ranges[0];
            g = Range();
            g.b = p0; g.e = p0 + avail; g.at_eof = at_eof; g.stop_bases = max_bases;
            CountSink cs;
            pe = parse_range(g.b, g.e, g.stop_bases, at_eof, cs, nullptr);
            g.nrec = cs.nrec; g.bases = cs.nbases; g.name_bytes = cs.name_bytes;
            if (at_eof || pe < p0 + avail) break; /* stopped on a complete record */
            p0 = view(r, avail * 2, &avail, &at_eof);
            if (!r->err.empty()) return;
        }
    }
    r->cur += (size_t)(pe - p0);
}

extern "C" void ntl_fastx_sizes(const ntl_fastx *r, uint64_t *nseq, uint64_t *bases, uint64_t *name_bytes)
{
    uint64_t n = 0, b = 0, nb = 0;
    if (r) for (auto &g : r->ranges) { n += g.nrec; b += g.bases; nb += g.name_bytes; }
    if (nseq) *nseq = n;
    if (bases) *bases = b;
    if (name_bytes) *name_bytes = nb;
}

/* Collects records until about max_bases bases are held (0 = to the end of the input). */
extern "C" int ntl_fastx_next(ntl_fastx *r, uint64_t max_bases, uint64_t *nseq)
{
    if (!r || !nseq) return NTL_EINVAL;
    r->materialized = false;
    *nseq = 0;
    for (;;) {
        next_blocks(r, max_bases);
        if (!r->err.empty()) return NTL_EINVAL;
        ntl_fastx_sizes(r, nseq, nullptr, nullptr);
        if (*nseq || r->ranges.empty()) return NTL_OK; /* a block without any record (blank lines): read on */
    }
}

/* Puts the current batch into caller-allocated arrays (sizes from ntl_fastx_sizes; offsets and
 * name_offsets have nseq + 1 entries): every range is parsed straight into place by its own thread. */
extern "C" int ntl_fastx_copy(const ntl_fastx *r, char *seqs, uint64_t *offsets, char *names, uint64_t *name_offsets)
{
    if (!r || !offsets || !name_offsets) return NTL_EINVAL;
    uint64_t n, b, nb;
    ntl_fastx_sizes(r, &n, &b, &nb);
    if ((b && !seqs) || (nb && !names)) return NTL_EINVAL;
    offsets[0] = 0; name_offsets[0] = 0;
    const size_t T = r->ranges.size();
    std::vector<uint64_t> rec0(T + 1, 0), b0(T + 1, 0), n0(T + 1, 0);
    for (size_t t = 0; t < T; t++) {
        rec0[t + 1] = rec0[t] + r->ranges[t].nrec;
        b0[t + 1] = b0[t] + r->ranges[t].bases;
        n0[t + 1] = n0[t] + r->ranges[t].name_bytes;
    }
    run_threads(T, [&](size_t t) {
        const Range &g = r->ranges[t];
        WriteSink ws{seqs, offsets + rec0[t], names, name_offsets + rec0[t], b0[t], n0[t]};
        parse_range(g.b, g.e, g.stop_bases, g.at_eof, ws, nullptr);
    });
    return NTL_OK;
}

/* Words of the packed array of a batch of `bases` bases (lead pad, bases, end pad; what ntl_batch_create_packed uploads). */
extern "C" uint64_t ntl_packed_words(uint64_t bases)
{
    return (NTL_PACK_LEAD + bases + NTL_PACK_END + 15) / 16 + 2;
}

/* The current batch in the device's layout: packed[ntl_packed_words(bases)], offsets / names as ntl_fastx_copy; *nruns = number
 * of ACGT runs (ntl_fastx_runs then fills seq_run_first[nseq + 1], run_start[nruns], run_len[nruns]). */
extern "C" int ntl_fastx_copy_packed(ntl_fastx *r, uint32_t *packed, uint64_t *offsets, char *names, uint64_t *name_offsets, uint64_t *nruns)
{
    if (!r || !packed || !offsets || !name_offsets || !nruns) return NTL_EINVAL;
    uint64_t n, b, nb;
    ntl_fastx_sizes(r, &n, &b, &nb);
    if (nb && !names) return NTL_EINVAL;
    offsets[0] = 0; name_offsets[0] = 0;
    const size_t T = r->ranges.size();
    std::vector<uint64_t> rec0(T + 1, 0), b0(T + 1, 0), n0(T + 1, 0);
    for (size_t t = 0; t < T; t++) {
        rec0[t + 1] = rec0[t] + r->ranges[t].nrec;
        b0[t + 1] = b0[t] + r->ranges[t].bases;
        n0[t + 1] = n0[t] + r->ranges[t].name_bytes;
    }
    const uint64_t nwords = ntl_packed_words(b);
    /* words that are OR-ed into or never written by a range: the lead pad, the first word of every range, everything from the
       word that holds the last base on */
    packed[0] = 0;
    for (size_t t = 0; t <= T; t++) packed[(NTL_PACK_LEAD + b0[t]) >> 4] = 0;
    for (uint64_t wd = (NTL_PACK_LEAD + b) >> 4; wd < nwords; wd++) packed[wd] = 0;
    r->pk_seq.assign(T, {}); r->pk_start.assign(T, {}); r->pk_len.assign(T, {});
    run_threads(T, [&](size_t t) {
        const Range &g = r->ranges[t];
        PackSink ps{packed, offsets + rec0[t], names, name_offsets + rec0[t], b0[t], n0[t], &r->pk_seq[t], &r->pk_start[t], &r->pk_len[t]};
        ps.begin();
        parse_range(g.b, g.e, g.stop_bases, g.at_eof, ps, nullptr);
        ps.finish();
    });
    uint64_t nr = 0;
    for (size_t t = 0; t < T; t++) nr += r->pk_seq[t].size();
    *nruns = nr;
    return NTL_OK;
}

extern "C" int ntl_fastx_runs(const ntl_fastx *r, uint32_t *seq_run_first, uint32_t *run_start, uint32_t *run_len)
{
    if (!r || !seq_run_first) return NTL_EINVAL;
    const size_t T = r->ranges.size();
    if (r->pk_seq.size() != T) return NTL_EINVAL; /* ntl_fastx_copy_packed comes first */
    uint64_t rec = 0, run = 0;
    for (size_t t = 0; t < T; t++) {
        const auto &sq = r->pk_seq[t];
        size_t j = 0;
        for (uint64_t i = 0; i < r->ranges[t].nrec; i++) {
            seq_run_first[rec + i] = (uint32_t)(run + j);
            while (j < sq.size() && sq[j] == (uint32_t)i) j++;
        }
        if (!sq.empty()) {
            if (!run_start || !run_len) return NTL_EINVAL;
            memcpy(run_start + run, r->pk_start[t].data(), sq.size() * 4);
            memcpy(run_len + run, r->pk_len[t].data(), sq.size() * 4);
        }
        rec += r->ranges[t].nrec;
        run += sq.size();
    }
    seq_run_first[rec] = (uint32_t)run;
    return NTL_OK;
}

/* ---- one pass: parse straight into place without counting first -------------------------------------------------------
 * The two-pass reader counts a batch (records, bases, id bytes per thread range) so that every thread can write at its final
 * offset.  Both passes walk the input, and on a page-cache input each is bound by the kernel's per-page work, not by the
 * parser: counting costs as much as parsing.  The device does not need the sequences of a batch to be contiguous: a sequence
 * is (position of its first base in the packed stream, length, ACGT runs).  So a range writes its bases at a position that is
 * an UPPER bound of where counting would have put it -- the bytes of the ranges before it (a range holds fewer bases than
 * bytes; half as many for FASTQ, whose qualities are as long as the bases) rounded up to whole packed words -- and collects
 * its records in its own vectors; a range that outgrows its bound (malformed FASTQ) or ends inside a quality section makes
 * the call fail with NTL_ERANGE and the caller reads that batch with the two-pass functions instead. */
struct SpanSink {
    uint32_t *packed;
    uint64_t pos0, bound;                                     /* base position of the range's first base; room */
    std::vector<uint64_t> *rec_pos; std::vector<uint32_t> *rec_len, *name_end;
    std::string *names;
    std::vector<uint32_t> *run_seq, *run_start, *run_len;
    uint64_t nb = 0, i = 0;
    uint64_t acc = 0, widx = 0;
    unsigned fill = 0;
    bool in_run = false, overflow = false;
    uint64_t seq_nb0 = 0, run_nb0 = 0;
    int simd = 0;
    void begin() { simd = ntl_pack_level(); widx = (NTL_PACK_LEAD + pos0) >> 4; fill = 0; } /* pos0 is a multiple of 16: whole words are this range's own */
    inline void push(uint64_t bits, unsigned nbits)
    {
        acc |= bits << fill;
        fill += nbits;
        while (fill >= 32) { packed[widx++] = (uint32_t)acc; acc >>= 32; fill -= 32; }
    }
    void finish() { if (fill) packed[widx] = (uint32_t)acc; }
    inline void close_run(uint64_t at)
    {
        if (!in_run) return;
        run_seq->push_back((uint32_t)i);
        run_start->push_back((uint32_t)(run_nb0 - seq_nb0));
        run_len->push_back((uint32_t)(at - run_nb0));
        in_run = false;
    }
    void id(const char *p, size_t n) { names->append(p, n); name_end->push_back((uint32_t)names->size()); rec_pos->push_back(pos0 + nb); }
    void seq(const char *p, size_t n)
    {
        if (nb + n > bound) { overflow = true; return; }
        const uint64_t K1 = 0x0101010101010101ull, K7F = 0x7F7F7F7F7F7F7F7Full, K80 = 0x8080808080808080ull;
        size_t k = 0;
        while (k < n) {
            if (simd >= 2 && n - k >= 64) {
                uint64_t v128[2];
                if (ntl_pack64(p + k, v128)) {
                    if (!in_run) { in_run = true; run_nb0 = nb + k; }
                    push(v128[0] & 0xFFFFFFFFull, 32);
                    push(v128[0] >> 32, 32);
                    push(v128[1] & 0xFFFFFFFFull, 32);
                    push(v128[1] >> 32, 32);
                    k += 64;
                    continue;
                }
            }
            if (simd >= 1 && n - k >= 32) {
                uint64_t v64;
                if (ntl_pack32(p + k, &v64)) {
                    if (!in_run) { in_run = true; run_nb0 = nb + k; }
                    push(v64 & 0xFFFFFFFFull, 32);
                    push(v64 >> 32, 32);
                    k += 32;
                    continue;
                }
            }
            if (n - k >= 8) {
                uint64_t x;
                memcpy(&x, p + k, 8);
                const uint64_t u = x & 0xDFDFDFDFDFDFDFDFull; /* fold the case */
                auto zero_bytes = [&](uint64_t z) { return ~(((z & K7F) + K7F) | z | K7F); };
                const uint64_t ok = zero_bytes(u ^ (0x41 * K1)) | zero_bytes(u ^ (0x43 * K1)) | zero_bytes(u ^ (0x47 * K1)) | zero_bytes(u ^ (0x54 * K1));
                if (ok == K80) {
                    uint64_t v = (x >> 1) & (3 * K1);
                    v ^= (v >> 1) & K1;
                    v = (v | (v >> 6)) & 0x000F000F000F000Full;
                    v = (v | (v >> 12)) & 0x000000FF000000FFull;
                    v = (v | (v >> 24)) & 0xFFFFull;
                    if (!in_run) { in_run = true; run_nb0 = nb + k; }
                    push(v, 16);
                    k += 8;
                    continue;
                }
            }
            const uint32_t c = (uint8_t)p[k], uc = c & 0xDFu;
            const bool okc = uc == 0x41u || uc == 0x43u || uc == 0x47u || uc == 0x54u;
            const uint32_t t = (c >> 1) & 3u;
            push(okc ? ((t ^ (t >> 1)) & 3u) : 0u, 2);
            if (okc) { if (!in_run) { in_run = true; run_nb0 = nb + k; } }
            else close_run(nb + k);
            k++;
        }
        nb += n;
    }
    void end_record() { close_run(nb); rec_len->push_back((uint32_t)(nb - seq_nb0)); i++; seq_nb0 = nb; }
    uint64_t bases() const { return nb; }
};

/* Cuts the next span of about max_bases bases (at record boundaries, into per-thread ranges, as ntl_fastx_next does) without
 * reading it: *span_bytes = input bytes it covers (0 at the end of the input), *packed_words = words the caller's packed
 * array must hold.  The span becomes the current batch only when ntl_fastx_parse_span succeeds. */
extern "C" int ntl_fastx_next_span(ntl_fastx *r, uint64_t max_bases, uint64_t *span_bytes, uint64_t *packed_words)
{
    if (!r || !span_bytes || !packed_words || !max_bases) return NTL_EINVAL;
    *span_bytes = 0; *packed_words = 0;
    r->materialized = false;
    r->ranges.clear();
    r->sp_advance = 0;
    const uint64_t want0 = max_bases * (r->fastq ? 2u : 1u) + (max_bases >> 6);
    size_t need = (size_t)want0 + (1u << 20);
    size_t avail; bool at_eof;
    const char *p0 = view(r, need, &avail, &at_eof);
    if (!r->err.empty()) return NTL_EINVAL;
    if (avail == 0) return NTL_OK;
    const uint64_t want = max_bases * (r->fastq ? 2u : 1u) + (max_bases >> 6);
    if (want > want0 && !at_eof) {
        need = (size_t)want + (1u << 20);
        p0 = view(r, need, &avail, &at_eof);
        if (!r->err.empty()) return NTL_EINVAL;
    }
    size_t end = avail;
    if (want < avail) {
        for (;;) {
            end = (size_t)(find_boundary(p0 + want, p0 + avail, r->fastq) - p0);
            if (end < avail || at_eof) break;
            need = avail * 2;
            p0 = view(r, need, &avail, &at_eof);
            if (!r->err.empty()) return NTL_EINVAL;
        }
    }
    const char *pe = p0 + end, *fe = p0 + avail;
    size_t min_chunk = 2u << 20;
    if (const char *e = getenv("NTL_IO_MIN_CHUNK")) { long v = atol(e); if (v > 0) min_chunk = (size_t)v; }
    const unsigned T = (unsigned)std::min<size_t>(io_threads(), std::max<size_t>(1, end / min_chunk));
    r->ranges.resize(T);
    r->sp_pos0.assign(T, 0); r->sp_bound.assign(T, 0);
    const char *prev = p0;
    uint64_t pos = 0;
    for (unsigned t = 0; t < T; t++) {
        const char *nxt = t + 1 == T ? pe : std::max(prev, find_boundary(p0 + end / T * (t + 1), pe, r->fastq));
        r->ranges[t].b = prev; r->ranges[t].e = nxt; r->ranges[t].at_eof = nxt == fe && at_eof;
        const uint64_t bytes = (uint64_t)(nxt - prev);
        const uint64_t bound = r->fastq ? bytes / 2 + 1 : bytes; /* a FASTQ record's qualities are as long as its bases */
        r->sp_pos0[t] = pos; r->sp_bound[t] = bound;
        pos += (bound + 15) & ~(uint64_t)15;
        prev = nxt;
    }
    r->sp_positions = pos;
    r->sp_advance = end;
    *span_bytes = end;
    *packed_words = ntl_packed_words(pos);
    return NTL_OK;
}

/* The one pass over the span: bases packed into place, records / ids / ACGT runs collected per range.  NTL_ERANGE: this span
 * cannot be read in one pass (see above): nothing was consumed, call ntl_fastx_next for it. */
extern "C" int ntl_fastx_parse_span(ntl_fastx *r, uint32_t *packed, uint64_t *nseq, uint64_t *bases, uint64_t *name_bytes, uint64_t *nruns)
{
    if (!r || !packed || !nseq || !bases || !name_bytes || !nruns) return NTL_EINVAL;
    const size_t T = r->ranges.size();
    if (!T || !r->sp_advance) return NTL_EINVAL;
    r->sp_rec_pos.assign(T, {}); r->sp_rec_len.assign(T, {}); r->sp_names.assign(T, std::string()); r->sp_name_end.assign(T, {});
    r->pk_seq.assign(T, {}); r->pk_start.assign(T, {}); r->pk_len.assign(T, {});
    const uint64_t nwords = ntl_packed_words(r->sp_positions);
    packed[0] = 0; /* the lead pad */
    for (uint64_t wd = (NTL_PACK_LEAD + r->sp_positions) >> 4; wd < nwords; wd++) packed[wd] = 0;
    std::vector<int> bad(T, 0);
    run_threads(T, [&](size_t t) {
        Range &g = r->ranges[t];
        g.bad_end = false;
        SpanSink ps{packed, r->sp_pos0[t], r->sp_bound[t], &r->sp_rec_pos[t], &r->sp_rec_len[t], &r->sp_name_end[t], &r->sp_names[t],
                    &r->pk_seq[t], &r->pk_start[t], &r->pk_len[t]};
        ps.begin();
        parse_range(g.b, g.e, 0, g.at_eof, ps, &g.bad_end);
        ps.finish();
        g.nrec = ps.i; g.bases = ps.nb; g.name_bytes = r->sp_names[t].size();
        if (ps.overflow || g.bad_end || r->sp_rec_pos[t].size() != r->sp_rec_len[t].size()) bad[t] = 1;
    });
    for (int x : bad) if (x) { r->ranges.clear(); return NTL_ERANGE; }
    uint64_t n = 0, b = 0, nb = 0, nr = 0;
    for (size_t t = 0; t < T; t++) { n += r->ranges[t].nrec; b += r->ranges[t].bases; nb += r->ranges[t].name_bytes; nr += r->pk_seq[t].size(); }
    *nseq = n; *bases = b; *name_bytes = nb; *nruns = nr;
    r->cur += r->sp_advance;
    r->sp_advance = 0;
    return NTL_OK;
}

/* The records of the span: positions[nseq] (first base of every sequence in the packed stream), lengths[nseq], ids, and the
 * ACGT-run table of ntl_fastx_runs.  *span_positions (may be NULL) = base positions the packed stream spans. */
extern "C" int ntl_fastx_copy_span(const ntl_fastx *r, uint64_t *positions, uint32_t *lengths, char *names, uint64_t *name_offsets,
                                   uint32_t *seq_run_first, uint32_t *run_start, uint32_t *run_len, uint64_t *span_positions)
{
    if (!r || !positions || !lengths || !name_offsets || !seq_run_first) return NTL_EINVAL;
    const size_t T = r->ranges.size();
    if (r->sp_rec_pos.size() != T) return NTL_EINVAL;
    uint64_t rec = 0, nb = 0;
    name_offsets[0] = 0;
    for (size_t t = 0; t < T; t++) {
        const size_t m = r->sp_rec_pos[t].size();
        if (m) {
            memcpy(positions + rec, r->sp_rec_pos[t].data(), m * 8);
            memcpy(lengths + rec, r->sp_rec_len[t].data(), m * 4);
        }
        for (size_t i = 0; i < m; i++) name_offsets[rec + i + 1] = nb + r->sp_name_end[t][i];
        if (!r->sp_names[t].empty()) { if (!names) return NTL_EINVAL; memcpy(names + nb, r->sp_names[t].data(), r->sp_names[t].size()); }
        nb += r->sp_names[t].size();
        rec += m;
    }
    if (span_positions) *span_positions = r->sp_positions;
    return ntl_fastx_runs(r, seq_run_first, run_start, run_len);
}

static void materialize(ntl_fastx *r)
{
    if (r->materialized) return;
    uint64_t n, b, nb;
    ntl_fastx_sizes(r, &n, &b, &nb);
    r->m_seqs.resize(b); r->m_names.resize(nb); r->m_off.resize(n + 1); r->m_name_off.resize(n + 1);
    ntl_fastx_copy(r, r->m_seqs.data(), r->m_off.data(), r->m_names.data(), r->m_name_off.data());
    r->materialized = true;
}

/* contiguous views of the current batch (a copy is made on first use; ntl_fastx_copy avoids it) */
extern "C" const char *ntl_fastx_seqs(ntl_fastx *r) { materialize(r); return r->m_seqs.data(); }
extern "C" const uint64_t *ntl_fastx_offsets(ntl_fastx *r) { materialize(r); return r->m_off.data(); }
extern "C" const char *ntl_fastx_names(ntl_fastx *r) { materialize(r); return r->m_names.data(); }
extern "C" const uint64_t *ntl_fastx_name_offsets(ntl_fastx *r) { materialize(r); return r->m_name_off.data(); }

/* ------------------------------------------------------------------ writers -------------- */

/* Append-only text buffer over a std::string: raw pointer writes, two digits per division. */
/* A formatter's output buffer.  Buffers are kept between calls (RawPool): a fresh std::string per chunk and call meant an
 * allocation, its page faults and a zero fill of the whole capacity for every few hundred kilobytes of text -- more than the
 * formatting itself. */
struct RawBuf {
    char *d = nullptr;
    size_t n = 0, cap = 0;
    void reserve(size_t c)
    {
        if (c <= cap) return;
        char *nd = (char *)realloc(d, c);
        if (!nd) abort();
        d = nd; cap = c;
    }
};
struct RawPool {
    std::mutex mu;
    std::vector<RawBuf> free_list;
    void take(std::vector<RawBuf> &out, size_t k)
    {
        std::lock_guard<std::mutex> g(mu);
        out.resize(k);
        for (size_t i = 0; i < k && !free_list.empty(); i++) { out[i] = free_list.back(); free_list.pop_back(); out[i].n = 0; }
    }
    void give(std::vector<RawBuf> &bufs)
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto &b : bufs) {
            if (free_list.size() < 2048 && b.cap <= ((size_t)8 << 20)) free_list.push_back(b);
            else free(b.d);
        }
        bufs.clear();
    }
};
static RawPool &raw_pool() { static RawPool p; return p; }

struct Out {
    RawBuf &s;
    char *p, *end;
    explicit Out(RawBuf &buf, size_t cap = 1 << 16) : s(buf)
    {
        s.reserve(cap);
        p = s.d; end = p + s.cap;
    }
    inline void need(size_t n)
    {
        if ((size_t)(end - p) >= n) return;
        const size_t used = (size_t)(p - s.d);
        s.reserve(std::max(s.cap * 2, used + n + 4096));
        p = s.d + used; end = s.d + s.cap;
    }
    inline void ch(char c) { *p++ = c; } /* callers reserve with need() */
    inline void bytes(const char *b, size_t n) { need(n + 96); memcpy(p, b, n); p += n; }
    inline void u32(uint32_t v)
    {
        static const char D2[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
                                 "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
                                 "8081828384858687888990919293949596979899";
        char tmp[12];
        int n = 0;
        while (v >= 100) { const uint32_t r = v % 100; v /= 100; tmp[n++] = D2[2 * r + 1]; tmp[n++] = D2[2 * r]; }
        if (v >= 10) { tmp[n++] = D2[2 * v + 1]; tmp[n++] = D2[2 * v]; } else tmp[n++] = (char)('0' + v);
        while (n) *p++ = tmp[--n];
    }
    inline void u64(uint64_t v)
    {
        if (v <= 0xFFFFFFFFull) { u32((uint32_t)v); return; }
        /* split at 10^9: at most three 32-bit pieces */
        const uint64_t hi = v / 1000000000ull;
        const uint32_t lo = (uint32_t)(v % 1000000000ull);
        u64(hi);
        char tmp[9];
        uint32_t x = lo;
        for (int i = 8; i >= 0; i--) { tmp[i] = (char)('0' + x % 10); x /= 10; }
        memcpy(p, tmp, 9); p += 9;
    }
    void finish() { s.n = (size_t)(p - s.d); }
};

static int write_all(int fd, const RawBuf &s)
{
    size_t done = 0;
    while (done < s.n) {
        const ssize_t k = write(fd, s.d + done, s.n - done);
        if (k <= 0) return NTL_EINVAL;
        done += (size_t)k;
    }
    return NTL_OK;
}

static int pwrite_all(int fd, const RawBuf &s, off_t at)
{
    size_t done = 0;
    while (done < s.n) {
        const ssize_t k = pwrite(fd, s.d + done, s.n - done, at + (off_t)done);
        if (k <= 0) return NTL_EINVAL;
        done += (size_t)k;
    }
    return NTL_OK;
}

/* Formats records [0, n) in chunks on several threads and writes the chunks in order.  On a seekable
 * descriptor every thread pwrite()s its own chunks at their final offsets (the page-cache copy is then
 * parallel too); pipes get the chunks one after the other. */
template <typename F>
static int format_parallel(int fd, uint64_t n, uint64_t weight_hint, F fmt)
{
    PoolKind emit_pool(1);
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr == 0) nthr = 1;
    if (nthr > 32) nthr = 32;
    if (weight_hint < (1u << 16)) nthr = 1;
    const uint64_t chunk = 2048; /* records per work item */
    const uint64_t nchunks = (n + chunk - 1) / chunk;
    widen_pipe(fd);
    off_t base = nthr > 1 ? lseek(fd, 0, SEEK_CUR) : (off_t)-1;
    if (base != (off_t)-1) { /* O_APPEND would ignore pwrite offsets */
        const int fl = fcntl(fd, F_GETFL);
        if (fl < 0 || (fl & O_APPEND)) base = (off_t)-1;
    }
    uint64_t c0 = 0;
    while (c0 < nchunks) {
        const uint64_t c1 = std::min<uint64_t>(nchunks, c0 + (uint64_t)nthr * 16);
        std::vector<RawBuf> bufs;
        raw_pool().take(bufs, c1 - c0);
        run_threads(nthr, [&](size_t t) {
            for (uint64_t c = c0 + t; c < c1; c += nthr) fmt(c * chunk, std::min<uint64_t>(n, (c + 1) * chunk), bufs[c - c0]);
        });
        if (base == (off_t)-1) {
            for (auto &b : bufs) { int rc = write_all(fd, b); if (rc) { raw_pool().give(bufs); return rc; } }
        } else {
            std::vector<off_t> at(bufs.size() + 1, base);
            for (size_t i = 0; i < bufs.size(); i++) at[i + 1] = at[i] + (off_t)bufs[i].n;
            std::vector<int> rcs(nthr, 0);
            run_threads(nthr, [&](size_t t) {
                for (size_t i = t; i < bufs.size(); i += nthr)
                    if (pwrite_all(fd, bufs[i], at[i])) rcs[t] = NTL_EINVAL;
            });
            for (int rc : rcs) if (rc) { raw_pool().give(bufs); return rc; }
            base = at[bufs.size()];
            if (lseek(fd, base, SEEK_SET) == (off_t)-1) { raw_pool().give(bufs); return NTL_EINVAL; }
        }
        raw_pool().give(bufs);
        c0 = c1;
    }
    return NTL_OK;
}

/* Text that was made elsewhere (on the device: ntl_mapres_format): n bytes behind the descriptor's position, in pieces pwrite()n
 * by the worker pool where the descriptor is seekable (the page-cache copy is what costs: one thread moves 2-3 GB/s into tmpfs). */
extern "C" int ntl_write_blob(int fd, const char *p, uint64_t n)
{
    if (n && !p) return NTL_EINVAL;
    if (!n) return NTL_OK;
    PoolKind emit_pool(1);
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr == 0) nthr = 1;
    if (nthr > 16) nthr = 16;
    const uint64_t piece = (uint64_t)4 << 20;
    const uint64_t npieces = (n + piece - 1) / piece;
    if (npieces < 2) nthr = 1;
    off_t base = nthr > 1 ? lseek(fd, 0, SEEK_CUR) : (off_t)-1;
    if (base != (off_t)-1) {
        const int fl = fcntl(fd, F_GETFL);
        if (fl < 0 || (fl & O_APPEND)) base = (off_t)-1;
    }
    if (base == (off_t)-1) {
        widen_pipe(fd);
        uint64_t done = 0;
        while (done < n) {
            const ssize_t w = write(fd, p + done, (size_t)std::min<uint64_t>(n - done, (uint64_t)1 << 30));
            if (w < 0) { if (errno == EINTR) continue; return NTL_EINVAL; }
            done += (uint64_t)w;
        }
        return NTL_OK;
    }
    std::vector<int> rcs(nthr, 0);
    run_threads(nthr, [&](size_t t) {
        for (uint64_t i = t; i < npieces; i += nthr) {
            uint64_t a = i * piece;
            const uint64_t b = std::min<uint64_t>(n, a + piece);
            while (a < b) {
                const ssize_t w = pwrite(fd, p + a, (size_t)(b - a), base + (off_t)a);
                if (w < 0) { if (errno == EINTR) continue; rcs[t] = NTL_EINVAL; break; }
                a += (uint64_t)w;
            }
        }
    });
    for (int rc : rcs) if (rc) return rc;
    if (lseek(fd, base + (off_t)n, SEEK_SET) == (off_t)-1) return NTL_EINVAL;
    return NTL_OK;
}

extern "C" int ntl_write_indexlr(int fd, uint64_t nseq, const char *names, const uint64_t *name_off, const uint32_t *lengths,
                                 const uint64_t *mx_off, const uint64_t *hash, const uint32_t *pos, const uint8_t *strand)
{
    if (nseq && (!names || !name_off || !mx_off)) return NTL_EINVAL;
    return format_parallel(fd, nseq, nseq ? mx_off[nseq] : 0, [&](uint64_t a, uint64_t b, RawBuf &s) {
        Out o(s, (size_t)(mx_off[b] - mx_off[a]) * 30 + (size_t)(b - a) * 48 + 256);
        for (uint64_t i = a; i < b; i++) {
            o.bytes(names + name_off[i], name_off[i + 1] - name_off[i]);
            o.ch('\t');
            if (lengths) { o.u32(lengths[i]); o.ch('\t'); }
            for (uint64_t j = mx_off[i]; j < mx_off[i + 1]; j++) {
                o.need(64);
                if (j > mx_off[i]) o.ch(' ');
                o.u64(hash[j]); o.ch(':'); o.u32(pos[j]);
                if (strand) { o.ch(':'); o.ch(strand[j] ? '+' : '-'); } /* NULL: `--pos` without `--strand` */
            }
            o.need(8);
            o.ch('\n');
        }
        o.finish();
    });
}

extern "C" int ntl_write_verbose(int fd, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *hits,
                                 const char *read_names, const uint64_t *read_name_off,
                                 const char *ctg_names, const uint64_t *ctg_name_off)
{
    if (n_maps && (!maps || !hits || !read_names || !read_name_off || !ctg_names || !ctg_name_off)) return NTL_EINVAL;
    uint64_t w = 0;
    if (n_maps) w = maps[n_maps - 1].hit_off + maps[n_maps - 1].n_hits;
    return format_parallel(fd, n_maps, w, [&](uint64_t a, uint64_t b, RawBuf &s) {
        const uint64_t nh = b > a ? maps[b - 1].hit_off + maps[b - 1].n_hits - maps[a].hit_off : 0;
        Out o(s, (size_t)nh * 22 + (size_t)(b - a) * 80 + 256);
        for (uint64_t i = a; i < b; i++) {
            const ntl_mapping &m = maps[i];
            o.bytes(read_names + read_name_off[m.read], read_name_off[m.read + 1] - read_name_off[m.read]);
            o.ch('\t');
            o.bytes(ctg_names + ctg_name_off[m.ctg], ctg_name_off[m.ctg + 1] - ctg_name_off[m.ctg]);
            o.ch('\t');
            o.u32(m.n_hits);
            o.ch('\t');
            for (uint32_t j = 0; j < m.n_hits; j++) {
                const ntl_hit &h = hits[m.hit_off + j];
                o.need(40);
                if (j) o.ch(' ');
                o.u32(h.ctg_pos); o.ch(':'); o.ch(h.ctg_strand ? '+' : '-'); o.ch('_');
                o.u32(h.read_pos); o.ch(':'); o.ch(h.read_strand ? '+' : '-');
            }
            o.need(8);
            o.ch('\n');
        }
        o.finish();
    });
}

extern "C" int ntl_write_paf(int fd, const ntl_paf *pafs, uint64_t n, const char *read_names, const uint64_t *read_name_off,
                             const uint32_t *read_len, const char *ctg_names, const uint64_t *ctg_name_off, const uint32_t *ctg_len)
{
    if (n && (!pafs || !read_names || !read_name_off || !read_len || !ctg_names || !ctg_name_off || !ctg_len)) return NTL_EINVAL;
    return format_parallel(fd, n, n * 12, [&](uint64_t a, uint64_t b, RawBuf &s) {
        Out o(s, (size_t)(b - a) * 160 + 256);
        for (uint64_t i = a; i < b; i++) {
            const ntl_paf &p = pafs[i];
            o.bytes(read_names + read_name_off[p.read], read_name_off[p.read + 1] - read_name_off[p.read]);
            o.ch('\t'); o.u32(read_len[p.read]);
            o.ch('\t'); o.u32(p.q_start);
            o.ch('\t'); o.u32(p.q_end);
            o.ch('\t'); o.ch(p.strand ? '+' : '-');
            o.ch('\t');
            o.bytes(ctg_names + ctg_name_off[p.ctg], ctg_name_off[p.ctg + 1] - ctg_name_off[p.ctg]);
            o.ch('\t'); o.u32(ctg_len[p.ctg]);
            o.ch('\t'); o.u32(p.t_start);
            o.ch('\t'); o.u32(p.t_end);
            o.ch('\t'); o.u32(p.n_hits);
            o.ch('\t'); o.u64((uint64_t)p.t_end - p.t_start);
            o.bytes("\t255\n", 5);
        }
        o.finish();
    });
}

/* ------------------------------------------------------------------ indexlr TSV parser ---- */

/*
 * The text side of operator B2: `id\t[len\t]H:pos:strand H:pos:strand ...` lines as the reference
 * splits them (bin/ntlink_pair.py:197-207 for contigs, :355-378 for reads: strip, split on tabs,
 * tokens separated by single spaces, `mx:pos:strand`).  Blocks of the input are cut at line ends
 * and parsed by several threads, first counting, then writing into the caller's arrays -- the same
 * two-pass scheme as the FASTA reader.  Every non-empty line keeps its place, with or without
 * minimizers.
 */
struct TsvRange {
    const char *b = nullptr, *e = nullptr;
    uint64_t nrec = 0, nmx = 0, name_bytes = 0;
    int bad = 0;
};

struct ntl_tsv {
    int fd = -1;
    bool with_len = false, pos_only = false, eof = false;
    char *buf = nullptr;   /* block being parsed + the partial line behind it (from the buffer cache) */
    size_t cap = 0, size_hint = 0;
    size_t have = 0, used = 0; /* buf[0..have) read; buf[0..used) = whole lines of the current batch */
    std::vector<TsvRange> ranges;
    std::string err;
};

static inline bool tsv_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

struct TsvCount {
    uint64_t nrec = 0, nmx = 0, name_bytes = 0;
    void rec(const char *, size_t n, uint32_t) { nrec++; name_bytes += n; }
    void mx(uint64_t, uint32_t, uint8_t) { nmx++; }
    void end() {}
};
struct TsvWrite {
    char *names; uint64_t *name_off; uint32_t *lengths; uint64_t *mx_off; uint64_t *hash; uint32_t *pos; uint8_t *strand;
    uint64_t r, m, nb; /* running record / minimizer / name-byte position (global) */
    void rec(const char *p, size_t n, uint32_t len)
    {
        memcpy(names + nb, p, n); nb += n;
        name_off[r + 1] = nb;
        if (lengths) lengths[r] = len;
    }
    void mx(uint64_t h, uint32_t p, uint8_t s) { hash[m] = h; pos[m] = p; strand[m] = s; m++; }
    void end() { mx_off[r + 1] = m; r++; }
};

/* parses the lines of [p, e) (e at a line end); returns 0 or the 1-based line of the first malformed token */
template <typename Sink>
static int tsv_parse(const char *p, const char *e, bool with_len, bool pos_only, Sink &out)
{
    int line_no = 0;
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        const char *le = nl ? nl : e;
        const char *a = p, *b = le;
        p = nl ? nl + 1 : e;
        line_no++;
        while (a < b && tsv_space(*a)) a++;
        while (b > a && tsv_space(b[-1])) b--;
        if (a == b) continue;
        const char *t1 = (const char *)memchr(a, '\t', (size_t)(b - a));
        const char *name_e = t1 ? t1 : b;
        const char *f = t1 ? t1 + 1 : b; /* field after the id */
        uint32_t len = 0;
        if (with_len) {
            const char *t2 = f < b ? (const char *)memchr(f, '\t', (size_t)(b - f)) : nullptr;
            const char *le2 = t2 ? t2 : b;
            uint64_t v = 0;
            for (const char *q = f; q < le2; q++) {
                if (*q < '0' || *q > '9') return line_no;
                v = v * 10 + (uint64_t)(*q - '0');
            }
            len = (uint32_t)v;
            f = t2 ? t2 + 1 : b;
        }
        out.rec(a, (size_t)(name_e - a), len);
        const char *fe = f < b ? (const char *)memchr(f, '\t', (size_t)(b - f)) : nullptr; /* later columns are ignored */
        if (!fe) fe = b;
        const char *q = f;
        while (q < fe) {
            if (*q == ' ') { q++; continue; }
            uint64_t h = 0;
            const char *s0 = q;
            while (q < fe && *q >= '0' && *q <= '9') { h = h * 10 + (uint64_t)(*q - '0'); q++; }
            if (q == s0 || q >= fe || *q != ':') return line_no;
            q++;
            uint64_t ps = 0;
            s0 = q;
            while (q < fe && *q >= '0' && *q <= '9') { ps = ps * 10 + (uint64_t)(*q - '0'); q++; }
            if (q == s0) return line_no;
            if (pos_only) { /* `--pos` without `--strand` (ntLink:244,249): the token ends here */
                out.mx(h, (uint32_t)ps, 1);
                if (q < fe && *q != ' ') return line_no;
                continue;
            }
            if (q >= fe || *q != ':') return line_no;
            q++;
            if (q >= fe || (*q != '+' && *q != '-')) return line_no;
            out.mx(h, (uint32_t)ps, *q == '+' ? 1 : 0);
            q++;
            if (q < fe && *q != ' ') return line_no;
        }
        out.end();
    }
    return 0;
}

extern "C" int ntl_tsv_open(const char *path, int with_len, ntl_tsv **out)
{
    if (!path || !out) return NTL_EINVAL;
    *out = nullptr;
    int fd = strcmp(path, "-") == 0 ? dup(0) : open(path, O_RDONLY);
    if (fd < 0) return NTL_EINVAL;
    ntl_tsv *r = new ntl_tsv();
    r->fd = fd;
    widen_pipe(fd);
    r->with_len = (with_len & 1) != 0;
    r->pos_only = (with_len & 2) != 0;
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) r->size_hint = (size_t)st.st_size + 1;
    *out = r;
    return NTL_OK;
}

extern "C" void ntl_tsv_close(ntl_tsv *r)
{
    if (!r) return;
    if (r->fd >= 0) close(r->fd);
    if (r->buf) buf_cache().give(r->buf, r->cap);
    delete r;
}

extern "C" const char *ntl_tsv_error(const ntl_tsv *r) { return r ? r->err.c_str() : "no reader"; }

extern "C" void ntl_tsv_sizes(const ntl_tsv *r, uint64_t *nrec, uint64_t *nmx, uint64_t *name_bytes)
{
    uint64_t a = 0, b = 0, c = 0;
    if (r) for (auto &g : r->ranges) { a += g.nrec; b += g.nmx; c += g.name_bytes; }
    if (nrec) *nrec = a;
    if (nmx) *nmx = b;
    if (name_bytes) *name_bytes = c;
}

/* Reads about max_bytes of text (0 = everything), whole lines only, and counts what it holds;
 * *nrec == 0 at the end of the input. */
extern "C" int ntl_tsv_next(ntl_tsv *r, uint64_t max_bytes, uint64_t *nrec)
{
    if (!r || !nrec) return NTL_EINVAL;
    *nrec = 0;
    /* drop the previous batch, keep the partial line behind it */
    if (r->used) { memmove(r->buf, r->buf + r->used, r->have - r->used); r->have -= r->used; r->used = 0; }
    r->ranges.clear();
    for (;;) {
        /* fill */
        const size_t target = max_bytes ? (size_t)max_bytes : (size_t)-1;
        while (!r->eof && r->have < target) {
            if (r->cap == r->have) {
                size_t want_cap = r->cap ? r->cap * 2 : std::max<size_t>((size_t)8 << 20, max_bytes ? (size_t)max_bytes + (1u << 20) : r->size_hint);
                size_t got = 0;
                char *nb = buf_cache().take(want_cap, &got);
                if (!nb) { r->err = "out of memory"; return NTL_ENOMEM; }
                if (r->have) memcpy(nb, r->buf, r->have);
                if (r->buf) buf_cache().give(r->buf, r->cap);
                r->buf = nb; r->cap = got;
            }
            const size_t want = std::min(r->cap - r->have, target - r->have);
            const ssize_t n = read(r->fd, r->buf + r->have, want);
            if (n < 0) { r->err = "read error"; return NTL_EINVAL; }
            if (n == 0) { r->eof = true; break; }
            r->have += (size_t)n;
        }
        if (r->have == 0) return NTL_OK;
        /* whole lines only, unless this is the end */
        size_t end = r->have;
        if (!r->eof) {
            while (end > 0 && r->buf[end - 1] != '\n') end--;
            if (end == 0) { /* one line longer than the block: read on */
                if (max_bytes) max_bytes *= 2;
                continue;
            }
        }
        r->used = end;
        break;
    }
    const char *p0 = r->buf, *pe = p0 + r->used;
    const size_t span = r->used;
    size_t min_chunk = 1u << 20;
    if (const char *e = getenv("NTL_IO_MIN_CHUNK")) { long v = atol(e); if (v > 0) min_chunk = (size_t)v; }
    const unsigned T = (unsigned)std::min<size_t>(io_threads(), std::max<size_t>(1, span / min_chunk));
    r->ranges.resize(T);
    const char *prev = p0;
    for (unsigned t = 0; t < T; t++) {
        const char *nxt = pe;
        if (t + 1 < T) {
            const char *g = p0 + span / T * (t + 1);
            if (g < prev) g = prev;
            const char *nl = (const char *)memchr(g, '\n', (size_t)(pe - g));
            nxt = nl ? nl + 1 : pe;
        }
        r->ranges[t].b = prev; r->ranges[t].e = nxt;
        prev = nxt;
    }
    run_threads(T, [&](size_t t) {
        TsvRange &g = r->ranges[t];
        TsvCount c;
        g.bad = tsv_parse(g.b, g.e, r->with_len, r->pos_only, c);
        g.nrec = c.nrec; g.nmx = c.nmx; g.name_bytes = c.name_bytes;
    });
    for (auto &g : r->ranges)
        if (g.bad) { r->err = r->pos_only ? "malformed minimizer token (expected hash:pos)" : "malformed minimizer token (expected hash:pos:strand)"; return NTL_EINVAL; }
    ntl_tsv_sizes(r, nrec, nullptr, nullptr);
    if (*nrec == 0 && !r->eof) return ntl_tsv_next(r, max_bytes, nrec); /* a block of blank lines */
    return NTL_OK;
}

/* name_off and mx_off have nrec + 1 entries; lengths may be NULL */
extern "C" int ntl_tsv_copy(const ntl_tsv *r, char *names, uint64_t *name_off, uint32_t *lengths, uint64_t *mx_off,
                            uint64_t *hash, uint32_t *pos, uint8_t *strand)
{
    if (!r || !name_off || !mx_off) return NTL_EINVAL;
    const size_t T = r->ranges.size();
    std::vector<uint64_t> r0(T + 1, 0), m0(T + 1, 0), n0(T + 1, 0);
    for (size_t t = 0; t < T; t++) {
        r0[t + 1] = r0[t] + r->ranges[t].nrec;
        m0[t + 1] = m0[t] + r->ranges[t].nmx;
        n0[t + 1] = n0[t] + r->ranges[t].name_bytes;
    }
    if ((m0[T] && (!hash || !pos || !strand)) || (n0[T] && !names)) return NTL_EINVAL;
    name_off[0] = 0; mx_off[0] = 0;
    run_threads(T, [&](size_t t) {
        const TsvRange &g = r->ranges[t];
        TsvWrite w{names, name_off, lengths, mx_off, hash, pos, strand, r0[t], m0[t], n0[t]};
        tsv_parse(g.b, g.e, r->with_len, r->pos_only, w);
    });
    return NTL_OK;
}
