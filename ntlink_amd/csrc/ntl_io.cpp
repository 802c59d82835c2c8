/*
 * Host-side native I/O of the pair stage (no GPU code): FASTA/FASTQ(.gz) ingest and the text
 * emitters, so that the Python host never loops over bases, minimizers or hits.
 *
 *   ntl_fastx_*        gzip -cd + SeqReader of the reference's pipe (ntLink:113-117,222-223); record
 *                      semantics of bin/read_fasta.py:6-46 (id = header up to the first whitespace,
 *                      multi-line sequences joined, FASTQ qualities skipped by length).
 *   ntl_write_indexlr  the TSV `indexlr --long --pos --strand [--len]` prints (ntLink:199,223).
 *   ntl_write_verbose  <prefix>.verbose_mapping.tsv lines (bin/ntlink_pair.py:308-313,382-388).
 *   ntl_write_paf      <prefix>.paf lines (bin/ntlink_paf_output.py:131-135).
 *
 * Formatting is split over threads by record ranges; every thread fills its own buffer and the
 * buffers are written in order.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ntlink_amd.h"

/* ------------------------------------------------------------------ reader -------------- */

struct ntl_fastx {
    gzFile gz = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    std::string pending;     /* header line read ahead (without the newline) */
    bool has_pending = false;
    /* current batch */
    std::string seqs, names;
    std::vector<uint64_t> off, name_off;
    std::string err;
};

static bool fill(ntl_fastx *r)
{
    if (r->eof) return false;
    if (r->pos < r->end) {
        memmove(r->buf.data(), r->buf.data() + r->pos, r->end - r->pos);
    }
    r->end -= r->pos;
    r->pos = 0;
    int n = gzread(r->gz, r->buf.data() + r->end, (unsigned)(r->buf.size() - r->end));
    if (n < 0) { int e; r->err = gzerror(r->gz, &e); r->eof = true; return false; }
    if (n == 0) { r->eof = true; return false; }
    r->end += (size_t)n;
    return true;
}

/* next line without its newline (and without a trailing '\r'); false at end of input */
static bool next_line(ntl_fastx *r, const char **p, size_t *len, std::string &spill)
{
    for (;;) {
        char *b = r->buf.data() + r->pos;
        char *nl = (char *)memchr(b, '\n', r->end - r->pos);
        if (nl) {
            size_t l = (size_t)(nl - b);
            r->pos += l + 1;
            if (l && b[l - 1] == '\r') l--;
            *p = b; *len = l;
            return true;
        }
        if (r->end - r->pos == r->buf.size()) r->buf.resize(r->buf.size() * 2); /* very long line */
        const size_t before = r->end - r->pos;
        if (!fill(r)) {
            if (before == 0) return false;
            /* last line without newline */
            spill.assign(r->buf.data() + r->pos, before);
            r->pos = r->end;
            if (!spill.empty() && spill.back() == '\r') spill.pop_back();
            *p = spill.data(); *len = spill.size();
            return true;
        }
    }
}

extern "C" int ntl_fastx_open(const char *path, ntl_fastx **out)
{
    if (!path || !out) return NTL_EINVAL;
    *out = nullptr;
    ntl_fastx *r = new ntl_fastx();
    r->gz = strcmp(path, "-") == 0 ? gzdopen(dup(0), "rb") : gzopen(path, "rb");
    if (!r->gz) { delete r; return NTL_EINVAL; }
    gzbuffer(r->gz, 1 << 20);
    r->buf.resize(8 << 20);
    *out = r;
    return NTL_OK;
}

extern "C" void ntl_fastx_close(ntl_fastx *r)
{
    if (!r) return;
    if (r->gz) gzclose(r->gz);
    delete r;
}

extern "C" const char *ntl_fastx_error(const ntl_fastx *r) { return r ? r->err.c_str() : "no reader"; }

/* Collects records until at least max_bases bases are held (0 = to the end of the input). */
extern "C" int ntl_fastx_next(ntl_fastx *r, uint64_t max_bases, uint64_t *nseq)
{
    if (!r || !nseq) return NTL_EINVAL;
    r->seqs.clear(); r->names.clear(); r->off.assign(1, 0); r->name_off.assign(1, 0);
    std::string spill;
    const char *p; size_t len;
    for (;;) {
        /* find the next header */
        if (!r->has_pending) {
            bool found = false;
            while (next_line(r, &p, &len, spill)) {
                if (len && (p[0] == '>' || p[0] == '@')) { r->pending.assign(p, len); found = true; break; }
            }
            if (!found) break;
        }
        r->has_pending = false;
        /* id = header up to the first whitespace */
        {
            const std::string &h = r->pending;
            size_t a = 1;
            while (a < h.size() && (h[a] == ' ' || h[a] == '\t')) a++; /* str.split(None, 1) skips leading blanks */
            size_t b = a;
            while (b < h.size() && h[b] != ' ' && h[b] != '\t' && h[b] != '\r' && h[b] != '\f' && h[b] != '\v') b++;
            r->names.append(h, a, b - a);
            r->name_off.push_back(r->names.size());
        }
        /* sequence lines */
        bool plus = false, more = false;
        while ((more = next_line(r, &p, &len, spill))) {
            if (len && (p[0] == '>' || p[0] == '@' || p[0] == '+')) {
                if (p[0] == '+') plus = true;
                else { r->pending.assign(p, len); r->has_pending = true; }
                break;
            }
            r->seqs.append(p, len);
        }
        const uint64_t slen = r->seqs.size() - r->off.back();
        r->off.push_back(r->seqs.size());
        if (plus) { /* FASTQ: skip quality lines until their length reaches the sequence length */
            uint64_t got = 0;
            while (next_line(r, &p, &len, spill)) {
                got += len;
                if (got >= slen) break;
            }
        }
        (void)more;
        if (max_bases && r->seqs.size() >= max_bases) break;
    }
    if (!r->err.empty()) return NTL_EINVAL;
    *nseq = r->off.size() - 1;
    return NTL_OK;
}

extern "C" const char *ntl_fastx_seqs(const ntl_fastx *r) { return r->seqs.data(); }
extern "C" const uint64_t *ntl_fastx_offsets(const ntl_fastx *r) { return r->off.data(); }
extern "C" const char *ntl_fastx_names(const ntl_fastx *r) { return r->names.data(); }
extern "C" const uint64_t *ntl_fastx_name_offsets(const ntl_fastx *r) { return r->name_off.data(); }

/* ------------------------------------------------------------------ writers -------------- */

static inline void put_u64(std::string &s, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) s.push_back(tmp[--n]);
}

static int write_all(int fd, const std::string &s)
{
    size_t done = 0;
    while (done < s.size()) {
        ssize_t n = write(fd, s.data() + done, s.size() - done);
        if (n <= 0) return NTL_EINVAL;
        done += (size_t)n;
    }
    return NTL_OK;
}

template <typename F>
static int format_parallel(int fd, uint64_t n, uint64_t weight_hint, F fmt)
{
    unsigned nthr = std::thread::hardware_concurrency();
    if (nthr == 0) nthr = 1;
    if (nthr > 16) nthr = 16;
    if (weight_hint < (1u << 16)) nthr = 1;
    const uint64_t chunk = 4096; /* records per work item: keeps buffers small and order simple */
    const uint64_t nchunks = (n + chunk - 1) / chunk;
    uint64_t c0 = 0;
    while (c0 < nchunks) {
        const uint64_t c1 = std::min<uint64_t>(nchunks, c0 + (uint64_t)nthr * 8);
        std::vector<std::string> bufs(c1 - c0);
        std::vector<std::thread> th;
        auto work = [&](unsigned t) {
            for (uint64_t c = c0 + t; c < c1; c += nthr) fmt(c * chunk, std::min<uint64_t>(n, (c + 1) * chunk), bufs[c - c0]);
        };
        if (nthr == 1) work(0);
        else { for (unsigned t = 0; t < nthr; t++) th.emplace_back(work, t); for (auto &x : th) x.join(); }
        for (auto &b : bufs) { int rc = write_all(fd, b); if (rc) return rc; }
        c0 = c1;
    }
    return NTL_OK;
}

extern "C" int ntl_write_indexlr(int fd, uint64_t nseq, const char *names, const uint64_t *name_off, const uint32_t *lengths,
                                 const uint64_t *mx_off, const uint64_t *hash, const uint32_t *pos, const uint8_t *strand)
{
    if (nseq && (!names || !name_off || !mx_off)) return NTL_EINVAL;
    return format_parallel(fd, nseq, nseq ? mx_off[nseq] : 0, [&](uint64_t a, uint64_t b, std::string &s) {
        for (uint64_t i = a; i < b; i++) {
            s.append(names + name_off[i], name_off[i + 1] - name_off[i]);
            s.push_back('\t');
            if (lengths) { put_u64(s, lengths[i]); s.push_back('\t'); }
            for (uint64_t j = mx_off[i]; j < mx_off[i + 1]; j++) {
                if (j > mx_off[i]) s.push_back(' ');
                put_u64(s, hash[j]); s.push_back(':'); put_u64(s, pos[j]);
                if (strand) { s.push_back(':'); s.push_back(strand[j] ? '+' : '-'); } /* NULL: `--pos` without `--strand` */
            }
            s.push_back('\n');
        }
    });
}

extern "C" int ntl_write_verbose(int fd, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *hits,
                                 const char *read_names, const uint64_t *read_name_off,
                                 const char *ctg_names, const uint64_t *ctg_name_off)
{
    if (n_maps && (!maps || !hits || !read_names || !read_name_off || !ctg_names || !ctg_name_off)) return NTL_EINVAL;
    uint64_t w = 0;
    if (n_maps) w = maps[n_maps - 1].hit_off + maps[n_maps - 1].n_hits;
    return format_parallel(fd, n_maps, w, [&](uint64_t a, uint64_t b, std::string &s) {
        for (uint64_t i = a; i < b; i++) {
            const ntl_mapping &m = maps[i];
            s.append(read_names + read_name_off[m.read], read_name_off[m.read + 1] - read_name_off[m.read]);
            s.push_back('\t');
            s.append(ctg_names + ctg_name_off[m.ctg], ctg_name_off[m.ctg + 1] - ctg_name_off[m.ctg]);
            s.push_back('\t');
            put_u64(s, m.n_hits);
            s.push_back('\t');
            for (uint32_t j = 0; j < m.n_hits; j++) {
                const ntl_hit &h = hits[m.hit_off + j];
                if (j) s.push_back(' ');
                put_u64(s, h.ctg_pos); s.push_back(':'); s.push_back(h.ctg_strand ? '+' : '-'); s.push_back('_');
                put_u64(s, h.read_pos); s.push_back(':'); s.push_back(h.read_strand ? '+' : '-');
            }
            s.push_back('\n');
        }
    });
}

extern "C" int ntl_write_paf(int fd, const ntl_paf *pafs, uint64_t n, const char *read_names, const uint64_t *read_name_off,
                             const uint32_t *read_len, const char *ctg_names, const uint64_t *ctg_name_off, const uint32_t *ctg_len)
{
    if (n && (!pafs || !read_names || !read_name_off || !read_len || !ctg_names || !ctg_name_off || !ctg_len)) return NTL_EINVAL;
    return format_parallel(fd, n, n * 12, [&](uint64_t a, uint64_t b, std::string &s) {
        for (uint64_t i = a; i < b; i++) {
            const ntl_paf &p = pafs[i];
            s.append(read_names + read_name_off[p.read], read_name_off[p.read + 1] - read_name_off[p.read]);
            s.push_back('\t'); put_u64(s, read_len[p.read]);
            s.push_back('\t'); put_u64(s, p.q_start);
            s.push_back('\t'); put_u64(s, p.q_end);
            s.push_back('\t'); s.push_back(p.strand ? '+' : '-');
            s.push_back('\t');
            s.append(ctg_names + ctg_name_off[p.ctg], ctg_name_off[p.ctg + 1] - ctg_name_off[p.ctg]);
            s.push_back('\t'); put_u64(s, ctg_len[p.ctg]);
            s.push_back('\t'); put_u64(s, p.t_start);
            s.push_back('\t'); put_u64(s, p.t_end);
            s.push_back('\t'); put_u64(s, p.n_hits);
            s.push_back('\t'); put_u64(s, (uint64_t)p.t_end - p.t_start);
            s.append("\t255\n");
        }
    });
}
