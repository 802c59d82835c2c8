/*
 * Liftover of <prefix>.verbose_mapping.tsv through an AGP: what
 * bin/ntlink_liftover_mappings.py does between two ntLink rounds (ntLink_rounds:124-125), file to file.
 *
 *   liftover_ctg_mappings  (bin/ntlink_liftover_mappings.py:61-87)   one verbose line -> path id + moved positions
 *   print_adjusted_mappings (:89-121)                                per read: runs by path id, subsumption on first
 *                                                                    occurrences, concatenation, strict monotonicity
 *   liftover_mappings       (:125-143)                               reads = runs of lines with the same read id
 *
 * The file is cut where the read id changes and the pieces are lifted by several threads; the output keeps the order
 * of the input.  Host only.
 */
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/ntlink_amd.h"

namespace {

struct AgpEntry {
    std::string_view path_id;
    int64_t scaf_start, ctg_start, ctg_end;
    char ori;
    bool renamed; /* path_id != contig id */
};

struct Pos {
    int64_t ctg_pos;
    std::string_view ctg_strand, read_pos, read_strand; /* read_pos is re-printed through int() by the reference */
    int64_t read_pos_v;
    char flipped; /* 0, or the strand byte after reverse_orientation */
};

struct LineMap {
    std::string_view path;
    size_t first, count; /* adjusted mappings in the read's pool */
    int pid;
};

inline bool sp(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

inline bool parse_int(std::string_view s, int64_t &v)
{
    /* int(): optional sign, digits (surrounding blanks and underscores are not produced by any writer of this format) */
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
    if (i == s.size()) return false;
    int64_t x = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        x = x * 10 + (s[i] - '0');
    }
    v = neg ? -x : x;
    return true;
}

inline void put_int(std::string &o, int64_t v)
{
    char b[24];
    int n = 0;
    uint64_t u = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) o.push_back('-');
    while (n) o.push_back(b[--n]);
}

struct Lifter {
    const std::unordered_map<std::string_view, AgpEntry> &agp;
    int k;
    std::vector<Pos> pool;
    std::vector<LineMap> maps;
    std::unordered_map<std::string_view, int> ids;
    std::vector<int> first_run, run_pid;
    std::vector<char> subsumed;
    uint64_t lines_out = 0;

    /* one line; false = malformed */
    bool add_line(std::string_view ctg, std::string_view mappings)
    {
        LineMap lm;
        lm.first = pool.size();
        lm.count = 0;
        lm.pid = -1;
        auto it = agp.find(ctg);
        if (it == agp.end()) { /* not in the AGP: keeps its name, loses its mappings (:66-67) */
            lm.path = ctg;
            maps.push_back(lm);
            /* the reference returns before parsing the tokens */
            return true;
        }
        const AgpEntry &e = it->second;
        lm.path = e.path_id;
        const int64_t lo = e.ctg_start - 1, hi = e.ctg_end - k, len = e.ctg_end - e.ctg_start + 1, offset = e.scaf_start - 1;
        size_t p = 0;
        while (p <= mappings.size()) {
            size_t q = mappings.find(' ', p);
            if (q == std::string_view::npos) q = mappings.size();
            std::string_view tok = mappings.substr(p, q - p);
            p = q + 1;
            /* ctgpos:strand_readpos:strand */
            const size_t us = tok.find('_');
            if (us == std::string_view::npos || tok.find('_', us + 1) != std::string_view::npos) return false;
            std::string_view a = tok.substr(0, us), b = tok.substr(us + 1);
            const size_t c1 = a.find(':'), c2 = b.find(':');
            if (c1 == std::string_view::npos || c2 == std::string_view::npos || a.find(':', c1 + 1) != std::string_view::npos ||
                b.find(':', c2 + 1) != std::string_view::npos)
                return false;
            Pos m;
            if (!parse_int(a.substr(0, c1), m.ctg_pos) || !parse_int(b.substr(0, c2), m.read_pos_v)) return false;
            m.ctg_strand = a.substr(c1 + 1);
            m.read_strand = b.substr(c2 + 1);
            m.flipped = 0;
            if (m.ctg_pos < lo || m.ctg_pos > hi) continue; /* outside the component's range (:72-73) */
            const int64_t adj = m.ctg_pos - lo;
            if (e.ori == '+' && e.renamed) m.ctg_pos = offset + adj;
            else if (e.ori == '-' && e.renamed) {
                m.ctg_pos = offset + (len - adj) - k;
                if (m.ctg_strand == "+") m.flipped = '-';
                else if (m.ctg_strand == "-") m.flipped = '+';
                else return false; /* reverse_orientation asserts */
            }
            pool.push_back(m);
            lm.count++;
        }
        maps.push_back(lm);
        return true;
    }

    void flush_read(std::string_view read_id, std::string &out)
    {
        const size_t n = maps.size();
        if (!n) return;
        ids.clear();
        for (auto &m : maps) {
            auto r = ids.emplace(m.path, (int)ids.size());
            m.pid = r.first->second;
        }
        const size_t np = ids.size();
        first_run.assign(np, -1);
        subsumed.assign(np, 0);
        run_pid.clear();
        for (size_t i = 0; i < n; i++)
            if (i == 0 || maps[i].pid != maps[i - 1].pid) run_pid.push_back(maps[i].pid);
        for (size_t r = 0; r < run_pid.size(); r++) {
            const int p = run_pid[r];
            if (first_run[p] < 0) first_run[p] = (int)r;
            else
                for (size_t j = (size_t)first_run[p] + 1; j < r; j++) subsumed[run_pid[j]] = 1;
        }
        /* runs of the surviving lines, by path */
        size_t i = 0;
        while (i < n) {
            if (subsumed[maps[i].pid]) { i++; continue; }
            const int p = maps[i].pid;
            size_t j = i, total = 0;
            /* the next surviving lines with the same path (subsumed ones in between have been filtered away) */
            std::vector<size_t> &members = tmp_members;
            members.clear();
            while (j < n) {
                if (subsumed[maps[j].pid]) { j++; continue; }
                if (maps[j].pid != p) break;
                members.push_back(j);
                total += maps[j].count;
                j++;
            }
            i = j;
            if (!total) continue;
            bool inc = true, dec = true, have = false;
            int64_t prev = 0;
            for (size_t mi : members)
                for (size_t t = 0; t < maps[mi].count; t++) {
                    const int64_t v = pool[maps[mi].first + t].ctg_pos;
                    if (have) { inc = inc && prev < v; dec = dec && prev > v; }
                    prev = v; have = true;
                }
            if (!inc && !dec) continue;
            out.append(read_id);
            out.push_back('\t');
            out.append(maps[members[0]].path);
            out.push_back('\t');
            put_int(out, (int64_t)total);
            out.push_back('\t');
            bool firstTok = true;
            for (size_t mi : members)
                for (size_t t = 0; t < maps[mi].count; t++) {
                    const Pos &m = pool[maps[mi].first + t];
                    if (!firstTok) out.push_back(' ');
                    firstTok = false;
                    put_int(out, m.ctg_pos);
                    out.push_back(':');
                    if (m.flipped) out.push_back(m.flipped); else out.append(m.ctg_strand);
                    out.push_back('_');
                    put_int(out, m.read_pos_v);
                    out.push_back(':');
                    out.append(m.read_strand);
                }
            out.push_back('\n');
            lines_out++;
        }
        pool.clear();
        maps.clear();
    }
    std::vector<size_t> tmp_members;
};

/* [a, b) of a line without the surrounding blanks (str.strip()) */
inline std::string_view stripped(const char *a, const char *b)
{
    while (a < b && sp(*a)) a++;
    while (b > a && sp(b[-1])) b--;
    return std::string_view(a, (size_t)(b - a));
}

inline std::string_view read_id_of(const char *p, const char *e)
{
    const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
    std::string_view s = stripped(p, nl ? nl : e);
    const size_t t = s.find('\t');
    return t == std::string_view::npos ? s : s.substr(0, t);
}

int lift_range(const char *p, const char *e, Lifter &L, std::string &out, uint64_t &lines_in)
{
    std::string_view cur;
    bool have = false;
    while (p < e) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
        std::string_view s = stripped(p, nl ? nl : e);
        p = nl ? nl + 1 : e;
        lines_in++;
        /* read_id, ctg, num_anchors, mappings = line (:63): exactly four fields */
        size_t t1 = s.find('\t'), t2 = t1 == std::string_view::npos ? t1 : s.find('\t', t1 + 1),
               t3 = t2 == std::string_view::npos ? t2 : s.find('\t', t2 + 1);
        if (t3 == std::string_view::npos || s.find('\t', t3 + 1) != std::string_view::npos) return NTL_EINVAL;
        std::string_view rid = s.substr(0, t1), ctg = s.substr(t1 + 1, t2 - t1 - 1), mp = s.substr(t3 + 1);
        if (have && rid != cur) L.flush_read(cur, out);
        cur = rid;
        have = true;
        if (!L.add_line(ctg, mp)) return NTL_EINVAL;
    }
    if (have) L.flush_read(cur, out);
    return NTL_OK;
}

} // namespace

extern "C" int ntl_liftover(const char *mappings_path, const char *out_path, int k, uint64_t n_agp, const char *ctg_ids,
                            const uint64_t *ctg_id_off, const char *path_ids, const uint64_t *path_id_off,
                            const int64_t *scaf_start, const int64_t *ctg_start, const int64_t *ctg_end,
                            const char *orientation, uint64_t *lines_in, uint64_t *lines_out)
{
    if (!mappings_path || !out_path || (n_agp && (!ctg_ids || !ctg_id_off || !path_ids || !path_id_off || !scaf_start ||
                                                  !ctg_start || !ctg_end || !orientation)))
        return NTL_EINVAL;
    std::unordered_map<std::string_view, AgpEntry> agp;
    agp.reserve((size_t)n_agp * 2);
    for (uint64_t i = 0; i < n_agp; i++) {
        std::string_view c(ctg_ids + ctg_id_off[i], (size_t)(ctg_id_off[i + 1] - ctg_id_off[i]));
        std::string_view p(path_ids + path_id_off[i], (size_t)(path_id_off[i + 1] - path_id_off[i]));
        agp[c] = AgpEntry{p, scaf_start[i], ctg_start[i], ctg_end[i], orientation[i], p != c}; /* later lines win (dict) */
    }
    int fd = open(mappings_path, O_RDONLY);
    if (fd < 0) return NTL_EINVAL;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return NTL_EINVAL; }
    const size_t size = (size_t)st.st_size;
    int ofd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (ofd < 0) { close(fd); return NTL_EINVAL; }
    uint64_t nin = 0, nout = 0;
    int rc = NTL_OK;
    if (size) {
        const char *base = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (base == MAP_FAILED) { close(fd); close(ofd); return NTL_ENOMEM; }
        const char *end = base + size;
        unsigned T = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        if (const char *ev = getenv("NTL_IO_THREADS")) { int v = atoi(ev); if (v > 0) T = (unsigned)v; }
        size_t min_chunk = 1u << 20;
        if (const char *ev = getenv("NTL_IO_MIN_CHUNK")) { long v = atol(ev); if (v > 0) min_chunk = (size_t)v; }
        T = (unsigned)std::min<size_t>(T, std::max<size_t>(1, size / min_chunk));
        /* The file is lifted block by block (about 256 MB of input, so that the text held in memory stays bounded); a block is
           cut into T pieces; every cut is a line start where the read id differs from the line before. */
        auto next_cut = [&](const char *g, const char *lo) -> const char * {
            if (g <= lo) g = lo;
            if (g >= end) return end;
            const char *nl = (const char *)memchr(g, '\n', (size_t)(end - g));
            const char *ls = nl ? nl + 1 : end;
            while (ls < end) {
                const char *pb = ls - 1; /* the '\n' ending the previous line */
                while (pb > base && pb[-1] != '\n') pb--;
                if (read_id_of(pb, end) != read_id_of(ls, end)) break;
                const char *nx = (const char *)memchr(ls, '\n', (size_t)(end - ls));
                ls = nx ? nx + 1 : end;
            }
            return ls;
        };
        size_t block = (size_t)256 << 20;
        if (const char *ev = getenv("NTL_LIFTOVER_BLOCK")) { long v = atol(ev); if (v > 0) block = (size_t)v; }
        const char *at = base;
        while (at < end && rc == NTL_OK) {
            const char *bend = (size_t)(end - at) <= block ? end : next_cut(at + block, at);
            const size_t bsize = (size_t)(bend - at);
            const unsigned Tb = (unsigned)std::min<size_t>(T, std::max<size_t>(1, bsize / min_chunk));
            std::vector<const char *> cut(Tb + 1, bend);
            cut[0] = at;
            for (unsigned t = 1; t < Tb; t++) cut[t] = std::min(bend, next_cut(at + bsize / Tb * t, cut[t - 1]));
            std::vector<std::string> outs(Tb);
            std::vector<uint64_t> li(Tb, 0), lo(Tb, 0);
            std::vector<int> rcs(Tb, NTL_OK);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < Tb; t++)
                th.emplace_back([&, t]() {
                    Lifter L{agp, k};
                    rcs[t] = lift_range(cut[t], cut[t + 1], L, outs[t], li[t]);
                    lo[t] = L.lines_out;
                });
            for (auto &x : th) x.join();
            for (unsigned t = 0; t < Tb && rc == NTL_OK; t++) {
                if (rcs[t] != NTL_OK) { rc = rcs[t]; break; }
                nin += li[t]; nout += lo[t];
                const char *w = outs[t].data();
                size_t left = outs[t].size();
                while (left) {
                    ssize_t n = write(ofd, w, left);
                    if (n <= 0) { rc = NTL_EINVAL; break; }
                    w += n; left -= (size_t)n;
                }
            }
            at = bend;
        }
        munmap((void *)base, size);
    }
    close(fd);
    close(ofd);
    if (rc != NTL_OK) unlink(out_path);
    if (lines_in) *lines_in = nin;
    if (lines_out) *lines_out = nout;
    return rc;
}
