/*
 * Host tail of the pair stage in native code: the contig-pair tally that
 * bin/ntlink_pair.py keeps while it walks the reads (no GPU code; BASELINE north_star leaves the
 * scaffold graph on the CPU).  Restated from
 *   tally_pairs_from_mappings :416-435   add_pair :315-334   calculate_pair_info :222-239
 *   calculate_gap_size :157-187          normalize_pair :213-219   PairInfo :58-83
 * Input = the mapping records of ntl_map_run, in read order; the tally is order-sensitive (gap
 * lists keep the order of the reads, pairs the order of first appearance), so it runs on one
 * thread -- a few tens of nanoseconds per pair against microseconds in the Python loop.
 */
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <algorithm>
#include <numeric>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/ntlink_amd.h"

struct PairEntry {
    uint64_t key;
    uint32_t src, tgt;          /* contig indices as first seen (any contig carrying the name) */
    std::vector<int64_t> gaps;
    uint32_t anchor = 0;
};

struct ntl_tally {
    int k = 0, f = 10;
    std::vector<uint32_t> ctg_len;
    std::vector<uint32_t> rank;  /* dense rank of the contig's NAME in byte order; equal names share one */
    std::unordered_map<uint64_t, uint32_t> slot;
    std::vector<PairEntry> pairs; /* first-insertion order */
    uint64_t ngaps = 0;
    std::string names;               /* the contig ids, for the writers (ntl_tally_write) */
    std::vector<uint64_t> name_off;
};

extern "C" int ntl_tally_create(const char *ctg_names, const uint64_t *ctg_name_off, const uint32_t *ctg_len, uint64_t n_ctg,
                                int k, int f, ntl_tally **out)
{
    if (!out || (n_ctg && (!ctg_names || !ctg_name_off || !ctg_len)) || n_ctg >= ((uint64_t)1 << 30)) return NTL_EINVAL;
    ntl_tally *t = new ntl_tally();
    t->k = k; t->f = f;
    t->ctg_len.assign(ctg_len, ctg_len + n_ctg);
    if (n_ctg) { t->names.assign(ctg_names, ctg_names + ctg_name_off[n_ctg]); t->name_off.assign(ctg_name_off, ctg_name_off + n_ctg + 1); }
    std::vector<uint32_t> order(n_ctg);
    std::iota(order.begin(), order.end(), 0u);
    auto cmp3 = [&](uint32_t a, uint32_t b) { /* str.__lt__ on ids = byte order, shorter first on a tie */
        const uint64_t la = ctg_name_off[a + 1] - ctg_name_off[a], lb = ctg_name_off[b + 1] - ctg_name_off[b];
        const int c = memcmp(ctg_names + ctg_name_off[a], ctg_names + ctg_name_off[b], (size_t)std::min(la, lb));
        return c ? c : (la < lb ? -1 : la > lb ? 1 : 0);
    };
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cmp3(a, b) < 0; });
    t->rank.resize(n_ctg);
    uint32_t r = 0;
    for (uint64_t i = 0; i < n_ctg; i++) {
        if (i && cmp3(order[i - 1], order[i]) != 0) r++;
        t->rank[order[i]] = r;
    }
    *out = t;
    return NTL_OK;
}

extern "C" void ntl_tally_destroy(ntl_tally *t) { delete t; }

/* One batch of mappings (reads in order; read_len is indexed by ntl_mapping.read).  Returns NTL_ERANGE
 * when an overhang of an evaluated pair comes out negative -- the reference's "Gap distance estimation less than 0". */
/* ENDS: `hits` holds two entries per mapping, its first and its last hit (ntl_text_download), instead of all of them */
template <bool ENDS>
static int tally_add(ntl_tally *t, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *hits, const uint32_t *read_len)
{
    if (!t || (n_maps && (!maps || !hits || !read_len))) return NTL_EINVAL;
    auto first_of = [&](uint64_t i) -> const ntl_hit & { return ENDS ? hits[2 * i] : hits[maps[i].hit_off]; };
    auto last_of = [&](uint64_t i) -> const ntl_hit & { return ENDS ? hits[2 * i + 1] : hits[maps[i].hit_off + maps[i].n_hits - 1]; };
    if (n_maps < 2) return NTL_OK;
    const int64_t k = t->k;
    /* per mapping: overhang behind its terminal hit when it is the source of a pair (a), in front of its
     * first hit when it is the target (b); bin/ntlink_pair.py:394-406,317-320 */
    std::vector<int64_t> a(n_maps), b(n_maps);
    std::vector<uint8_t> sp(n_maps), tp(n_maps);
    for (uint64_t i = 0; i < n_maps; i++) {
        const ntl_mapping &m = maps[i];
        if (m.ctg >= t->ctg_len.size() || m.n_hits == 0) return NTL_EINVAL;
        const ntl_hit &hf = first_of(i), &hl = last_of(i);
        const int64_t clen = t->ctg_len[m.ctg];
        sp[i] = hl.read_strand == hl.ctg_strand;
        tp[i] = hf.read_strand == hf.ctg_strand;
        a[i] = sp[i] ? clen - (int64_t)hl.ctg_pos - k : (int64_t)hl.ctg_pos;
        b[i] = tp[i] ? (int64_t)hf.ctg_pos : clen - (int64_t)hf.ctg_pos - k;
    }
    std::unordered_set<uint64_t> added;
    std::vector<uint64_t> strong;
    uint64_t s0 = 0;
    while (s0 < n_maps) {
        uint64_t s1 = s0 + 1;
        while (s1 < n_maps && maps[s1].read == maps[s0].read) s1++;
        const uint64_t m = s1 - s0;
        if (m > 1) {
            const int64_t rl = read_len[maps[s0].read];
            /* returns the normalized key, or 0 when the pair was not recorded (keys are never 0: see below) */
            bool negative = false; /* the reference asserts a >= 0 and b >= 0 inside calculate_gap_size (:173-184): only for
                                      pairs it evaluates, before the |gap| > read length test */
            auto add = [&](uint64_t i, uint64_t j, const std::unordered_set<uint64_t> *check) -> uint64_t {
                if (a[i] < 0 || b[j] < 0) { negative = true; return 0; }
                const ntl_hit &hf = first_of(j), &hl = last_of(i);
                const int64_t gap = (int64_t)hf.read_pos - (int64_t)hl.read_pos - a[i] - b[j];
                uint32_t ci = maps[i].ctg, cj = maps[j].ctg;
                uint32_t so = sp[i], to = tp[j];
                if (!(t->rank[ci] < t->rank[cj])) { /* normalize_pair: smaller name first, both orientations flipped */
                    std::swap(ci, cj);
                    const uint32_t x = so; so = to ^ 1u; to = x ^ 1u;
                }
                const uint64_t key = ((uint64_t)t->rank[ci] << 33) | ((uint64_t)so << 32) | ((uint64_t)t->rank[cj] << 1) | to | (1ull << 63);
                if ((gap < 0 ? -gap : gap) > rl) return 0;
                if (check && check->count(key)) return 0;
                auto it = t->slot.find(key);
                uint32_t e;
                if (it == t->slot.end()) {
                    e = (uint32_t)t->pairs.size();
                    t->slot.emplace(key, e);
                    t->pairs.emplace_back();
                    t->pairs[e].key = key; t->pairs[e].src = ci; t->pairs[e].tgt = cj;
                } else e = it->second;
                t->pairs[e].gaps.push_back(gap);
                t->ngaps++;
                if (maps[i].n_hits > 1 && maps[j].n_hits > 1) t->pairs[e].anchor++;
                return key;
            };
            if (m <= (uint64_t)t->f) {
                for (uint64_t i = s0; i < s1 && !negative; i++)
                    for (uint64_t j = i + 1; j < s1 && !negative; j++) add(i, j, nullptr);
            } else { /* long chains: neighbours, then neighbours among the contigs with more than one hit */
                added.clear();
                for (uint64_t i = s0; i + 1 < s1; i++) {
                    const uint64_t key = add(i, i + 1, nullptr);
                    if (key) added.insert(key);
                }
                strong.clear();
                for (uint64_t i = s0; i < s1; i++) if (maps[i].n_hits > 1) strong.push_back(i);
                for (size_t q = 0; q + 1 < strong.size(); q++) add(strong[q], strong[q + 1], &added);
            }
            if (negative) return NTL_ERANGE;
        }
        s0 = s1;
    }
    return NTL_OK;
}

extern "C" int ntl_tally_add(ntl_tally *t, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *hits, const uint32_t *read_len)
{
    return tally_add<false>(t, maps, n_maps, hits, read_len);
}

extern "C" int ntl_tally_add_ends(ntl_tally *t, const ntl_mapping *maps, uint64_t n_maps, const ntl_hit *ends, const uint32_t *read_len)
{
    return tally_add<true>(t, maps, n_maps, ends, read_len);
}

extern "C" uint64_t ntl_tally_npairs(const ntl_tally *t) { return t ? t->pairs.size() : 0; }
extern "C" uint64_t ntl_tally_ngaps(const ntl_tally *t) { return t ? t->ngaps : 0; }

/* All pairs in order of first appearance: contig indices + orientations (1 = '+') of the normalized
 * pair, anchor count, and gaps[gap_off[i] .. gap_off[i+1]) in read order. */
extern "C" int ntl_tally_export(const ntl_tally *t, uint32_t *src, uint8_t *src_ori, uint32_t *tgt, uint8_t *tgt_ori,
                                uint32_t *anchor, uint64_t *gap_off, int64_t *gaps)
{
    if (!t || !gap_off) return NTL_EINVAL;
    uint64_t g = 0;
    gap_off[0] = 0;
    for (size_t i = 0; i < t->pairs.size(); i++) {
        const PairEntry &e = t->pairs[i];
        src[i] = e.src; tgt[i] = e.tgt;
        src_ori[i] = (uint8_t)((e.key >> 32) & 1u);
        tgt_ori[i] = (uint8_t)(e.key & 1u);
        anchor[i] = e.anchor;
        if (!e.gaps.empty()) memcpy(gaps + g, e.gaps.data(), e.gaps.size() * sizeof(int64_t));
        g += e.gaps.size();
        gap_off[i + 1] = g;
    }
    return NTL_OK;
}

/* Appends the export of another tally (same contigs, same k and f) that covers LATER reads: pairs it saw first come
 * behind the ones known here, its gaps behind the gaps of the same pair, anchors add up -- the state one tally would
 * have after walking both read ranges in order.  Multi-GPU driver: every rank tallies its own reads, rank 0 merges the
 * exports in rank order (= read order). */
extern "C" int ntl_tally_merge(ntl_tally *t, uint64_t npairs, const uint32_t *src, const uint8_t *src_ori, const uint32_t *tgt,
                               const uint8_t *tgt_ori, const uint32_t *anchor, const uint64_t *gap_off, const int64_t *gaps)
{
    if (!t || (npairs && (!src || !src_ori || !tgt || !tgt_ori || !anchor || !gap_off))) return NTL_EINVAL;
    for (uint64_t i = 0; i < npairs; i++) {
        if (src[i] >= t->rank.size() || tgt[i] >= t->rank.size() || gap_off[i + 1] < gap_off[i]) return NTL_EINVAL;
        const uint64_t key = ((uint64_t)t->rank[src[i]] << 33) | ((uint64_t)(src_ori[i] & 1u) << 32) | ((uint64_t)t->rank[tgt[i]] << 1) |
                             (uint64_t)(tgt_ori[i] & 1u) | (1ull << 63);
        auto it = t->slot.find(key);
        uint32_t e;
        if (it == t->slot.end()) {
            e = (uint32_t)t->pairs.size();
            t->slot.emplace(key, e);
            t->pairs.emplace_back();
            t->pairs[e].key = key; t->pairs[e].src = src[i]; t->pairs[e].tgt = tgt[i];
        } else e = it->second;
        const uint64_t ng = gap_off[i + 1] - gap_off[i];
        if (ng && !gaps) return NTL_EINVAL;
        t->pairs[e].gaps.insert(t->pairs[e].gaps.end(), gaps + gap_off[i], gaps + gap_off[i + 1]);
        t->ngaps += ng;
        t->pairs[e].anchor += anchor[i];
    }
    return NTL_OK;
}

/* ---- the two small files behind the tally, written natively (round 5: the Python loops over the pairs were 15 % of a 32-Gbases
 * file-to-file run) ----
 * <prefix>.pairs.tsv (write_pairs, bin/ntlink_pair.py:490-496) and <prefix>.n<n>.scaffold.dot (build_scaffold_graph :263-305,
 * filter_graph_global :498-506, print_directed_graph :133-155) from the pairs that pass filter_pairs_distances (:247-255: the gap
 * estimate must be larger than minus either contig's length) and filter_weak_anchor_pairs (:241-244: anchor >= a).
 * gap estimate = int(numpy.median(gaps)): the middle element, or the mean of the two middle ones, truncated toward zero (:70-74).
 * The dot file keeps what the Python dicts keep: sources and vertices in order of first appearance, an edge named twice keeps its
 * place and takes the later value; "scaf_num" = the largest N of a contig called ntLink_N, else None. */
namespace {
thread_local int t_io_errno = 0;

struct Buf {
    FILE *f;
    std::string s;
    bool bad = false; /* a short write: the stream's error flag alone is not enough (fclose may still return 0) */
    explicit Buf(FILE *fh) : f(fh) { s.reserve(1 << 20); }
    void flush()
    {
        if (!s.empty()) {
            if (!bad && fwrite(s.data(), 1, s.size(), f) != s.size()) { bad = true; t_io_errno = errno ? errno : EIO; }
            s.clear();
        }
    }
    void put(const char *p, size_t n) { s.append(p, n); if (s.size() > (1u << 20) - 4096) flush(); }
    void put(const std::string &x) { put(x.data(), x.size()); }
    void put(const char *z) { put(z, strlen(z)); }
    void num(long long v) { char b[24]; int n = snprintf(b, sizeof b, "%lld", v); put(b, (size_t)n); }
};

/* a file written beside its place and renamed when complete: never a partial .pairs.tsv / .dot */
struct TmpFile {
    std::string path, tmp;
    FILE *fh = nullptr;
    explicit TmpFile(const char *p) : path(p), tmp(std::string(p) + ".tmp." + std::to_string((long long)getpid()))
    {
        fh = fopen(tmp.c_str(), "w");
        if (!fh) t_io_errno = errno ? errno : EIO;
    }
    int finish(Buf &o)
    {
        o.flush();
        bool ok = !o.bad && !ferror(fh);
        if (fclose(fh) != 0 && ok) { ok = false; t_io_errno = errno ? errno : EIO; }
        fh = nullptr;
        if (ok && rename(tmp.c_str(), path.c_str()) != 0) { ok = false; t_io_errno = errno ? errno : EIO; }
        if (!ok) { if (!t_io_errno) t_io_errno = EIO; (void)remove(tmp.c_str()); return NTL_EIO; }
        return NTL_OK;
    }
    ~TmpFile() { if (fh) { fclose(fh); (void)remove(tmp.c_str()); } }
};

long long gap_estimate(const std::vector<int64_t> &gaps, std::vector<int64_t> &tmp)
{
    tmp = gaps;
    const size_t n = tmp.size(), m = n / 2;
    std::nth_element(tmp.begin(), tmp.begin() + (long)m, tmp.end());
    if (n & 1) return (long long)tmp[m];
    const int64_t hi = tmp[m], lo = *std::max_element(tmp.begin(), tmp.begin() + (long)m);
    const double med = ((double)lo + (double)hi) / 2.0; /* numpy: mean of the two middle values, in float64 */
    return (long long)med;                              /* int(): toward zero */
}
} // namespace

extern "C" int ntl_tally_write(const ntl_tally *t, int a, int min_n, const char *pairs_path, const char *dot_path, uint64_t *n_kept)
{
    if (!t) return NTL_EINVAL;
    const size_t nc = t->ctg_len.size();
    auto name = [&](uint32_t c) { return std::string(t->names.data() + t->name_off[c], (size_t)(t->name_off[c + 1] - t->name_off[c])); };
    /* name -> length as dict(zip(names, lengths)) has it: the last contig of a name wins */
    std::unordered_map<std::string, uint32_t> len_of;
    len_of.reserve(nc * 2);
    for (size_t c = 0; c < nc; c++) len_of[name((uint32_t)c)] = t->ctg_len[c];
    struct Kept { uint32_t e; long long d; };
    std::vector<Kept> kept;
    std::vector<int64_t> tmp;
    for (size_t i = 0; i < t->pairs.size(); i++) {
        const PairEntry &e = t->pairs[i];
        if (e.gaps.empty()) continue; /* (a pair is only ever made with a gap) */
        const long long g = gap_estimate(e.gaps, tmp);
        if (g <= -(long long)len_of[name(e.src)] || g <= -(long long)len_of[name(e.tgt)]) continue;
        if ((long long)e.anchor < (long long)a) continue;
        kept.push_back({(uint32_t)i, g});
    }
    if (n_kept) *n_kept = kept.size();
    auto ori = [](const PairEntry &e, bool src) { return ((src ? (e.key >> 32) : e.key) & 1u) ? '+' : '-'; };
    t_io_errno = 0;
    if (pairs_path) {
        TmpFile tf(pairs_path);
        if (!tf.fh) return NTL_EIO;
        Buf o(tf.fh);
        for (const Kept &kp : kept) {
            const PairEntry &e = t->pairs[kp.e];
            o.put(name(e.src)); o.put(ori(e, true) == '+' ? "+" : "-"); o.put("\t");
            o.put(name(e.tgt)); o.put(ori(e, false) == '+' ? "+" : "-"); o.put("\tn=");
            o.num((long long)e.gaps.size()); o.put(", gap_estimates=[");
            for (size_t j = 0; j < e.gaps.size(); j++) { if (j) o.put(", "); o.num((long long)e.gaps[j]); }
            o.put("], anchor="); o.num((long long)e.anchor); o.put("\n");
        }
        if (int rc = tf.finish(o)) return rc;
    }
    if (dot_path) {
        std::vector<std::string> vertices;            /* insertion order */
        std::unordered_map<std::string, uint32_t> vid;
        struct Edge { uint32_t to; long long d; uint64_t n; };
        std::vector<std::pair<uint32_t, std::vector<Edge>>> adj; /* sources in insertion order, their targets in insertion order */
        std::unordered_map<uint32_t, uint32_t> src_at;
        auto vertex = [&](const std::string &v) {
            auto it = vid.find(v);
            if (it != vid.end()) return it->second;
            const uint32_t id = (uint32_t)vertices.size();
            vid.emplace(v, id); vertices.push_back(v);
            return id;
        };
        auto edge = [&](uint32_t from, uint32_t to, long long d, uint64_t n) {
            auto it = src_at.find(from);
            if (it == src_at.end()) { it = src_at.emplace(from, (uint32_t)adj.size()).first; adj.push_back({from, {}}); }
            for (Edge &x : adj[it->second].second)
                if (x.to == to) { x.d = d; x.n = n; return; }
            adj[it->second].second.push_back({to, d, n});
        };
        for (const Kept &kp : kept) {
            const PairEntry &e = t->pairs[kp.e];
            const std::string s = name(e.src), g = name(e.tgt);
            const char so = ori(e, true), to = ori(e, false);
            const uint32_t f0 = vertex(s + so), f1 = vertex(g + to), r0 = vertex(g + (to == '+' ? '-' : '+')), r1 = vertex(s + (so == '+' ? '-' : '+'));
            edge(f0, f1, kp.d, e.gaps.size());
            edge(r0, r1, kp.d, e.gaps.size());
        }
        bool have = false;
        unsigned long long largest = 0;
        for (size_t c = 0; c < nc; c++) { /* ^ntLink_(\d+)$ */
            const std::string nm = name((uint32_t)c);
            if (nm.size() > 7 && nm.compare(0, 7, "ntLink_") == 0 && std::all_of(nm.begin() + 7, nm.end(), [](char ch) { return ch >= '0' && ch <= '9'; })) {
                const unsigned long long v = strtoull(nm.c_str() + 7, nullptr, 10);
                if (!have || v > largest) { largest = v; have = true; }
            }
        }
        TmpFile tf(dot_path);
        if (!tf.fh) return NTL_EIO;
        Buf o(tf.fh);
        o.put("digraph G {\ngraph [scaf_num=");
        if (have) o.num((long long)largest); else o.put("None");
        o.put("]\n");
        for (const std::string &v : vertices) {
            o.put("\""); o.put(v); o.put("\" [l="); o.num((long long)len_of[v.substr(0, v.size() - 1)]); o.put("]\n");
        }
        for (auto &sa : adj)
            for (const Edge &x : sa.second)
                if (x.n >= (uint64_t)std::max(min_n, 0)) {
                    o.put("\""); o.put(vertices[sa.first]); o.put("\" -> \""); o.put(vertices[x.to]); o.put("\" [d="); o.num(x.d);
                    o.put(" e=100 n="); o.num((long long)x.n); o.put("]\n");
                }
        o.put("}\n");
        if (int rc = tf.finish(o)) return rc;
    }
    return NTL_OK;
}

extern "C" int ntl_io_errno(void) { return t_io_errno; }
