"""Host tail of the pair stage: contig-pair tally, filters, `.pairs.tsv`, `.scaffold.dot`.

Stays on the CPU by design (BASELINE.json north_star: "scaffold-graph ... stay on CPU"); it consumes
the mapping records the GPU produced.  Behaviour restated from bin/ntlink_pair.py:
  tally_pairs_from_mappings :416-435   add_pair :315-334   calculate_pair_info :222-239
  calculate_gap_size :157-187          normalize_pair :213-219   PairInfo :58-83
  filter_pairs_distances :247-255      filter_weak_anchor_pairs :241-244   write_pairs :490-496
  build_scaffold_graph :263-305        filter_graph_global :498-506        print_directed_graph :133-155
The per-mapping quantities (orientation and overhang of a contig used as source / as target) are
computed for all mappings at once with numpy; only reads that touch >= 2 contigs enter the Python loop.
"""
import re

import numpy as np


class PairTally:
    """Accumulates pairs over batches of reads, in read order (gap lists are order-sensitive)."""

    def __init__(self, ctg_names, ctg_len, k, f=10):
        self.names = ctg_names
        self.ctg_len = np.asarray(ctg_len, np.int64)
        self.k, self.f = int(k), int(f)
        self.pairs = {}  # (src, src_ori, tgt, tgt_ori) -> [gaps, anchor]; insertion-ordered

    def add_batch(self, res, read_len):
        maps, hits = res["maps"], res["hits"]
        if len(maps) < 2:
            return
        rd = maps["read"].astype(np.int64)
        # reads with at least two accepted contigs
        starts = np.flatnonzero(np.r_[True, rd[1:] != rd[:-1]])
        counts = np.diff(np.r_[starts, len(rd)])
        first = maps["hit_off"].astype(np.int64)
        last = first + maps["n_hits"].astype(np.int64) - 1
        ctg = maps["ctg"].astype(np.int64)
        clen = self.ctg_len[ctg]
        nh = maps["n_hits"].astype(np.int64)
        hl, hf = hits[last], hits[first]
        # as source: terminal hit; as target: first hit (bin/ntlink_pair.py:394-406,317-320)
        s_plus = hl["read_strand"] == hl["ctg_strand"]
        t_plus = hf["read_strand"] == hf["ctg_strand"]
        a = np.where(s_plus, clen - hl["ctg_pos"].astype(np.int64) - self.k, hl["ctg_pos"].astype(np.int64))
        b = np.where(t_plus, hf["ctg_pos"].astype(np.int64), clen - hf["ctg_pos"].astype(np.int64) - self.k)
        rp_last = hl["read_pos"].astype(np.int64)
        rp_first = hf["read_pos"].astype(np.int64)
        if (a < 0).any() or (b < 0).any():
            raise AssertionError("Gap distance estimation less than 0")  # bin/ntlink_pair.py:173-184
        names, pairs, f = self.names, self.pairs, self.f
        for s0, m in zip(starts[counts > 1].tolist(), counts[counts > 1].tolist()):
            rl = int(read_len[rd[s0]])

            def add(i, j, check=None):
                i, j = s0 + i, s0 + j
                gap = int(rp_first[j] - rp_last[i] - a[i] - b[j])
                so = "+" if s_plus[i] else "-"
                to = "+" if t_plus[j] else "-"
                ni, nj = names[ctg[i]], names[ctg[j]]
                if ni < nj:
                    key = (ni, so, nj, to)
                else:  # normalize_pair: lexicographically smaller NAME first, orientations flipped
                    key = (nj, "-" if to == "+" else "+", ni, "-" if so == "+" else "+")
                if abs(gap) > rl:
                    return None
                if check is not None and key in check:
                    return None
                e = pairs.get(key)
                if e is None:
                    e = pairs[key] = [[], 0]
                e[0].append(gap)
                if nh[i] > 1 and nh[j] > 1:
                    e[1] += 1
                return key

            if m <= f:
                for i in range(m):
                    for j in range(i + 1, m):
                        add(i, j)
            else:
                added = set()
                for i in range(m - 1):
                    added.add(add(i, i + 1))
                strong = [i for i in range(m) if nh[s0 + i] > 1]
                for i, j in zip(strong, strong[1:]):
                    add(i, j, check=added)

    def add_checkpoint_read(self, entries, ctg_index):
        """A read re-read from <prefix>.verbose_mapping.tsv (bin/ntlink_pair.py:460-488): the read
        length is replaced by the largest mapped read position."""
        n = len(entries)
        maps = np.zeros(n, dtype=[("read", "<u4"), ("ctg", "<u4"), ("n_hits", "<u4"), ("pad", "<u4"), ("hit_off", "<u8")])
        hl = []
        rmax = 0
        for i, (contig, hs) in enumerate(entries):
            maps[i] = (0, ctg_index[contig], len(hs), 0, len(hl))
            hl.extend(hs)
            rmax = max(rmax, hs[0][2], hs[-1][2])
        hits = np.zeros(len(hl), dtype=[("ctg_pos", "<u4"), ("read_pos", "<u4"), ("ctg_strand", "u1"), ("read_strand", "u1")])
        for j, (cp, cs, rp, rs) in enumerate(hl):
            hits[j] = (cp, rp, cs, rs)
        self.add_batch({"maps": maps, "hits": hits}, [rmax])

    # ---- filters and writers -------------------------------------------------------------

    @staticmethod
    def gap_estimate(gaps):
        return int(np.median(gaps))  # truncates toward zero (bin/ntlink_pair.py:70-74)

    def filtered(self, a=1):
        by = dict(zip(self.names, self.ctg_len.tolist()))
        out = {}
        for key, (gaps, anchor) in self.pairs.items():
            g = self.gap_estimate(gaps)
            if g <= -by[key[0]] or g <= -by[key[2]]:
                continue
            if anchor < a:
                continue
            out[key] = (gaps, anchor)
        return out


def write_pairs(fh, pairs):
    for (s, so, t, to), (gaps, anchor) in pairs.items():
        fh.write(f"{s}{so}\t{t}{to}\tn={len(gaps)}, gap_estimates={gaps}, anchor={anchor}\n")


def _flip(o):
    return "-" if o == "+" else "+"


def write_dot(fh, pairs, ctg_names, ctg_len, min_n=1):
    """Each pair gives the edge and its reverse complement; sources in first-insertion order; every
    vertex of a kept or dropped edge is listed (node order in the reference is Python-set order)."""
    length = dict(zip(ctg_names, (int(v) for v in ctg_len)))
    edges, vertices = {}, {}
    for (s, so, t, to), (gaps, _anchor) in pairs.items():
        fwd = (s + so, t + to)
        rc = (t + _flip(to), s + _flip(so))
        for v in (fwd[0], fwd[1], rc[0], rc[1]):
            vertices.setdefault(v, None)
        d, n = PairTally.gap_estimate(gaps), len(gaps)
        edges.setdefault(fwd[0], {})[fwd[1]] = (d, n)
        edges.setdefault(rc[0], {})[rc[1]] = (d, n)
    largest = None
    pat = re.compile(r"^ntLink_(\d+)$")
    for name in ctg_names:
        mm = pat.search(name)
        if mm and (largest is None or int(mm.group(1)) > largest):
            largest = int(mm.group(1))
    fh.write("digraph G {\n")
    fh.write(f"graph [scaf_num={largest}]\n")
    for v in vertices:
        fh.write(f"\"{v}\" [l={length[v[:-1]]}]\n")
    for a in edges:
        for b, (d, n) in edges[a].items():
            if n >= min_n:
                fh.write(f"\"{a}\" -> \"{b}\" [d={d} e=100 n={n}]\n")
    fh.write("}\n")
