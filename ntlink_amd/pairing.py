"""Host tail of the pair stage: contig-pair tally, filters, `.pairs.tsv`, `.scaffold.dot`.

Stays on the CPU by design (BASELINE.json north_star: "scaffold-graph ... stay on CPU"); it consumes
the mapping records the GPU produced.  Behaviour restated from bin/ntlink_pair.py:
  tally_pairs_from_mappings :416-435   add_pair :315-334   calculate_pair_info :222-239
  calculate_gap_size :157-187          normalize_pair :213-219   PairInfo :58-83
  filter_pairs_distances :247-255      filter_weak_anchor_pairs :241-244   write_pairs :490-496
  build_scaffold_graph :263-305        filter_graph_global :498-506        print_directed_graph :133-155
The order-sensitive per-read tally itself is native (csrc/ntl_pairs.cpp); filters and the two small
writers work on its export.
"""
import ctypes as C
import os
import re

import numpy as np


class PairTally:
    """Accumulates pairs over batches of reads, in read order (gap lists are order-sensitive).  The
    per-read loop is native (csrc/ntl_pairs.cpp, ntl_tally_*); this class feeds it record arrays and
    turns its export into the dict the writers below take."""

    def __init__(self, ctg_names, ctg_len, k, f=10):
        from . import capi
        from .seqio import Names
        self.names = ctg_names
        self.ctg_len = np.asarray(ctg_len, np.int64)
        self.k, self.f = int(k), int(f)
        self._L = capi.load()
        self._maps_dt, self._hits_dt = capi.MAPPING_DT, capi.HIT_DT
        nm = Names.of(ctg_names)
        cl = np.ascontiguousarray(ctg_len, np.uint32)
        h = C.c_void_p()
        blob = np.ascontiguousarray(nm.blob)
        rc = self._L.ntl_tally_create(blob.ctypes.data if len(blob) else None, nm.off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      cl.ctypes.data_as(C.POINTER(C.c_uint32)), len(nm), self.k, self.f, C.byref(h))
        if rc != 0:
            raise ValueError(f"ntl_tally_create failed with {rc}")
        self._h = h
        self._pairs = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.ntl_tally_destroy(h)

    @staticmethod
    def _as(arr, dt):
        if arr.dtype == dt:
            return np.ascontiguousarray(arr)
        out = np.zeros(len(arr), dt)
        for name in arr.dtype.names:
            out[name] = arr[name]
        return out

    def add_batch_ends(self, maps, ends, read_len):
        """The same from the first and last hit of every mapping (capi.Text.download: ends[2 i], ends[2 i + 1])."""
        maps, ends = self._as(maps, self._maps_dt), self._as(ends, self._hits_dt)
        rl = np.ascontiguousarray(read_len, np.uint32)
        rc = self._L.ntl_tally_add_ends(self._h, maps.ctypes.data, len(maps), ends.ctypes.data, rl.ctypes.data_as(C.POINTER(C.c_uint32)))
        self._pairs = None
        if rc == -5:
            raise AssertionError("Gap distance estimation less than 0")
        if rc != 0:
            raise ValueError(f"ntl_tally_add_ends failed with {rc}")

    def add_batch(self, res, read_len):
        maps, hits = self._as(res["maps"], self._maps_dt), self._as(res["hits"], self._hits_dt)
        rl = np.ascontiguousarray(read_len, np.uint32)
        self._pairs = None
        rc = self._L.ntl_tally_add(self._h, maps.ctypes.data, len(maps), hits.ctypes.data, rl.ctypes.data_as(C.POINTER(C.c_uint32)))
        if rc == -5:
            raise AssertionError("Gap distance estimation less than 0")  # bin/ntlink_pair.py:173-184
        if rc != 0:
            raise ValueError(f"ntl_tally_add failed with {rc}")

    def export(self):
        """(src u32, src_ori u8, tgt u32, tgt_ori u8, anchor u32, gap_off u64[n+1], gaps i64): every pair in order of
        first appearance, contig indices, 1 = '+'.  The form merge() takes -- small enough to send between ranks."""
        L, h = self._L, self._h
        n, g = int(L.ntl_tally_npairs(h)), int(L.ntl_tally_ngaps(h))
        src, tgt, anchor = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        so, to = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        goff, gaps = np.zeros(n + 1, np.uint64), np.zeros(g, np.int64)
        L.ntl_tally_export(h, src.ctypes.data, so.ctypes.data, tgt.ctypes.data, to.ctypes.data, anchor.ctypes.data,
                           goff.ctypes.data, gaps.ctypes.data)
        return src, so, tgt, to, anchor, goff, gaps

    def merge(self, exported):
        """Appends the export() of a tally that covers LATER reads (multi-GPU driver: rank order = read order)."""
        src, so, tgt, to, anchor, goff, gaps = (np.ascontiguousarray(x) for x in exported)
        self._pairs = None
        rc = self._L.ntl_tally_merge(self._h, len(src), src.ctypes.data, so.ctypes.data, tgt.ctypes.data, to.ctypes.data,
                                     anchor.ctypes.data, goff.ctypes.data, gaps.ctypes.data if len(gaps) else None)
        if rc != 0:
            raise ValueError(f"ntl_tally_merge failed with {rc}")

    @property
    def pairs(self):
        """(src, src_ori, tgt, tgt_ori) -> [gaps, anchor] in order of first appearance."""
        if self._pairs is None:
            src, so, tgt, to, anchor, goff, gaps = self.export()
            names = self.names.tolist() if hasattr(self.names, "tolist") else list(self.names)
            gl, go = gaps.tolist(), goff.tolist()
            out = {}
            for i, (a, ao, b, bo, an) in enumerate(zip(src.tolist(), so.tolist(), tgt.tolist(), to.tolist(), anchor.tolist())):
                out[(names[a], "+" if ao else "-", names[b], "+" if bo else "-")] = [gl[go[i]:go[i + 1]], an]
            self._pairs = out
        return self._pairs

    def add_checkpoint_read(self, entries, ctg_index):
        """A read re-read from <prefix>.verbose_mapping.tsv (bin/ntlink_pair.py:460-488): the read
        length is replaced by the largest mapped read position."""
        n = len(entries)
        maps = np.zeros(n, self._maps_dt)
        hl = []
        rmax = 0
        for i, (contig, hs) in enumerate(entries):
            maps[i] = (0, ctg_index[contig], len(hs), 0, len(hl))
            hl.extend(hs)
            rmax = max(rmax, hs[0][2], hs[-1][2])
        hits = np.zeros(len(hl), self._hits_dt)
        if hl:
            cols = np.array(hl, np.int64)
            hits["ctg_pos"], hits["ctg_strand"], hits["read_pos"], hits["read_strand"] = cols[:, 0], cols[:, 1], cols[:, 2], cols[:, 3]
        self.add_batch({"maps": maps, "hits": hits}, [rmax])

    def write(self, a, min_n, pairs_path, dot_path):
        """<prefix>.pairs.tsv (pairs_path may be None) and <prefix>.n<n>.scaffold.dot from the pairs that pass the two filters, written
        by the native tally (ntl_tally_write); the Python forms below (filtered / write_pairs / write_dot) give the same bytes and
        stay for callers that want the dict.  -> pairs kept."""
        kept = C.c_uint64()
        rc = self._L.ntl_tally_write(self._h, int(a), int(min_n), pairs_path.encode() if pairs_path else None,
                                     dot_path.encode() if dot_path else None, C.byref(kept))
        if rc == -6:  # NTL_EIO: with the errno, as the Python writers it replaces would raise
            err = int(self._L.ntl_io_errno())
            raise OSError(err, f"{os.strerror(err)}: writing {pairs_path} / {dot_path}")
        if rc != 0:
            raise OSError(f"ntl_tally_write failed with {rc} ({pairs_path}, {dot_path})")
        return int(kept.value)

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
