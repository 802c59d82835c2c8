"""`get_accepted_anchor_contigs` on the GPU (second half).

The gap-filling stage of the reference re-maps one chosen read against the two scaffolds around a
gap with the same function the pair stage uses (bin/ntlink_patch_gaps.py:397-441 ->
bin/ntlink_utils.py:200-268).  This module offers that call with the reference's signature and
return shape, computed by the pair stage's map kernel through the C ABI:

    accepted, order = get_accepted_anchor_contigs(mx_list, read_length, scaffolds, list_mx_info, args)

* ``mx_list``       [(hash, read_pos, read_strand)] of the read, hashes present in ``list_mx_info``
                    (hash as str or int, strand "+"/"-"), in read order
* ``scaffolds``     name -> object with ``.length``
* ``list_mx_info``  hash -> object with ``.contig .position .strand`` (duplicates already removed,
                    bin/ntlink_pair.py:189-211)
* ``args``          needs ``.k .z .x .sensitive``
* returns           ({contig: ContigRun}, [contig, ...]) with ``ContigRun.hits`` a list of
                    ``MinimizerPositions(mx, ctg_pos, ctg_strand, read_pos, read_strand)``

One call costs a few device round trips, which is fine for the thousands of gaps of an assembly;
`AnchorMapper.map_many` batches reads that share an index.
"""
from collections import namedtuple

import numpy as np

from . import capi

Minimizer = namedtuple("Minimizer", ["contig", "position", "strand"])  # bin/ntlink_pair.py:24
MinimizerPositions = namedtuple("MinimizerPositions", ["mx", "ctg_pos", "ctg_strand", "read_pos", "read_strand"])  # ntlink_utils.py:20

_STRAND = ("-", "+")


class ContigRun:
    """Fields of bin/ntlink_pair.py:85-113 that callers of get_accepted_anchor_contigs read."""

    def __init__(self, contig, list_hits):
        self.contig = contig
        self.index = None
        self.hit_count = len(list_hits)
        self.hits = list_hits
        self.subsumed = False
        self.first_mx = None
        self.terminal_mx = None


class AnchorMapper:
    """A contig-minimizer dict on the device; maps reads given as minimizer lists."""

    def __init__(self, list_mx_info, scaffolds, dev=None):
        self.dev = dev or _default_device()
        self.names = sorted({m.contig for m in list_mx_info.values()})
        cid = {n: i for i, n in enumerate(self.names)}
        items = sorted(((cid[m.contig], int(m.position), int(h), 1 if m.strand == "+" else 0, h)
                        for h, m in list_mx_info.items()), key=lambda t: (t[0], t[1]))
        off = np.zeros(len(self.names) + 1, np.uint64)
        for c, *_ in items:
            off[c + 1] += 1
        np.cumsum(off, out=off)
        self.ctg_len = np.array([int(scaffolds[n].length) for n in self.names], np.uint32)
        self._key_type = type(next(iter(list_mx_info))) if list_mx_info else str
        self._sketch = self.dev.sketch_from_arrays(off, np.array([t[2] for t in items], np.uint64),
                                                   np.array([t[1] for t in items], np.uint32),
                                                   np.array([t[3] for t in items], np.uint8))
        self._index = self.dev.index(self._sketch, self.ctg_len)

    def close(self):
        self._index.close()
        self._sketch.close()

    def map_many(self, reads, args):
        """reads: [(mx_list, read_length)] -> [(accepted dict, order list)] per read."""
        off = np.zeros(len(reads) + 1, np.uint64)
        hs, ps, ss = [], [], []
        for i, (mx_list, _rl) in enumerate(reads):
            off[i + 1] = off[i] + len(mx_list)
            for h, p, s in mx_list:
                hs.append(int(h)); ps.append(int(p)); ss.append(1 if s == "+" else 0)
        rlen = np.array([int(rl) for _m, rl in reads], np.uint32)
        with self.dev.sketch_from_arrays(off, np.array(hs, np.uint64), np.array(ps, np.uint32), np.array(ss, np.uint8)) as rsk, \
                self.dev.map(self._index, rsk, rlen, k=int(args.k), z=int(args.z), x=float(args.x),
                             sensitive=bool(args.sensitive), repeat_filter=False) as res:
            rec = res.download()
        out = [({}, []) for _ in reads]
        maps, hits = rec["maps"], rec["hits"]
        # hash of a hit: the read minimizer at that read position and strand (positions are unique per read)
        by_pos = [{(int(p), 1 if s == "+" else 0): h for h, p, s in mx_list} for mx_list, _rl in reads]
        for m in maps:
            r = int(m["read"])
            name = self.names[int(m["ctg"])]
            hl = []
            for h in hits[int(m["hit_off"]):int(m["hit_off"]) + int(m["n_hits"])]:
                key = by_pos[r][(int(h["read_pos"]), int(h["read_strand"]))]
                hl.append(MinimizerPositions(mx=key, ctg_pos=int(h["ctg_pos"]), ctg_strand=_STRAND[int(h["ctg_strand"])],
                                             read_pos=int(h["read_pos"]), read_strand=_STRAND[int(h["read_strand"])]))
            out[r][0][name] = ContigRun(name, hl)
            out[r][1].append(name)
        return out


_dev = None


def _default_device():
    global _dev
    if _dev is None:
        _dev = capi.Device(0)
    return _dev


def get_accepted_anchor_contigs(mx_list, read_length, scaffolds, list_mx_info, args, dev=None):
    """Drop-in for ntlink_utils.get_accepted_anchor_contigs (bin/ntlink_utils.py:200-268)."""
    mapper = AnchorMapper(list_mx_info, scaffolds, dev)
    try:
        return mapper.map_many([(mx_list, read_length)], args)[0]
    finally:
        mapper.close()
