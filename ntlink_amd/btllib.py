"""The slice of btllib's Python API that ntLink uses in-process, on the MI355X sketch.

    import ntlink_amd.btllib as btllib
    with btllib.Indexlr(path, k, w, btllib.IndexlrFlag.LONG_MODE, threads) as recs:
        for rec in recs: rec.id, rec.readlen, [(m.out_hash, m.pos, m.forward) for m in rec.minimizers]
        rec = recs.read()

Reference call sites: bin/ntlink_patch_gaps.py:397-441 (records are mutable there: `.id` is reassigned).
"""
from . import seqio

BATCH_BASES = 500_000_000


class IndexlrFlag:
    NO_ID, BX, SEQ, FILTER_IN, FILTER_OUT, SHORT_MODE, LONG_MODE = 1, 2, 4, 8, 16, 32, 64


class Minimizer:
    __slots__ = ("min_hash", "out_hash", "pos", "forward", "seq")

    def __init__(self, out_hash, pos, forward):
        self.min_hash = None  # the selection hash is not kept on the device path
        self.out_hash, self.pos, self.forward, self.seq = out_hash, pos, forward, ""


class Record:
    __slots__ = ("num", "id", "barcode", "readlen", "minimizers")

    def __init__(self, num, id_, readlen, minimizers):
        self.num, self.id, self.barcode, self.readlen, self.minimizers = num, id_, "", readlen, minimizers

    def __bool__(self):
        return True


class Indexlr:
    """Iterates the records of a FASTA/FASTQ(.gz) file with their (k,w) minimizers, input order."""

    def __init__(self, seqfile, k, w, flags=IndexlrFlag.LONG_MODE, threads=1, verbose=False, device=None):
        # the reference only ever passes LONG_MODE (bin/ntlink_patch_gaps.py:417-420); btllib's other modes (barcodes, Bloom-filter
        # filtering, kept sequences, short-read mode) are not this path's and must not be taken for granted silently
        if int(flags) != IndexlrFlag.LONG_MODE:
            raise NotImplementedError(f"ntlink_amd.btllib.Indexlr supports flags=IndexlrFlag.LONG_MODE only (got {int(flags)})")
        from . import capi
        self._own = device is None
        self.dev = device if device is not None else capi.Device(0)
        self.k, self.w = int(k), int(w)
        self._gen = self._records(seqfile)

    def _records(self, path):
        num = 0
        for ss in seqio.load([path], max_bases=BATCH_BASES):
            if not len(ss):
                continue
            with self.dev.batch(ss.buf, ss.offsets) as b, self.dev.sketch(b, self.k, self.w) as sk:
                off, h, p, s = sk.download()
            hs, ps, fs = h.tolist(), p.tolist(), s.tolist()
            o = off.tolist()
            ln = ss.lengths.tolist()
            for i, name in enumerate(ss.names):
                yield Record(num, name, ln[i], [Minimizer(hs[j], ps[j], bool(fs[j])) for j in range(o[i], o[i + 1])])
                num += 1

    def read(self):
        """Next record, or None at the end (btllib returns a falsy record)."""
        return next(self._gen, None)

    def __iter__(self):
        return self._gen

    def close(self):
        self._gen.close()
        if self._own and self.dev is not None:
            self.dev.close()
            self.dev = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
