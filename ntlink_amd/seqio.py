"""FASTA/FASTQ(.gz) input for the pair stage.

Record semantics follow what the reference feeds to indexlr / reads with bin/read_fasta.py:6-46:
the id is the header up to the first whitespace, multi-line sequences are joined, FASTQ qualities
are skipped, `gzip -cd -f` style transparent decompression (ntLink:113-117,222), several read
files are concatenated in the order given.
"""
import gzip
import io
import os
import sys
import time

import numpy as np


def _open(path):
    if path == "-":
        raw = sys.stdin.buffer
        head = raw.peek(2)[:2] if hasattr(raw, "peek") else b""
    else:
        raw = open(path, "rb")
        head = raw.peek(2)[:2]
    if head == b"\x1f\x8b":
        return gzip.open(raw, "rb")
    return raw


def read_fastx(path):
    """Yield (name:str, sequence:bytes)."""
    with _open(path) as f:
        fh = io.BufferedReader(f) if not isinstance(f, io.BufferedReader) else f
        last = None
        while True:
            if last is None:
                for line in fh:
                    if line[:1] in (b">", b"@"):
                        last = line
                        break
            if last is None:
                return
            parts = last[1:].split(None, 1)
            name = parts[0].decode() if parts else ""
            last = None
            seqs = []
            for line in fh:
                c = line[:1]
                if c in (b">", b"@", b"+"):
                    last = line
                    break
                seqs.append(line.rstrip(b"\r\n"))
            seq = b"".join(seqs)
            if last is None or last[:1] != b"+":
                yield name, seq
                if last is None:
                    return
            else:  # FASTQ: skip len(seq) quality bytes
                got = 0
                last = None
                for line in fh:
                    got += len(line.rstrip(b"\r\n"))
                    if got >= len(seq):
                        break
                yield name, seq


class Names:
    """Sequence ids as one byte blob + offsets[n+1] (the form the native reader and writers use), with
    list-like access for the Python host."""

    def __init__(self, blob, off):
        self.blob = blob if isinstance(blob, np.ndarray) else np.frombuffer(blob, np.uint8)
        self.off = np.ascontiguousarray(off, np.uint64)
        self._list = None

    @classmethod
    def from_list(cls, names):
        enc = [n.encode() for n in names]
        off = np.zeros(len(enc) + 1, np.uint64)
        if enc:
            np.cumsum(np.fromiter((len(e) for e in enc), np.uint64, len(enc)), out=off[1:])
        obj = cls(np.frombuffer(b"".join(enc), np.uint8) if enc else np.zeros(0, np.uint8), off)
        obj._list = list(names)
        return obj

    @classmethod
    def of(cls, names):
        return names if isinstance(names, cls) else cls.from_list(names)

    def tolist(self):
        if self._list is None:
            raw = self.blob.tobytes()
            o = self.off.tolist()
            self._list = [raw[o[i]:o[i + 1]].decode() for i in range(len(o) - 1)]
        return self._list

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            lo, hi, step = i.indices(len(self))
            assert step == 1
            b0, b1 = int(self.off[lo]), int(self.off[hi]) if hi > lo else int(self.off[lo])
            return Names(self.blob[b0:b1], self.off[lo:max(hi, lo) + 1] - np.uint64(b0))
        return self.tolist()[i]

    def __iter__(self):
        return iter(self.tolist())

    def __eq__(self, other):
        return self.tolist() == (other.tolist() if isinstance(other, Names) else list(other))


class SeqSet:
    """Names + one contiguous uint8 buffer + offsets: the form ntl_batch_create takes.  A set read with packed=True has
    buf = None and carries the device's layout instead (ntl_batch_create_packed): `packed` (uint32 words, 2 bits per base,
    in the buffer `pinned` that came from alloc) and the ACGT-run table seq_run_first / run_start / run_len."""

    def __init__(self, names, buf, offsets, packed=None, runs=None, pinned=None, positions=None, span_positions=0):
        self.names, self.buf, self.offsets = Names.of(names), buf, offsets
        self.packed, self.pinned = packed, pinned
        self.seq_run_first, self.run_start, self.run_len = runs if runs is not None else (None, None, None)
        # read in one pass (ntl_fastx_parse_span): where every sequence starts in the packed stream, and how far the stream spans
        self.positions, self.span_positions = positions, span_positions

    def __len__(self):
        return len(self.names)

    @property
    def lengths(self):
        return np.diff(self.offsets).astype(np.uint32)

    @property
    def bases(self):
        return int(self.offsets[-1])


def _np_empty(nbytes):
    return np.empty(nbytes, np.uint8)


def _open_native(L, item):
    """item: a path, or (path, lo, hi) = the records of a plain file whose first byte lies in [lo, hi)"""
    import ctypes as C
    h = C.c_void_p()
    if isinstance(item, tuple):
        path, lo, hi = item
        if L.ntl_fastx_open_range(path.encode(), int(lo), int(hi), C.byref(h)) != 0:
            raise OSError(f"cannot open bytes {lo}..{hi} of {path}")
        return h
    if L.ntl_fastx_open(item.encode(), C.byref(h)) != 0:
        raise OSError(f"cannot open {item}")
    return h


def _splittable(path):
    """A plain regular file (not gzip), or a BGZF file (bgzip: gzip members that carry their own size in a `BC` extra field):
    byte ranges of it can be read independently (ntl_fastx_open_range; for BGZF the ranges are in the compressed file and
    cut at member starts)."""
    try:
        if path == "-" or not os.path.isfile(path) or os.path.getsize(path) == 0:
            return False
        with open(path, "rb") as f:
            head = f.read(18)
        if head[:2] != b"\x1f\x8b":
            return True
        return len(head) >= 18 and head[2] == 8 and bool(head[3] & 4) and head[12:14] == b"BC" and not os.environ.get("NTL_IO_NO_BGZF")
    except OSError:
        return False


def shard_plan(paths, rank, world):
    """The share of rank `rank` of the read files concatenated in the order given (ntLink:222): the bytes
    [total*rank/world, total*(rank+1)/world) of the concatenation.  Plain files are cut at those offsets (the readers
    move the cuts to record starts: a record belongs to the range that holds its first byte); a file that cannot be cut
    (gzip, stdin) goes whole to the rank whose share holds its first byte.  -> list of paths / (path, lo, hi) in input
    order; the ranks' lists, taken in rank order, cover every record once, in input order."""
    if isinstance(paths, str):
        paths = [paths]
    if world == 1:
        return list(paths)
    import stat as _stat
    sizes, regular = [], []
    for p in paths:
        if p == "-":
            raise ValueError("reads from stdin cannot be shared out between ranks: give the files by name")
        st = os.stat(p)  # a missing input is an error here, as it is for the single-process reader
        regular.append(_stat.S_ISREG(st.st_mode))
        sizes.append(st.st_size if regular[-1] else 0)
    total = sum(sizes)
    lo_r, hi_r = total * rank // world, total * (rank + 1) // world
    plan, at = [], 0
    for p, sz, reg in zip(paths, sizes, regular):
        f0, f1 = at, at + sz
        at = f1
        if not reg:
            # a FIFO / /dev/fd/N (process substitution): its size is unknown and it can be read once -- the whole input goes
            # to the rank whose share begins where it stands in the concatenation (the last rank when nothing follows it)
            if lo_r <= f0 < hi_r or (f0 >= total and rank == world - 1):
                plan.append(p)
            continue
        if sz == 0:
            continue  # an empty regular file: no records
        if not _splittable(p):
            if lo_r <= f0 < hi_r:  # whole file to the owner of its first byte
                plan.append(p)
            continue
        a, b = max(lo_r, f0), min(hi_r, f1)
        if a < b:
            plan.append((p, a - f0, b - f0))
    return plan


def _span_batch(L, h, path, max_bases, alloc, stats):
    """One batch through the one-pass reader (ntl_fastx_next_span / _parse_span / _copy_span): SeqSet, None at the end of the
    input, False when this span has to be read the two-pass way."""
    import ctypes as C
    t_0 = time.perf_counter()
    span, nw = C.c_uint64(), C.c_uint64()
    if L.ntl_fastx_next_span(h, max_bases, C.byref(span), C.byref(nw)) != 0:
        raise OSError(f"{path if not isinstance(path, tuple) else path[0]}: {L.ntl_fastx_error(h).decode()}")
    if span.value == 0:
        return None
    t_a = time.perf_counter()
    pinned = alloc(nw.value * 4)
    words = pinned.view(np.uint32)
    t_1 = time.perf_counter()
    n, nb, nn, nr = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = L.ntl_fastx_parse_span(h, words.ctypes.data, C.byref(n), C.byref(nb), C.byref(nn), C.byref(nr))
    if rc != 0:
        if hasattr(alloc, "__self__") and hasattr(alloc.__self__, "pinned_release"):
            alloc.__self__.pinned_release(pinned)
        if rc == -5:  # NTL_ERANGE
            return False
        raise OSError(f"{path}: {L.ntl_fastx_error(h).decode()}")
    n = n.value
    names = np.empty(nn.value, np.uint8)
    pos, lens, noff = np.empty(n, np.uint64), np.empty(n, np.uint32), np.empty(n + 1, np.uint64)
    srf, rst, rln = np.empty(n + 1, np.uint32), np.empty(nr.value, np.uint32), np.empty(nr.value, np.uint32)
    spanpos = C.c_uint64()
    if L.ntl_fastx_copy_span(h, pos.ctypes.data, lens.ctypes.data, names.ctypes.data, noff.ctypes.data, srf.ctypes.data, rst.ctypes.data,
                             rln.ctypes.data, C.byref(spanpos)) != 0:
        raise OSError(f"{path}: gather failed")
    off = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=off[1:])
    if stats is not None:
        t_2 = time.perf_counter()
        for key, dt in (("t_reader_count", t_a - t_0), ("t_reader_alloc", t_1 - t_a), ("t_reader_parse", t_2 - t_1)):
            stats[key] = stats.get(key, 0.0) + dt
        stats["reader_batches"] = stats.get("reader_batches", 0) + 1
        stats["one_pass_batches"] = stats.get("one_pass_batches", 0) + 1
    return SeqSet(Names(names, noff), None, off, packed=words, runs=(srf, rst, rln), pinned=pinned, positions=pos, span_positions=spanpos.value)


def load(paths, max_bases=None, alloc=None, ahead=None, stats=None, packed=False):
    """Native reader (ntl_fastx_*, csrc/ntl_io.cpp).  Whole input as one SeqSet, or, with max_bases,
    SeqSets of about that many bases; several files are concatenated in the order given and a batch
    never spans two files.  alloc(nbytes) -> uint8 array supplies the sequence buffers (the pair
    driver passes the device's page-locked pool); default numpy.  Opening a gzip file inflates it, so
    the next `ahead` files (default min(32, cores), at most 2 GiB of compressed input) are opened by
    background threads while the current one is consumed: many .fq.gz files decode in parallel.  An entry of `paths` may
    be (path, lo, hi): the records of a plain file that start in that byte range (shard_plan).  stats["parsed_bytes"]
    accumulates the input bytes consumed (file bytes of plain files and ranges, compressed bytes of gzip files).
    packed=True: the parser threads write 2-bit bases and the ACGT-run table instead of ASCII (SeqSet.packed; a quarter of
    the bytes for the device to fetch); a batch never mixes the two forms."""
    import collections
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    from . import capi
    L = capi.load()
    if isinstance(paths, str):
        paths = [paths]
    trace = bool(os.environ.get("NTL_IO_TRACE"))
    one_pass = packed and max_bases is not None and os.environ.get("NTL_IO_ONE_PASS", "1") != "0"
    n_ahead = ahead if ahead is not None else min(32, max(2, os.cpu_count() or 1))  # opening a gzip file inflates it: one thread per file
    whole = []
    pending = collections.deque()  # (path, future of an open handle, compressed bytes)
    todo = iter(paths)
    pool = ThreadPoolExecutor(max_workers=max(1, n_ahead))

    def top_up():
        while len(pending) < max(1, n_ahead) and (not pending or sum(p[2] for p in pending) < (2 << 30)):
            path = next(todo, None)
            if path is None:
                return
            try:
                size = os.path.getsize(path) if not isinstance(path, tuple) else int(path[2]) - int(path[1])
            except OSError:
                size = 0
            pending.append((path, pool.submit(_open_native, L, path), size))

    try:
        top_up()
        while pending:
            path, fut, _size = pending.popleft()
            h = fut.result()
            top_up()
            if stats is not None:
                lo, hi = C.c_uint64(), C.c_uint64()
                L.ntl_fastx_range(h, C.byref(lo), C.byref(hi))
                stats["parsed_bytes"] = stats.get("parsed_bytes", 0) + (hi.value - lo.value if hi.value else _size)
            try:
                while True:
                    if one_pass:
                        ss = _span_batch(L, h, path, int(max_bases), alloc or _np_empty, stats)
                        if ss is None:
                            break
                        if ss is not False:
                            yield ss
                            continue
                        # this span cannot be read in one pass (NTL_ERANGE): the two-pass reader below takes the same records
                    n = C.c_uint64()
                    t_0 = time.perf_counter()
                    if L.ntl_fastx_next(h, int(max_bases or 0), C.byref(n)) != 0:
                        raise OSError(f"{path if not isinstance(path, tuple) else path[0]}: {L.ntl_fastx_error(h).decode()}")
                    n = n.value
                    if n == 0:
                        break
                    t_1 = time.perf_counter()
                    nb, nn = C.c_uint64(), C.c_uint64()
                    L.ntl_fastx_sizes(h, None, C.byref(nb), C.byref(nn))
                    names = np.empty(nn.value, np.uint8)
                    off, noff = np.empty(n + 1, np.uint64), np.empty(n + 1, np.uint64)
                    if packed:
                        nw = int(L.ntl_packed_words(nb.value))
                        pinned = (alloc or _np_empty)(nw * 4)
                        words = pinned.view(np.uint32)
                        t_2 = time.perf_counter()
                        nr = C.c_uint64()
                        if L.ntl_fastx_copy_packed(h, words.ctypes.data, off.ctypes.data, names.ctypes.data, noff.ctypes.data, C.byref(nr)) != 0:
                            raise OSError(f"{path}: gather failed")
                        srf, rst, rln = np.empty(n + 1, np.uint32), np.empty(nr.value, np.uint32), np.empty(nr.value, np.uint32)
                        if L.ntl_fastx_runs(h, srf.ctypes.data, rst.ctypes.data, rln.ctypes.data) != 0:
                            raise OSError(f"{path}: run table failed")
                        buf = None
                    else:
                        buf = (alloc or _np_empty)(nb.value)
                        t_2 = time.perf_counter()
                        if L.ntl_fastx_copy(h, buf.ctypes.data, off.ctypes.data, names.ctypes.data, noff.ctypes.data) != 0:
                            raise OSError(f"{path}: gather failed")
                    if stats is not None:  # busy time of the reader thread, by phase (the rest of its wall time it waits for a free queue slot)
                        t_3 = time.perf_counter()
                        for key, dt in (("t_reader_count", t_1 - t_0), ("t_reader_alloc", t_2 - t_1), ("t_reader_parse", t_3 - t_2)):
                            stats[key] = stats.get(key, 0.0) + dt
                        stats["reader_batches"] = stats.get("reader_batches", 0) + 1
                    if trace:
                        print(f"ntl_fastx batch: read+count {t_1 - t_0:.4f}s alloc {t_2 - t_1:.4f}s parse {time.perf_counter() - t_2:.4f}s "
                              f"bases {nb.value} t={time.perf_counter():.4f}", file=sys.stderr)
                    ss = SeqSet(Names(names, noff), buf, off) if not packed else \
                        SeqSet(Names(names, noff), None, off, packed=words, runs=(srf, rst, rln), pinned=pinned)
                    if max_bases is None:
                        whole.append(ss)
                        break
                    yield ss
            finally:
                pool.submit(L.ntl_fastx_close, h)  # unmapping a multi-GB file takes ~0.1 s: not on the reader's path
    finally:
        for _path, fut, _size in pending:  # abandoned early: close what the background threads opened
            try:
                L.ntl_fastx_close(fut.result())
            except OSError:
                pass
        pool.shutdown(wait=True)
    if max_bases is None:
        yield concat(whole)


def load_parallel(paths, readers=None, chunk_bytes=None, max_bases=None, stats=None, **kw):
    """`load` with several readers at once.  The input is cut into chunks -- a file that cannot be cut (gzip stream, FIFO) is
    one chunk, plain and BGZF files are cut into byte ranges of about chunk_bytes (default: four batches' worth of text) --
    and `readers` threads each run `load` on one chunk at a time, the next unclaimed one; the batches come out in input
    order (a chunk's batches wait in its own queue until the chunks before it have been consumed, so at most `readers` chunks
    are held).  One reader spends its time in two passes per batch (count, then parse into place) that are bound by the kernel's
    per-page work on the page cache, not by the parser threads; two or three readers on different chunks overlap them.
    max_bases is required (whole-input loads have nothing to overlap)."""
    import itertools
    import queue
    import threading
    if isinstance(paths, str):
        paths = [paths]
    if readers is None:
        readers = int(os.environ.get("NTL_IO_READERS", "0")) or (2 if (os.cpu_count() or 1) >= 32 else 1)  # measured on the 256-core GPU host: 1 -> 2 readers +7 %, more only move the wait to the writers
    if max_bases is None or readers <= 1:
        yield from load(paths, max_bases=max_bases, stats=stats, **kw)
        return
    if chunk_bytes is None:
        chunk_bytes = int(os.environ.get("NTL_IO_CHUNK_BYTES", "0")) or max(4 * int(max_bases), 64 << 20)
    chunks = []
    for p in paths:
        path, lo, hi = p if isinstance(p, tuple) else (p, 0, None)
        if not _splittable(path):
            # consecutive files that cannot be cut (gzip streams) form ONE chunk: `load` opens -- inflates -- the next files of its
            # list ahead on a pool of threads, which a reader that is handed one file at a time cannot do (16 .fq.gz files on the
            # GPU box: 1.3 Gbases/s one file per reader, 4.0 as one list; profiles/r03ah_gz_diag.txt)
            if chunks and isinstance(chunks[-1], list):
                chunks[-1].append(p)
            else:
                chunks.append([p])
            continue
        hi = os.path.getsize(path) if hi is None else int(hi)
        n = max(1, -(-(hi - int(lo)) // chunk_bytes))
        if n == 1 and not isinstance(p, tuple):
            chunks.append(p)
            continue
        for i in range(n):
            a, b = int(lo) + (hi - int(lo)) * i // n, int(lo) + (hi - int(lo)) * (i + 1) // n
            if a < b:
                chunks.append((path, a, b))
    chunks = [c[0] if isinstance(c, list) and len(c) == 1 else c for c in chunks]
    if len(chunks) <= 1:
        yield from load(paths, max_bases=max_bases, stats=stats, **kw)
        return
    END = object()
    # bounded: a chunk that cannot be cut (a list of gzip files) may hold any number of batches, and every batch is a
    # page-locked buffer -- a reader parks (stop-aware) once `depth` of its batches wait for the consumer
    depth = max(2, int(os.environ.get("NTL_IO_QUEUE_DEPTH", "4")))
    qs = [queue.Queue(maxsize=depth) for _ in chunks]

    def put(q, item):
        while True:
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                if stop.is_set():
                    return False
    claim = itertools.count()
    lock, stop = threading.Lock(), threading.Event()
    go = threading.Semaphore(readers)  # a reader starts a new chunk only when fewer than `readers` chunks are unconsumed
    part_stats = []

    def work():
        while not stop.is_set():
            go.acquire()
            if stop.is_set():
                return
            with lock:
                i = next(claim)
            if i >= len(chunks):
                return
            st = {}
            try:
                for ss in load(chunks[i] if isinstance(chunks[i], list) else [chunks[i]], max_bases=max_bases, stats=st, **kw):
                    if not put(qs[i], ss) or stop.is_set():
                        break
                put(qs[i], END)
            except BaseException as exc:  # re-raised by the consumer at this chunk's place
                put(qs[i], exc)
            with lock:
                part_stats.append(st)

    threads = [threading.Thread(target=work, daemon=True) for _ in range(min(readers, len(chunks)))]
    for t in threads:
        t.start()
    try:
        for i in range(len(chunks)):
            while True:
                item = qs[i].get()
                if item is END:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
            go.release()
    finally:
        stop.set()
        for _ in threads:
            go.release()
        for t in threads:
            t.join()
        if stats is not None:
            for st in part_stats:
                for key, v in st.items():
                    stats[key] = stats.get(key, 0) + v
            stats["readers"] = len(threads)


def concat(sets):
    if len(sets) == 1:
        return sets[0]
    if any(s.packed is not None for s in sets):
        raise ValueError("packed sequence sets cannot be concatenated: read the files one at a time")
    if not sets:
        return SeqSet(Names(np.zeros(0, np.uint8), np.zeros(1, np.uint64)), np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    bufs = [s.buf for s in sets]
    offs, noffs, b0, n0 = [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)], 0, 0
    for s in sets:
        offs.append(s.offsets[1:] + np.uint64(b0))
        noffs.append(s.names.off[1:] + np.uint64(n0))
        b0 += int(s.offsets[-1])
        n0 += int(s.names.off[-1])
    return SeqSet(Names(np.concatenate([s.names.blob for s in sets]), np.concatenate(noffs)), np.concatenate(bufs), np.concatenate(offs))


def load_all(paths, alloc=None, packed=False):
    return next(load(paths, alloc=alloc, packed=packed))
