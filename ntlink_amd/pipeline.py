"""The pair stage on MI355X: drivers that mirror the reference's two operators and the fused path.

  run_indexlr(...)      = `indexlr --long --pos --strand [--len] -k K -w W`        (ntLink:199,223)
  run_ntlink_pair(...)  = `ntlink_pair.py -p P -n N -m contigs.tsv -s target.fa -k K -a A -z Z -f F
                           -x X [--verbose --pairs --sensitive --repeat-filter --paf] FILES`
                                                                    (bin/ntlink_pair.py:509-613)
  run_pair(...)         = `ntLink pair target= reads= ...` without the text TSV between the two
                          (ntLink:165,198-199,221-225); still leaves <target>.kK.wW.tsv behind.

Compute is on the GPU through ntlink_amd.capi (C ABI -> HIP kernels); this module only moves
records between files and the device and runs the small CPU tail (pairing.py).
"""
import datetime
import json
import os
import sys
import time

import numpy as np

from . import capi, formats, pairing, seqio

TSV_BLOCK_BYTES = 256 << 20  # text of the read-minimizer TSV parsed per device batch (operator B2)
DEFAULT_BATCH_BASES = 512_000_000  # read bases per device batch (packed: 128 MB); the next batches are parsed meanwhile (256 M: 7 % slower file to file)


_TRACE = [] if os.environ.get("NTL_PIPE_TRACE") else None  # (seconds, thread, what, batch): a timeline of the pair driver, dumped at the end


def _mark(what, seq=-1):
    if _TRACE is not None:
        import threading
        _TRACE.append((time.perf_counter(), threading.current_thread().name, what, seq))


def _log(*a):
    print(datetime.datetime.today(), ":", *a, file=sys.stdout, flush=True)


class PairOutputs:
    """<prefix>.verbose_mapping.tsv / .paf writers + the pair tally, fed batch by batch in read order."""

    def __init__(self, prefix, ctg_names, ctg_len, k, f, verbose, paf, part=""):
        """part: suffix of this rank's part files in a multi-process run (the final names appear by a rename at the end)"""
        self.prefix, self.part = prefix, part
        self.ctg_names, self.ctg_len = ctg_names, ctg_len
        self.verbose_fh = open(prefix + ".verbose_mapping.tsv" + part, "w") if verbose else None
        self.paf_fh = open(prefix + ".paf" + part, "w") if paf else None
        self.tally = pairing.PairTally(ctg_names, ctg_len, k, f)
        self.t_write = self.t_tally = 0.0
        self._exc = None

    def add(self, res, read_names, read_len):
        t0 = time.perf_counter()
        if "verbose" in res:  # the lines were made on the device (capi.MapResult.format): they are written as they are
            paf_job = None
            if self.paf_fh and len(res["paf"]):
                import threading
                paf_job = threading.Thread(target=self._guard, args=(formats.write_blob, self.paf_fh, res["paf"]))
                paf_job.start()
            try:
                if self.verbose_fh:
                    formats.write_blob(self.verbose_fh, res["verbose"])
                t1 = time.perf_counter()
                self.tally.add_batch_ends(res["maps"], res["ends"], read_len)
            finally:
                if paf_job:
                    paf_job.join()
            if self._exc is not None:
                exc, self._exc = self._exc, None
                raise exc
            self.t_write += t1 - t0
            self.t_tally += time.perf_counter() - t1
            return
        paf_job = None
        if self.paf_fh:  # the two files are independent: the PAF is written next to the verbose mapping
            import threading
            paf_job = threading.Thread(target=self._guard, args=(formats.write_paf, self.paf_fh, res, read_names, read_len,
                                                                  self.ctg_names, self.ctg_len))
            paf_job.start()
        try:
            if self.verbose_fh:
                formats.write_verbose(self.verbose_fh, res, read_names, self.ctg_names)
            t1 = time.perf_counter()
            self.tally.add_batch(res, read_len)
        finally:
            if paf_job:
                paf_job.join()
        if self._exc is not None:
            exc, self._exc = self._exc, None
            raise exc
        self.t_write += t1 - t0
        self.t_tally += time.perf_counter() - t1

    def _guard(self, fn, *args):
        try:
            fn(*args)
        except BaseException as exc:
            self._exc = exc

    def close(self):
        for fh in (self.verbose_fh, self.paf_fh):
            if fh:
                fh.close()

    def remove_partial(self):
        """Error convention of the reference: partial outputs are deleted (bin/ntlink_pair.py:608-613)."""
        self.close()
        for ext, on in ((".verbose_mapping.tsv", self.verbose_fh), (".paf", self.paf_fh)):
            for path in {self.prefix + ext + self.part, self.prefix + ext, self.prefix + ext + ".assembling"}:
                if on and os.path.exists(path):
                    try:
                        os.remove(path)
                    except FileNotFoundError:  # another rank was faster
                        pass


def finish_pairs(tally, prefix, n, a, write_pairs_tsv):
    """filter_pairs_distances, filter_weak_anchor_pairs, write_pairs, the scaffold graph (bin/ntlink_pair.py:594-606): the native
    tally writes both files (NTL_NATIVE_PAIRS=0: the Python forms in pairing.py, same bytes)."""
    dot = f"{prefix}.n{n}.scaffold.dot"
    if os.environ.get("NTL_NATIVE_PAIRS", "1") != "0":
        _log("Printing graph", dot)
        return tally.write(a, int(n), prefix + ".pairs.tsv" if write_pairs_tsv else None, dot)
    pairs = tally.filtered(a)
    if write_pairs_tsv:
        with open(prefix + ".pairs.tsv", "w") as fh:
            pairing.write_pairs(fh, pairs)
    _log("Printing graph", dot)
    with open(dot, "w") as fh:
        pairing.write_dot(fh, pairs, tally.names, tally.ctg_len, int(n))
    return len(pairs)


def run_indexlr(dev, paths, k, w, out, with_len, batch_bases=DEFAULT_BATCH_BASES, with_strand=True):
    """Sketch FASTA/FASTQ files on the device and print indexlr's TSV.  Three stages on threads: the reader
    parses the next batch into page-locked memory, the device sketches the current one, the emitter
    formats and writes the previous one."""
    batches = Prefetch(seqio.load(paths, max_bases=batch_bases, alloc=dev.pinned_empty))
    drain = Drain(lambda names, lens, off, h, p, s: formats.write_indexlr(out, names, lens, off, h, p, s, with_len, with_strand))
    try:
        for ss in batches:
            if not len(ss):
                continue
            with dev.batch(ss.buf, ss.offsets) as b:
                dev.pinned_release(ss.buf)
                ss.buf = None
                with dev.sketch(b, k, w) as sk:
                    off, h, p, s = sk.download()
            drain.put(ss.names, ss.lengths, off, h, p, s)
        drain.close()
    except BaseException:
        batches.stop()
        try:
            drain.close()
        except BaseException:
            pass
        raise


def _contig_lengths(fasta):
    """ids, lengths and id -> record index of the target FASTA.  The reference keeps ONE Scaffold per id, the last
    record's (dict overwrite, bin/ntlink_utils.py:65-73): every record of a repeated id gets that length (it feeds the
    z filter, the gap estimates and the PAF target length)."""
    ss = seqio.load_all([fasta])
    names, ctg_len = ss.names.tolist(), ss.lengths
    index_of = {n: i for i, n in enumerate(names)}
    if len(index_of) != len(names):
        last = ctg_len.copy()
        for i, n in enumerate(names):
            ctg_len[i] = last[index_of[n]]
    return names, ctg_len, index_of


def run_ntlink_pair(dev, args):
    """args: the namespace of the reference's argparse (FILES s m p n k z a f x checkpoint pairs paf
    sensitive repeat_filter verbose)."""
    ckpt = args.checkpoint
    if os.path.isfile(args.p + ".verbose_mapping.tsv"):
        ckpt = args.p + ".verbose_mapping.tsv"  # bin/ntlink_pair.py:565-567
    _log("Reading fasta file", args.s)
    names, ctg_len, index_of = _contig_lengths(args.s)
    if ckpt:
        print("Found checkpoint file, bypassing read mapping...\n")
        if args.paf:
            print("Warning: --paf specified, but not compatible with checkpoint")
        tally = pairing.PairTally(names, ctg_len, args.k, args.f)
        _log("Finding pairs")
        with open(ckpt) as fh:
            for _read, entries in formats.parse_verbose(fh):
                tally.add_checkpoint_read(entries, index_of)
        finish_pairs(tally, args.p, args.n, args.a, args.pairs)
        _log("DONE!")
        return
    _log("Reading minimizers", args.s)
    blocks = list(formats.read_indexlr(args.m, False))  # native parser; "-" = stdin
    if blocks:
        cn, _, coff, ch, cp, cs = blocks[0]
    else:
        cn, coff, ch, cp, cs = [], np.zeros(1, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint32), np.zeros(0, np.uint8)
    # contig-id order = order of the FASTA; TSV lines may be any subset in any order
    nctg = len(names)
    ids = np.array([index_of[n] for n in cn], np.int64)
    cid = np.repeat(ids, np.diff(coff).astype(np.int64)) if len(ids) else np.zeros(0, np.int64)
    order = np.argsort(cid, kind="stable")
    cnt = np.bincount(cid, minlength=nctg).astype(np.uint64)
    full_off = np.zeros(nctg + 1, np.uint64)
    np.cumsum(cnt, out=full_off[1:])
    out = PairOutputs(args.p, names, ctg_len, args.k, args.f, args.verbose, args.paf)
    try:
        with dev.sketch_from_arrays(full_off, ch[order], cp[order], cs[order]) as csk, dev.index(csk, ctg_len) as ix:
            _log("Finding pairs")
            for path in args.FILES:
                # blocks of whole lines, parsed by several threads while the previous block is on the device
                for rn, rlen, roff, rh, rp, rs in prefetched(formats.read_indexlr(path, True, TSV_BLOCK_BYTES), depth=1):
                    with dev.sketch_from_arrays(roff, rh, rp, rs) as rsk, \
                            dev.map(ix, rsk, rlen, k=args.k, z=args.z, x=args.x, sensitive=args.sensitive,
                                    repeat_filter=args.repeat_filter) as res:
                        out.add(res.download(), rn, rlen)
        out.close()
        finish_pairs(out.tally, args.p, args.n, args.a, args.pairs)
        _log("DONE!")
    except BaseException:
        out.remove_partial()
        raise


class Prefetch:
    """Runs a generator in a background thread from the moment it is created (the native reader
    releases the GIL), so that parsing the next read batch overlaps device work and output writing of
    the current one -- and the first batch overlaps the contig stage."""

    def __init__(self, gen, depth=2):
        import queue
        import threading
        self._q, self._end, self._stop = queue.Queue(maxsize=depth), object(), False

        def run():
            try:
                for item in gen:
                    _mark("reader_out")
                    self._q.put(item)
                    _mark("reader_put_done")
                    if self._stop:
                        gen.close()  # closes the readers the generator holds
                        return
                self._q.put(self._end)
            except BaseException as exc:  # re-raised in the consumer
                self._q.put(exc)

        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def stop(self):
        """Abandon the stream (error path): lets the producer finish its current item and exit, so that
        nothing is parsing into device-owned buffers when the device goes away."""
        import queue
        self._stop = True
        while self._t.is_alive():
            try:
                self._q.get(timeout=0.05)
            except queue.Empty:
                pass

    def __iter__(self):
        while True:
            item = self._q.get()
            if item is self._end:
                return
            if isinstance(item, BaseException):
                raise item
            yield item


def prefetched(gen, depth=2):
    return iter(Prefetch(gen, depth))


class Drain:
    """Ordered background consumer: `put(*args)` hands one batch to `fn` on a worker thread (the native
    writers release the GIL), `close()` waits for it and re-raises what it raised."""

    def __init__(self, fn, depth=2):
        import queue
        import threading
        self._q, self._exc, self._end = queue.Queue(maxsize=depth), None, object()

        def run():
            while True:
                item = self._q.get()
                if item is self._end:
                    return
                if self._exc is None:
                    try:
                        fn(*item)
                    except BaseException as exc:  # keep draining so that put() never blocks forever
                        self._exc = exc

        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def put(self, *args):
        if self._exc is not None:
            self.close()
        self._q.put(args)

    def close(self):
        if self._t is not None:
            self._q.put(self._end)
            self._t.join()
            self._t = None
        if self._exc is not None:
            exc, self._exc = self._exc, None
            raise exc


class LocalComm:
    """Single process.  The multi-GPU launcher passes a torch.distributed-backed object with the same
    members (ntlink_amd/dist_pair.py)."""
    rank, world = 0, 1

    def gather(self, obj):
        return [obj]

    def allgather(self, obj):
        return [obj]

    def barrier(self):
        pass


def _node_id():
    """What two ranks must share to be on one node: the host name AND the kernel's boot id (containers on different machines may carry
    the same host name) AND the device and inode of the directory the shared file would live in (two containers of one machine have
    separate /dev/shm mounts)."""
    import socket
    name = os.environ.get("NTL_FAKE_HOSTNAME") or socket.gethostname()
    if os.environ.get("NTL_FAKE_HOSTNAME"):
        return name
    try:
        with open("/proc/sys/kernel/random/boot_id") as fh:
            boot = fh.read().strip()
    except OSError:
        boot = ""
    try:
        st = os.stat(_shm_dir())
        where = f"{st.st_dev}:{st.st_ino}"
    except OSError:
        where = ""
    return f"{name}/{boot}/{where}"


def _shm_dir():
    return os.environ.get("NTL_SHM_DIR") or ("/dev/shm" if os.path.isdir("/dev/shm") else (os.environ.get("TMPDIR") or "/tmp"))


def _node_leader(comm):
    """(lowest rank on this rank's node, ranks on this node): the ranks of one node share what only one of them has to make."""
    hosts = comm.allgather(_node_id())
    mine = [r for r, h in enumerate(hosts) if h == hosts[comm.rank]]
    return mine[0], mine


_shared_files = set()  # published and not yet unlinked: removed at exit if this process dies between the two collectives


def _cleanup_shared():
    for path in list(_shared_files):
        try:
            os.unlink(path)
        except OSError:
            pass
        _shared_files.discard(path)


def _publish(arrays):
    """The packed contig set into one file of the node's shared-memory directory: a name nobody can guess, created exclusively, not
    through a link, readable by this user only.  Raises OSError when there is no room (checked before a byte is written: a tmpfs
    that fills up takes the machine's memory with it) or the directory cannot be written."""
    import atexit
    import json
    import secrets
    shm = _shm_dir()
    head, at = {}, 4096
    for key in _SHARED_FIELDS:
        a = np.ascontiguousarray(arrays[key])
        head[key] = (str(a.dtype), int(a.size), at)
        at += (a.nbytes + 63) & ~63
    hb = json.dumps(head).encode()
    assert len(hb) + 8 <= 4096
    vfs = os.statvfs(shm)
    if vfs.f_bavail * vfs.f_frsize < at + (64 << 20):
        raise OSError(f"{shm}: {vfs.f_bavail * vfs.f_frsize >> 20} MB free, the packed target takes {at >> 20} MB")
    path = os.path.join(shm, f"ntlink_amd.ctg.{os.getuid()}.{secrets.token_hex(12)}")
    fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_WRONLY | getattr(os, "O_NOFOLLOW", 0), 0o600)
    if not _shared_files:
        atexit.register(_cleanup_shared)
    _shared_files.add(path)
    try:
        with os.fdopen(fd, "wb") as fh:
            fh.write(len(hb).to_bytes(8, "little") + hb)
            for key in _SHARED_FIELDS:
                fh.seek(head[key][2])
                np.ascontiguousarray(arrays[key]).tofile(fh)
            fh.truncate(max(at, 4096))
    except BaseException:
        try:
            os.unlink(path)
        except OSError:
            pass
        _shared_files.discard(path)
        raise
    return path


_SHARED_FIELDS = ("names_blob", "names_off", "offsets", "packed", "seq_run_first", "run_start", "run_len")


def shared_contigs(comm, target, alloc, packed, stats=None):
    """The target FASTA parsed and packed ONCE PER NODE (round 5; BASELINE configs[3]: eight ranks on one host each parsed the same 3-GB
    file -- eight times 0.2 s of all parser threads, on a CPU quota that eight ranks share).  The lowest rank of a host parses as a
    single process would (seqio.load_all), writes the arrays of the packed set -- names, offsets, 2-bit words, ACGT-run table -- into one
    file under /dev/shm and tells the others its name; they map it read-only and upload from the mapping.  The file is unlinked as
    soon as every rank of the host has it open.  Every GPU still sketches and indexes the contigs itself (6 ms of kernels: a peer
    copy of the finished table would save less than its synchronisation costs).  NTL_SHARE_CONTIGS=0, one rank per host, or an
    unpacked load: every rank parses."""
    import json
    import mmap
    import numpy as np
    share = comm.world > 1 and packed and os.environ.get("NTL_SHARE_CONTIGS", "1") != "0"
    leader, peers = _node_leader(comm) if share else (comm.rank, [comm.rank])
    if not share or len(peers) == 1:
        if stats is not None:
            stats["contigs_parsed_by"] = comm.rank
        if share:  # the collectives below are the whole world's: keep in step with the hosts that do share
            comm.allgather(None)
            comm.allgather(None)
        return seqio.load_all([target], alloc=alloc if packed else None, packed=packed), None
    path, ss, failure = None, None, None
    if comm.rank == leader:
        # Parsing and publishing fail differently (round 6): a target that does not parse fails on every rank, as it did when
        # every rank parsed; a copy that cannot be PUBLISHED (no room in /dev/shm -- Docker's default is 64 MB --, a read-only or
        # quota'd tmpfs) only costs the sharing: the other ranks of the host are told to parse for themselves.
        try:
            ss = seqio.load_all([target], alloc=alloc, packed=True)
        except BaseException as exc:
            # the others wait in the collective below: they must hear of it, not hang
            failure, path = exc, ("failed", f"{type(exc).__name__}: {exc}")
        if failure is None:
            try:
                path = _publish(dict(names_blob=ss.names.blob, names_off=ss.names.off, offsets=np.ascontiguousarray(ss.offsets, np.uint64),
                                     packed=ss.packed, seq_run_first=ss.seq_run_first, run_start=ss.run_start, run_len=ss.run_len))
            except (OSError, AssertionError) as exc:
                path = ("noshare", f"{type(exc).__name__}: {exc}")
    paths = comm.allgather(path)  # (also: the file is complete)
    if failure is not None:
        raise failure
    mine = paths[leader]
    if isinstance(mine, tuple) and mine[0] == "failed":
        raise OSError(f"{target}: the rank that parses it for this host (rank {leader}) failed: {mine[1]}")
    mapping, parsed_by = None, leader
    if comm.rank != leader:
        if isinstance(mine, str):
            try:
                with open(mine, "rb") as fh:
                    mapping = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
                n = int.from_bytes(mapping[:8], "little")
                head = json.loads(bytes(mapping[8:8 + n]))
                arr = {key: np.frombuffer(mapping, dtype=np.dtype(dt), count=cnt, offset=off) for key, (dt, cnt, off) in head.items()}
                arr = {key: (a if key == "packed" else np.array(a)) for key, a in arr.items()}  # only the 2-bit words stay a view of the mapping
                ss = seqio.SeqSet(seqio.Names(arr["names_blob"], arr["names_off"]), None, arr["offsets"], packed=arr["packed"],
                                  runs=(arr["seq_run_first"], arr["run_start"], arr["run_len"]))
            except (OSError, ValueError, KeyError) as exc:  # a file this rank cannot see after all (another mount): parse here
                mapping, ss = None, None
                if stats is not None:
                    stats["contigs_share_failed"] = f"{type(exc).__name__}: {exc}"
        elif stats is not None:
            stats["contigs_share_failed"] = mine[1]
    elif isinstance(mine, tuple) and stats is not None:
        stats["contigs_share_failed"] = mine[1]
    comm.allgather(None)  # every rank of every host has its leader's file open (or has given it up)
    if comm.rank == leader and isinstance(path, str):
        try:
            os.unlink(path)
        except OSError:
            pass
        _shared_files.discard(path)
    if ss is None:  # nothing was shared with this rank: parse as a single process would
        ss = seqio.load_all([target], alloc=alloc, packed=True)
        parsed_by = comm.rank
    if stats is not None:
        stats["contigs_parsed_by"] = parsed_by
    return ss, mapping


def shard_range(offsets, rank, world):
    """Contiguous read range [lo, hi) of this rank, balanced by bases; concatenating the ranks' ranges
    in rank order restores the input order.  (Sharding of in-memory batches: the bench and callers that
    already hold the reads.  The file-to-file driver shards the INPUT BYTES instead: seqio.shard_plan.)"""
    n = len(offsets) - 1
    if world == 1:
        return 0, n
    total = int(offsets[-1]) - int(offsets[0])
    cuts = [int(np.searchsorted(offsets, int(offsets[0]) + total * r // world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, world + 1):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def _append_part(final_path, part_path, offset):
    """This rank's part file into its place in the final file (in-kernel copy; ranks write disjoint ranges at once)."""
    size = os.path.getsize(part_path)
    src = os.open(part_path, os.O_RDONLY)
    dst = os.open(final_path, os.O_WRONLY | os.O_CREAT, 0o644)
    try:
        done = 0
        while done < size:
            try:
                n = os.copy_file_range(src, dst, min(size - done, 1 << 30), done, offset + done)
            except OSError:
                n = 0
            if n <= 0:  # file systems without copy_file_range
                chunk = os.pread(src, min(size - done, 64 << 20), done)
                if not chunk:
                    raise OSError(f"short read from {part_path}")
                n = os.pwrite(dst, chunk, offset + done)
            done += n
    finally:
        os.close(src)
        os.close(dst)
    os.remove(part_path)


def _map_batches(dev, ix, batches, drain, stats, t_mark, w, first=None, text=None, **map_kw):
    """The device stage of the pair driver: read batches -> records, handed to `drain` in input order.

    NTL_DEVICE_STREAMS (default 2) worker threads, each with its own context on the GPU (stream, block cache, page-locked
    result buffers), take batches in turn: while one batch's kernels and record download run, the next batch's bases
    cross PCIe.  The batches' results are committed in input order.  `first` (if given) runs on the calling thread, on `dev`,
    once the other workers have started: work of the contig stage that the mapping does not depend on."""
    import threading
    n_workers = max(1, int(os.environ.get("NTL_DEVICE_STREAMS", "2")))
    devs = [dev] + dev.workers(n_workers)  # kept by `dev` from one call to the next
    it = iter(batches)
    lock, commit = threading.Lock(), threading.Condition()
    state = {"next_seq": 0, "commit_seq": 0, "error": None, "t_mark": t_mark}

    def take():
        with lock:
            while True:
                if state["error"] is not None:
                    return None
                rs_ = next(it, None)
                if rs_ is None:
                    return None
                if len(rs_):
                    seq = state["next_seq"]
                    state["next_seq"] += 1
                    now = time.perf_counter()
                    stats["t_ingest"] += max(0.0, now - state["t_mark"])  # the device stage waited for the reader thread
                    state["t_mark"] = now
                    return seq, rs_

    def work(wdev):
        try:
            while True:
                got = take()
                if got is None:
                    return
                seq, rs_ = got
                _mark("dev_start", seq)
                t_dev = time.perf_counter()
                rl = rs_.lengths
                with (wdev.batch_packed(rs_) if rs_.packed is not None else wdev.batch(rs_.buf, rs_.offsets)) as rb:
                    t_up = time.perf_counter()
                    dev.pinned_release(rs_.pinned if rs_.packed is not None else rs_.buf)  # on the device: the reader may refill it
                    rs_.buf = rs_.packed = rs_.pinned = None
                    with wdev.sketch(rb, map_kw["k"], w, index=ix, records=False) as rsk:  # made for this map only: no records
                        t_sk = time.perf_counter()
                        with wdev.map(ix, rsk, rl, **map_kw) as res:
                            t_mp = time.perf_counter()
                            pres = None
                            if text is not None:  # (contig name table, verbose, paf): the lines are formatted on the device
                                try:
                                    with wdev.names(rs_.names, rl) as rn, res.format(rn, text[0], text[1], text[2]) as txt:
                                        pres = txt.download()
                                except capi.NtlError as exc:
                                    if getattr(exc, "code", 0) != capi.NTL_ERANGE:  # more than 4 GB of text in one batch: the host's emitters take it
                                        raise
                            if pres is None:
                                pres = res.download(pinned=True)
                            n_mx, n_hit = rsk.count, res.n_index_hits
                t_done = time.perf_counter()
                _mark("dev_done", seq)
                parts = (t_up - t_dev, t_sk - t_up, t_mp - t_sk, t_done - t_mp)
                with commit:
                    while state["commit_seq"] != seq and state["error"] is None:
                        commit.wait(0.05)
                    if state["error"] is not None:
                        return
                    t_put = time.perf_counter()
                    _mark("commit", seq)
                    drain.put(pres, rs_.names, rl)
                    _mark("handed_over", seq)
                    now = time.perf_counter()
                    stats["t_device"] += t_done - t_dev      # H2D + pack + kernels + D2H, summed over the worker threads
                    for key, v in zip(("upload_pack", "sketch", "map", "download_free"), parts):
                        stats["t_device_parts"][key] = round(stats["t_device_parts"].get(key, 0.0) + v, 4)
                    stats["t_handover"] += now - t_put       # waiting for the writer thread to take the batch
                    stats["read_minimizers"] += n_mx
                    stats["index_hits"] += n_hit
                    stats["read_bases"] += rs_.bases
                    stats["reads"] += len(rs_)
                    state["commit_seq"] += 1
                    state["t_mark"] = max(state["t_mark"], now)
                    commit.notify_all()
        except BaseException as exc:
            with commit:
                if state["error"] is None:
                    state["error"] = exc
                commit.notify_all()

    threads = [threading.Thread(target=work, args=(d,), daemon=True) for d in devs[1:]]
    for t in threads:
        t.start()
    if first is not None:
        try:
            first()
        except BaseException as exc:
            with commit:
                if state["error"] is None:
                    state["error"] = exc
                commit.notify_all()
    work(devs[0])
    for t in threads:
        t.join()
    # (Results of the worker contexts may still sit in the writer's queue as views into their page-locked buffers: the contexts
    # live as long as `dev` does.)
    if state["error"] is not None:
        raise state["error"]


def run_pair(dev, target, reads, prefix=None, k=32, w=100, n=1, a=1, z=1000, f=10, x=0.0, paf=False, verbose=True,
             sensitive=False, repeats=False, pairs_tsv=False, batch_bases=DEFAULT_BATCH_BASES, write_contig_tsv=True,
             comm=None):
    """`ntLink pair target=T reads='R1 R2' k= w= ...`: the fused device path.

    With a communicator of world > 1 (one process per GPU) every rank builds the same contig index on its own GPU and
    owns a contiguous BYTE RANGE of the concatenated read files (seqio.shard_plan): it parses only that, maps it, formats
    its own text into part files and keeps its own pair tally.  At the end the ranks exchange three numbers (the byte
    counts of their parts) and copy their parts, all at once, to their offsets in the final files; rank 0 merges the
    pair tallies in rank order -- which is read order -- and writes `.pairs.tsv` / `.scaffold.dot`.  No record array
    leaves the process that produced it and no collective touches the data path on the device."""
    comm = comm or LocalComm()
    root = comm.rank == 0
    prefix = prefix or f"{target}.k{k}.w{w}.z{z}"
    if os.path.isfile(prefix + ".verbose_mapping.tsv"):
        # same silent switch as the reference
        if root:
            import argparse
            run_ntlink_pair(dev, argparse.Namespace(FILES=[], s=target, m=None, p=prefix, n=n, k=k, z=z, a=a, f=f, x=x,
                                                    checkpoint=None, pairs=pairs_tsv, paf=paf, sensitive=sensitive,
                                                    repeat_filter=repeats, verbose=verbose))
        comm.barrier()
        return None
    if comm.world > 1 and root:
        # leftovers of a run that died: part files and half-assembled outputs must not be taken for this run's
        import glob
        for ext in (".verbose_mapping.tsv", ".paf"):
            for stale in glob.glob(glob.escape(prefix + ext) + ".part*") + glob.glob(glob.escape(prefix + ext) + ".assembling"):
                os.remove(stale)
    comm.barrier()  # nobody creates the verbose file before everyone has looked for it
    t_start = time.perf_counter()
    read_paths = reads.split() if isinstance(reads, str) else list(reads)
    plan = seqio.shard_plan(read_paths, comm.rank, comm.world)
    io_stats = {}
    # this rank's share of the read files is opened, inflated and parsed from now on, behind the contig stage
    packed = os.environ.get("NTL_HOST_PACK", "1") != "0"  # the parser threads pack to 2 bits per base: a quarter of the PCIe bytes
    batches = Prefetch(seqio.load_parallel(plan, max_bases=batch_bases, alloc=dev.pinned_empty, stats=io_stats, packed=packed))
    # the device workers' contexts (streams, tables, a first slab of device memory each) are made while the contig file is parsed
    import threading
    make_workers = threading.Thread(target=dev.workers, args=(max(1, int(os.environ.get("NTL_DEVICE_STREAMS", "2"))),), daemon=True)
    make_workers.start()
    # into a page-locked buffer: registering one costs 45 us per MB (csrc/ntl_hip.hip, pin_alloc), the staged copy of a pageable one 130
    ctg_share = {}
    ctg, ctg_mapping = shared_contigs(comm, target, dev.pinned_empty, packed, ctg_share)  # (several ranks on a host: one of them parses)
    ctg_len = ctg.lengths
    make_workers.join()
    t_ctg_parsed = time.perf_counter()
    # Several ranks: EVERY rank (0 too) writes part files; the final names appear only by a rename after all parts are in
    # place -- a run that dies half way leaves no well-formed <prefix>.verbose_mapping.tsv with a fraction of the reads,
    # which the next run (this driver and the reference alike, bin/ntlink_pair.py:565-567) would take for a checkpoint.
    part = "" if comm.world == 1 else f".part{comm.rank}"
    out = PairOutputs(prefix, ctg.names, ctg_len, k, f, verbose, paf, part=part)

    def consume(pres, names, lens):
        _mark("write_start")
        out.add(pres, names, lens)
        _mark("write_done")
        pres.get("_owner", dev).pinned_release(pres.get("_pinned"))

    def emit_contig_tsv(off, h, p, s):
        with open(f"{target}.k{k}.w{w}.tsv", "w") as fh:
            formats.write_indexlr(fh, ctg.names, ctg_len, off, h, p, s, False)

    drain = Drain(consume)  # text emitters + pair tally run behind the device
    tsv_drain = Drain(emit_contig_tsv) if root and write_contig_tsv else None
    stats = dict(read_bases=0, reads=0, read_minimizers=0, index_hits=0, t_contigs=0.0, t_ingest=0.0, t_device=0.0, t_handover=0.0, t_device_parts={})
    try:
        with (dev.batch_packed(ctg) if ctg.packed is not None else dev.batch(ctg.buf, ctg.offsets)) as cb:
            dev.pinned_release(ctg.pinned)
            ctg.buf = ctg.packed = ctg.pinned = None
            if ctg_mapping is not None:  # the node's shared copy of the packed contigs: uploaded, not needed any more
                try:
                    ctg_mapping.close()
                except BufferError:  # (a view still alive somewhere: the mapping goes with it)
                    pass
                ctg_mapping = None
            t_ctg_up = time.perf_counter()
            with dev.sketch(cb, k, w) as csk:
                t_ctg_sk = time.perf_counter()
                with dev.index(csk, ctg_len) as ix:
                    stats["index_size"] = len(ix)
                    t_mark = time.perf_counter()
                    stats["t_contigs"] = t_mark - t_start
                    stats["t_contigs_parts"] = {"parse": round(t_ctg_parsed - t_start, 4), "upload_pack": round(t_ctg_up - t_ctg_parsed, 4),
                                                "sketch": round(t_ctg_sk - t_ctg_up, 4), "index": round(t_mark - t_ctg_sk, 4)}

                    def contig_tsv():
                        # <target>.k<k>.w<w>.tsv: its records come off the device while the other workers already map reads,
                        # and are written while the reads are mapped
                        t0 = time.perf_counter()
                        tsv_drain.put(*csk.download())
                        stats["t_contigs_parts"]["download_for_tsv_beside_mapping"] = round(time.perf_counter() - t0, 4)
                    # NTL_DEVICE_TEXT=0: records cross PCIe and the host's emitter threads format them (ntl_write_verbose / _paf)
                    ctg_tab = dev.names(ctg.names, ctg_len) if os.environ.get("NTL_DEVICE_TEXT", "1") != "0" else None
                    try:
                        _map_batches(dev, ix, batches, drain, stats, t_mark, w, first=contig_tsv if tsv_drain else None,
                                     text=(ctg_tab, verbose, paf) if ctg_tab is not None else None,
                                     k=k, z=z, x=x, sensitive=sensitive, repeat_filter=repeats)
                    finally:
                        if ctg_tab is not None:
                            ctg_tab.close()
        t_fin = time.perf_counter()
        drain.close()
        if tsv_drain:
            tsv_drain.close()
        out.close()
        stats["t_drain_tail"] = time.perf_counter() - t_fin  # writers still busy after the last device batch
        stats["parsed_bytes"] = io_stats.get("parsed_bytes", 0)
        stats["reader"] = {key: round(io_stats.get(key, 0.0), 4) for key in ("t_reader_count", "t_reader_alloc", "t_reader_parse")}
        stats["reader"]["batches"] = io_stats.get("reader_batches", 0)
        stats["reader"]["readers"] = io_stats.get("readers", 1)
        if comm.world > 1:
            t_fin = time.perf_counter()
            exts = [e for e, on in ((".verbose_mapping.tsv", verbose), (".paf", paf)) if on]
            sizes = comm.allgather([os.path.getsize(prefix + e + part) for e in exts])
            for j, e in enumerate(exts):
                _append_part(prefix + e + ".assembling", prefix + e + part, sum(sz[j] for sz in sizes[:comm.rank]))
            comm.barrier()  # every part is in place
            if root:
                for e in exts:
                    os.replace(prefix + e + ".assembling", prefix + e)
            mine = (out.tally.export(), {key: stats[key] for key in ("read_bases", "reads", "read_minimizers", "index_hits", "parsed_bytes")},
                    ctg_share.get("contigs_parsed_by"),
                    {key: round(float(val), 3) for key, val in stats.items() if key.startswith("t_") and isinstance(val, (int, float))})
            parts = comm.gather(mine)  # pair-tally deltas and five counters per rank
            if root:
                stats["parsed_bytes_per_rank"] = [p[1]["parsed_bytes"] for p in parts]
                stats["contigs_parsed_by_per_rank"] = [p[2] for p in parts]  # (one parser per host: shared_contigs)
                stats["t_stages_per_rank"] = json.dumps([p[3] for p in parts])  # every rank's stage seconds (tools/dist_host_scaling.py)
                parts = [p[:2] for p in parts]
                for exported, st in parts[1:]:
                    out.tally.merge(exported)
                    for key in ("read_bases", "reads", "read_minimizers", "index_hits", "parsed_bytes"):
                        stats[key] += st[key]
            stats["t_merge"] = time.perf_counter() - t_fin
        if _TRACE is not None:
            with open(os.environ["NTL_PIPE_TRACE"], "w") as fh:
                fh.write(f"# t_start {t_start:.6f} (perf_counter; NTL_IO_TRACE lines carry the same clock)\n")
                for t, th, what, seq in _TRACE:
                    fh.write(f"{t - t_start:.6f}\t{th}\t{what}\t{seq}\n")
        if root:
            t_fin = time.perf_counter()
            finish_pairs(out.tally, prefix, n, a, pairs_tsv)
            stats["t_graph"] = time.perf_counter() - t_fin
            stats["t_write"], stats["t_tally"] = out.t_write, out.t_tally
        comm.barrier()
    except BaseException:
        for d in (drain, tsv_drain):
            try:
                if d:
                    d.close()
            except BaseException:
                pass
        out.remove_partial()
        batches.stop()
        raise
    return stats
