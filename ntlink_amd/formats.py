"""Text formats at the boundary of the pair stage (byte-for-byte those of the reference).

* indexlr TSV     `id\\t[len\\t]H:pos:strand H:pos:strand ...`   (ntLink:199,223; parsers
                  bin/ntlink_pair.py:197-207,355-378)
* verbose mapping `read\\tcontig\\tn\\tctgpos:ctgstrand_readpos:readstrand ...`
                  (bin/ntlink_pair.py:308-313,382-388)
* PAF-like        12 columns (bin/ntlink_paf_output.py:131-135)
"""
import ctypes as C

import numpy as np

from .seqio import Names

_STRAND = ("-", "+")


def _fd(fh):
    """File descriptor of a real file (native writers), or None for in-memory text objects."""
    try:
        fd = fh.fileno()
    except (AttributeError, OSError, ValueError):
        return None
    fh.flush()
    return fd


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _native():
    from . import capi
    return capi.load()


def write_blob(fh, text):
    """Text made elsewhere (capi.Text.download: uint8 array) behind what the file holds."""
    if not len(text):
        return
    fd = _fd(fh)
    if fd is None:
        fh.write(bytes(text).decode())
        return
    t = np.ascontiguousarray(text)
    if _native().ntl_write_blob(fd, t.ctypes.data, len(t)) != 0:
        raise OSError("write failed")


def write_indexlr(fh, names, lengths, mx_off, mx_hash, pos, strand, with_len, with_strand=True):
    """Every record prints its id even without minimizers (such lines are skipped by the consumers,
    bin/ntlink_pair.py:200,357)."""
    fd = _fd(fh)
    if fd is not None:  # native emitter (csrc/ntl_io.cpp)
        nm = Names.of(names)
        ln = np.ascontiguousarray(lengths, np.uint32) if with_len else None
        off64 = np.ascontiguousarray(mx_off, np.uint64)
        h = np.ascontiguousarray(mx_hash, np.uint64); p = np.ascontiguousarray(pos, np.uint32)
        st = np.ascontiguousarray(strand, np.uint8)
        rc = _native().ntl_write_indexlr(fd, len(nm), nm.blob.ctypes.data, _p(nm.off, C.c_uint64),
                                         _p(ln, C.c_uint32) if with_len else None, _p(off64, C.c_uint64),
                                         _p(h, C.c_uint64), _p(p, C.c_uint32), _p(st, C.c_uint8) if with_strand else None)
        if rc != 0:
            raise OSError("write failed")
        return
    hs = mx_hash.tolist()
    ps = pos.tolist()
    ss = strand.tolist()
    off = [int(v) for v in mx_off]
    for i, name in enumerate(names):
        a, b = off[i], off[i + 1]
        if with_strand:
            toks = " ".join(f"{hs[j]}:{ps[j]}:{_STRAND[ss[j]]}" for j in range(a, b))
        else:
            toks = " ".join(f"{hs[j]}:{ps[j]}" for j in range(a, b))
        if with_len:
            fh.write(f"{name}\t{int(lengths[i])}\t{toks}\n")
        else:
            fh.write(f"{name}\t{toks}\n")


def parse_indexlr(fh, with_len):
    """-> (names, lengths u32 or None, mx_off u64[n+1], hash u64, pos u32, strand u8).  Lines keep their
    place even when they carry no minimizers."""
    names, lens, off, hs, ps, ss = [], [], [0], [], [], []
    col = 2 if with_len else 1
    for line in fh:
        f = line.strip().split("\t")
        if not f or f == [""]:
            continue
        names.append(f[0])
        if with_len:
            lens.append(int(f[1]) if len(f) > 1 else 0)
        if len(f) > col and f[col]:
            for tok in f[col].split(" "):
                a, b, c = tok.split(":")
                hs.append(int(a)); ps.append(int(b)); ss.append(1 if c == "+" else 0)
        off.append(len(hs))
    return (names, np.array(lens, np.uint32) if with_len else None, np.array(off, np.uint64),
            np.array(hs, np.uint64), np.array(ps, np.uint32), np.array(ss, np.uint8))


def read_indexlr(path, with_len, max_bytes=0, with_strand=True):
    """Native, multi-threaded form of parse_indexlr (csrc/ntl_io.cpp, ntl_tsv_*): yields
    (Names, lengths u32 or None, mx_off u64[n+1], hash u64, pos u32, strand u8) per block of about
    max_bytes of text (0: the whole input as one block).  path "-" = stdin.  with_strand=False reads the `H:pos` tokens of
    `indexlr --pos` without `--strand` (strand is then all 1)."""
    L = _native()
    h = C.c_void_p()
    if L.ntl_tsv_open(path.encode(), (1 if with_len else 0) | (0 if with_strand else 2), C.byref(h)) != 0:
        raise OSError(f"cannot open {path}")
    try:
        while True:
            n = C.c_uint64()
            if L.ntl_tsv_next(h, int(max_bytes), C.byref(n)) != 0:
                raise ValueError(f"{path}: {L.ntl_tsv_error(h).decode()}")
            n = n.value
            if n == 0:
                return
            nmx, nb = C.c_uint64(), C.c_uint64()
            L.ntl_tsv_sizes(h, None, C.byref(nmx), C.byref(nb))
            names, noff = np.empty(nb.value, np.uint8), np.empty(n + 1, np.uint64)
            lens = np.empty(n, np.uint32) if with_len else None
            off = np.empty(n + 1, np.uint64)
            hs, ps, ss = np.empty(nmx.value, np.uint64), np.empty(nmx.value, np.uint32), np.empty(nmx.value, np.uint8)
            if L.ntl_tsv_copy(h, names.ctypes.data, noff.ctypes.data, lens.ctypes.data if with_len else None, off.ctypes.data,
                              hs.ctypes.data, ps.ctypes.data, ss.ctypes.data) != 0:
                raise ValueError(f"{path}: copy failed")
            yield Names(names, noff), lens, off, hs, ps, ss
    finally:
        L.ntl_tsv_close(h)


def write_verbose(fh, res, read_names, ctg_names, read_base=0):
    maps, hits = res["maps"], res["hits"]
    fd = _fd(fh)
    if fd is not None and read_base == 0:
        rn, cn = Names.of(read_names), Names.of(ctg_names)
        m = np.ascontiguousarray(maps); h = np.ascontiguousarray(hits)
        rc = _native().ntl_write_verbose(fd, m.ctypes.data, len(m), h.ctypes.data, rn.blob.ctypes.data, _p(rn.off, C.c_uint64),
                                         cn.blob.ctypes.data, _p(cn.off, C.c_uint64))
        if rc != 0:
            raise OSError("write failed")
        return
    cp = hits["ctg_pos"].tolist(); rp = hits["read_pos"].tolist()
    cs = hits["ctg_strand"].tolist(); rs = hits["read_strand"].tolist()
    for r, c, n, o in zip(maps["read"].tolist(), maps["ctg"].tolist(), maps["n_hits"].tolist(), maps["hit_off"].tolist()):
        toks = " ".join(f"{cp[j]}:{_STRAND[cs[j]]}_{rp[j]}:{_STRAND[rs[j]]}" for j in range(o, o + n))
        fh.write(f"{read_names[r + read_base]}\t{ctg_names[c]}\t{n}\t{toks}\n")


def write_paf(fh, res, read_names, read_len, ctg_names, ctg_len, read_base=0):
    p = res["pafs"]
    fd = _fd(fh)
    if fd is not None and read_base == 0:
        rn, cn = Names.of(read_names), Names.of(ctg_names)
        q = np.ascontiguousarray(p)
        rl = np.ascontiguousarray(read_len, np.uint32); cl = np.ascontiguousarray(ctg_len, np.uint32)
        rc = _native().ntl_write_paf(fd, q.ctypes.data, len(q), rn.blob.ctypes.data, _p(rn.off, C.c_uint64), _p(rl, C.c_uint32),
                                     cn.blob.ctypes.data, _p(cn.off, C.c_uint64), _p(cl, C.c_uint32))
        if rc != 0:
            raise OSError("write failed")
        return
    for r, c, qs, qe, ts, te, n, st in zip(p["read"].tolist(), p["ctg"].tolist(), p["q_start"].tolist(), p["q_end"].tolist(),
                                            p["t_start"].tolist(), p["t_end"].tolist(), p["n_hits"].tolist(), p["strand"].tolist()):
        fh.write(f"{read_names[r + read_base]}\t{int(read_len[r + read_base])}\t{qs}\t{qe}\t{_STRAND[st]}\t"
                 f"{ctg_names[c]}\t{int(ctg_len[c])}\t{ts}\t{te}\t{n}\t{te - ts}\t255\n")


def parse_verbose(fh):
    """Checkpoint reader (bin/ntlink_pair.py:437-488, bin/ntlink_utils.py:296-305): yields
    (read_id, [(contig, [(ctg_pos, ctg_strand, read_pos, read_strand), ...]), ...]) per read."""
    cur, entries = None, []
    for line in fh:
        read_id, contig, _n, mx = line.strip().split("\t")
        hl = []
        for tok in mx.split(" "):
            c, r = tok.split("_")
            cpos, cst = c.split(":")
            rpos, rst = r.split(":")
            hl.append((int(cpos), 1 if cst == "+" else 0, int(rpos), 1 if rst == "+" else 0))
        if read_id != cur:
            if cur is not None:
                yield cur, entries
            cur, entries = read_id, []
        entries.append((contig, hl))
    if cur is not None:
        yield cur, entries
