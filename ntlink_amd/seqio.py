"""FASTA/FASTQ(.gz) input for the pair stage.

Record semantics follow what the reference feeds to indexlr / reads with bin/read_fasta.py:6-46:
the id is the header up to the first whitespace, multi-line sequences are joined, FASTQ qualities
are skipped, `gzip -cd -f` style transparent decompression (ntLink:113-117,222), several read
files are concatenated in the order given.
"""
import gzip
import io
import sys

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


class SeqSet:
    """Names + one contiguous uint8 buffer + offsets: the form ntl_batch_create takes."""

    def __init__(self, names, buf, offsets):
        self.names, self.buf, self.offsets = names, buf, offsets

    def __len__(self):
        return len(self.names)

    @property
    def lengths(self):
        return np.diff(self.offsets).astype(np.uint32)

    @property
    def bases(self):
        return int(self.offsets[-1])


def load(paths, max_bases=None):
    """Read whole files.  With max_bases, yields SeqSets of about that many bases (batches)."""
    if isinstance(paths, str):
        paths = [paths]

    def gen():
        for p in paths:
            yield from read_fastx(p)

    names, parts, total = [], [], 0
    for name, seq in gen():
        names.append(name)
        parts.append(seq)
        total += len(seq)
        if max_bases is not None and total >= max_bases:
            yield _pack(names, parts)
            names, parts, total = [], [], 0
    if names or max_bases is None:
        yield _pack(names, parts)


def _pack(names, parts):
    off = np.zeros(len(parts) + 1, np.uint64)
    if parts:
        np.cumsum(np.fromiter((len(p) for p in parts), np.uint64, len(parts)), out=off[1:])
    buf = np.frombuffer(b"".join(parts), np.uint8) if parts else np.zeros(0, np.uint8)
    return SeqSet(names, buf, off)


def load_all(paths):
    return next(load(paths))
