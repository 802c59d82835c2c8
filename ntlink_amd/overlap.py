"""Consumer of the overlap-stage sketch: the minimizers `ntlink_overlap_sequences.py` keeps per contig.

`indexlr --long --pos -k 15 -w 5` (one minimizer per ~3 bases, ntLink:243-251) feeds `read_minimizers` /
`read_minimizers_path` / `read_minimizer_line` of bin/ntlink_overlap_sequences.py:145-190: minimizers outside the contig's valid
regions are dropped, and so is every hash that occurs more than once on the contig.  Here the text is parsed by the native
TSV reader (csrc/ntl_io.cpp, `H:pos` tokens) and the filter runs on the GPU (csrc/overlap_kernels.h, `ntl_overlap_filter`);
`filter_sketch` is the same on a device-resident sketch, without any text in between.  The functions of the reference keep
their names, arguments and return shapes."""
import io
import os
import re
import tempfile
from collections import defaultdict

import numpy as np

from . import formats

final_sequence_marker_re = re.compile(r"^LAST(ntLink_\d+)$")  # bin/ntlink_overlap_sequences.py:21


def is_in_valid_region(pos, valid_minimizer_positions):
    "True if the minimizer is in a valid position for overlap detection (bin/ntlink_overlap_sequences.py:138-143)"
    return any(start <= pos <= end for start, end in valid_minimizer_positions)


def _regions(names, valid_mx_positions):
    """CSR arrays of the valid regions of the named sequences (none for names outside valid_mx_positions); regions that
    cannot contain a position (end < 0 or end < start) are left out, negative starts clamp to 0"""
    off = np.zeros(len(names) + 1, np.uint64)
    starts, ends = [], []
    for i, name in enumerate(names):
        for start, end in valid_mx_positions.get(name, ()):
            if end < 0 or end < start:
                continue
            starts.append(max(0, min(int(start), 0xFFFFFFFF)))
            ends.append(min(int(end), 0xFFFFFFFF))
        off[i + 1] = len(starts)
    return off, np.array(starts, np.uint32), np.array(ends, np.uint32)


def filter_sketch(dev, sketch, names, valid_mx_positions):
    """Device-resident sketch (e.g. dev.sketch(batch, 15, 5)) -> (mx_off u64[n+1], hash u64, pos u32) of the kept
    minimizers, sequences in input order."""
    off, rs, re_ = _regions(names, valid_mx_positions)
    with dev.overlap_filter(sketch, off, rs, re_) as kept:
        mx_off, h, p, _ = kept.download()
    return mx_off, h, p


def _to_dicts(names, in_off, mx_off, h, p, valid_mx_positions, mx_info, mxs):
    for i, name in enumerate(names):
        if name not in valid_mx_positions or in_off[i + 1] == in_off[i]:
            continue  # a line without a minimizer column is skipped before anything is stored (:173-174)
        a, b = int(mx_off[i]), int(mx_off[i + 1])
        keys = [str(x) for x in h[a:b].tolist()]
        mx_info[name] = {k: (name, q) for k, q in zip(keys, p[a:b].tolist())}
        mxs[name] = [keys]


def _read_blocks(dev, path, valid_mx_positions, mx_info, mxs):
    seen = set()
    for names, _lens, mx_off, h, p, s in formats.read_indexlr(path, False, max_bytes=256 << 20, with_strand=False):
        names = list(names)
        for n in names:
            if n in valid_mx_positions:
                if n in seen:
                    # the reference would carry the first line's dict into the second one (bin/ntlink_overlap_sequences.py:181);
                    # indexlr never prints an id twice, so this is an input error here
                    raise ValueError(f"sequence id {n} occurs on more than one line of the minimizer file")
                seen.add(n)
        with dev.sketch_from_arrays(mx_off, h, p, s) as sk:
            k_off, kh, kp = filter_sketch(dev, sk, names, valid_mx_positions)
        _to_dicts(names, mx_off, k_off, kh, kp, valid_mx_positions, mx_info, mxs)


def read_minimizers(tsv_filename, valid_mx_positions, dev=None):
    """Read the minimizers from a file, removing duplicate minimizers for a given contig
    (bin/ntlink_overlap_sequences.py:145-155) -> (mx_info: contig -> mx -> (contig, position), mxs: contig -> [[mx, ...]])"""
    own = dev is None
    if own:
        from . import capi
        dev = capi.Device(0)
    try:
        mx_info, mxs = defaultdict(dict), {}
        _read_blocks(dev, tsv_filename, valid_mx_positions, mx_info, mxs)
        return mx_info, mxs
    finally:
        if own:
            dev.close()


def read_minimizers_path(mx_reader, valid_mx_positions, dev=None):
    """Read in the minimizers for the given path, stopping at the LAST marker (bin/ntlink_overlap_sequences.py:157-168):
    consumes lines of the open text stream up to and including the `LASTntLink_<n>` line."""
    lines = []
    for line in mx_reader:
        name = line.strip().split("\t")[0]
        if re.search(final_sequence_marker_re, name):
            break
        lines.append(line if line.endswith("\n") else line + "\n")
    mx_info, mxs = defaultdict(dict), {}
    if not lines:
        return mx_info, mxs
    own = dev is None
    if own:
        from . import capi
        dev = capi.Device(0)
    try:
        with tempfile.NamedTemporaryFile("w", suffix=".tsv", dir=os.environ.get("TMPDIR")) as tmp:
            tmp.writelines(lines)
            tmp.flush()
            _read_blocks(dev, tmp.name, valid_mx_positions, mx_info, mxs)
        return mx_info, mxs
    finally:
        if own:
            dev.close()
