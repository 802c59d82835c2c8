"""Liftover of the verbose mappings to the coordinates of the scaffolded sequences.

Same entry points as the reference's bin/ntlink_liftover_mappings.py (`read_agp` :39-50, `liftover_mappings`
:125-143, `main` :146-161); `read_agp` returns plain tuples of the columns the native call takes.  The per-line / per-read work (`liftover_ctg_mappings` :61-87, `print_adjusted_mappings`
:89-121) is native (csrc/ntl_liftover.cpp, `ntl_liftover` of include/ntlink_amd.h), several threads over pieces of the
file cut where the read id changes.  No GPU is involved."""
import argparse
import ctypes as C

import numpy as np

from . import capi


# columns of an AGP sequence line, in file order; ntl_liftover takes five of them
_AGP_COLUMNS = ("object", "object_beg", "object_end", "part_number", "component_type", "component_id", "component_beg",
                "component_end", "orientation")


def read_agp(agp_filename):
    """AGP file -> {contig id: (path id, scaf_start, ctg_start, ctg_end, orientation)}: the five columns `ntl_liftover`
    takes, one tuple per component (the last line of a contig id wins: the reference keeps a dict, read_agp :39-50).
    Gap lines (component type N or P) carry no contig and are left out; a line that does not have the nine AGP columns, or
    whose coordinates are not integers, raises ValueError (the reference fails on the same lines)."""
    table = {}
    with open(agp_filename, "r", encoding="utf-8") as fh:
        for lineno, line in enumerate(fh, 1):
            cols = line.strip().split("\t")
            if len(cols) != len(_AGP_COLUMNS):
                raise ValueError(f"{agp_filename}:{lineno}: {len(cols)} columns, an AGP line has {len(_AGP_COLUMNS)}")
            rec = dict(zip(_AGP_COLUMNS, cols))
            if rec["component_type"] in ("N", "P"):
                continue
            int(rec["object_end"]), int(rec["part_number"])  # the reference converts these too: same inputs fail
            table[rec["component_id"]] = (rec["object"], int(rec["object_beg"]), int(rec["component_beg"]),
                                          int(rec["component_end"]), rec["orientation"])
    return table


def _blob(strings):
    enc = [s.encode() for s in strings]
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        off[1:] = np.cumsum([len(b) for b in enc], dtype=np.uint64)
    return b"".join(enc), off


def liftover_mappings(mappings_filename, agp_dict, output, k):
    "Lifts <prefix>.verbose_mapping.tsv over to `output`; returns (lines read, lines written)"
    lib = capi.load()
    entries = list(agp_dict.items())
    ctg_blob, ctg_off = _blob([c for c, _ in entries])
    path_blob, path_off = _blob([e[0] for _, e in entries])
    scaf_start = np.array([e[1] for _, e in entries], dtype=np.int64)
    ctg_start = np.array([e[2] for _, e in entries], dtype=np.int64)
    ctg_end = np.array([e[3] for _, e in entries], dtype=np.int64)
    # only `+` and `-` move anything; every other orientation string means "leave the positions alone"
    ori = bytes((e[4].encode()[0] if len(e[4].encode()) == 1 else ord("?")) for _, e in entries)
    nin, nout = C.c_uint64(0), C.c_uint64(0)
    rc = lib.ntl_liftover(str(mappings_filename).encode(), str(output).encode(), int(k), len(entries),
                          ctg_blob, capi._ptr(ctg_off, C.c_uint64), path_blob, capi._ptr(path_off, C.c_uint64),
                          scaf_start.ctypes.data, ctg_start.ctypes.data, ctg_end.ctypes.data, ori, C.byref(nin), C.byref(nout))
    if rc == capi.NTL_EINVAL:
        raise ValueError(f"liftover of {mappings_filename}: unreadable file or malformed line "
                         "(expected read<TAB>contig<TAB>count<TAB>ctgpos:strand_readpos:strand ...)")
    if rc:
        raise capi.NtlError(f"ntl_liftover failed ({rc})")
    return nin.value, nout.value


def main(argv=None):
    "Liftover the ntLink verbose mappings file (arguments of bin/ntlink_liftover_mappings.py:146-153)"
    parser = argparse.ArgumentParser(description="Liftover of ntLink mappings")
    parser.add_argument("-m", "--mappings", help="Path to the verbose mappings file", required=True)
    parser.add_argument("-a", "--agp", help="Path to the AGP file", required=True)
    parser.add_argument("-o", "--output", help="Output file name", required=True)
    parser.add_argument("-k", "--kmer", help="Kmer size", required=True, type=int)
    from .cli import VERSION
    parser.add_argument("-v", "--version", action="version", version=VERSION)
    args = parser.parse_args(argv)
    liftover_mappings(args.mappings, read_agp(args.agp), args.output, args.kmer)
    return 0
