"""Liftover of the verbose mappings to the coordinates of the scaffolded sequences.

Same names and arguments as the reference's bin/ntlink_liftover_mappings.py (`read_agp` :39-50, `liftover_mappings`
:125-143, `main` :146-161); the per-line / per-read work (`liftover_ctg_mappings` :61-87, `print_adjusted_mappings`
:89-121) is native (csrc/ntl_liftover.cpp, `ntl_liftover` of include/ntlink_amd.h), several threads over pieces of the
file cut where the read id changes.  No GPU is involved."""
import argparse
import ctypes as C

import numpy as np

from . import capi


class AGP:
    "One sequence line of an AGP file (attributes of the reference's class, bin/ntlink_liftover_mappings.py:16-36)"

    def __init__(self, path_id, scaf_start, scaf_end, contig_id, orientation, ctg_start, ctg_end, component_id):
        self.path_id = path_id
        self.scaf_start = int(scaf_start)
        self.scaf_end = int(scaf_end)
        self.contig_id = contig_id
        self.orientation = orientation
        self.ctg_start = int(ctg_start)
        self.ctg_end = int(ctg_end)
        self.component_id = int(component_id)

    def get_ctg_length(self):
        return self.ctg_end - self.ctg_start + 1

    def get_scaf_length(self):
        return self.scaf_end - self.scaf_start + 1

    def __str__(self):
        return f"{self.path_id}, {self.scaf_start}, {self.contig_id}"


def read_agp(agp_filename):
    "AGP file -> {contig id: AGP}; gap lines (component type N or P) are skipped, nine columns are required"
    agp_dict = {}
    with open(agp_filename, "r", encoding="utf-8") as agp_file:
        for line in agp_file:
            path_id, scaf_start, scaf_end, component_id, component_type, ctg_id, ctg_start, ctg_end, orientation = \
                line.strip().split("\t")
            if component_type in ("N", "P"):
                continue
            agp_dict[ctg_id] = AGP(path_id, scaf_start, scaf_end, ctg_id, orientation, ctg_start, ctg_end, component_id)
    return agp_dict


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
    path_blob, path_off = _blob([e.path_id for _, e in entries])
    scaf_start = np.array([e.scaf_start for _, e in entries], dtype=np.int64)
    ctg_start = np.array([e.ctg_start for _, e in entries], dtype=np.int64)
    ctg_end = np.array([e.ctg_end for _, e in entries], dtype=np.int64)
    # only `+` and `-` move anything; every other orientation string means "leave the positions alone"
    ori = bytes((e.orientation.encode()[0] if len(e.orientation.encode()) == 1 else ord("?")) for _, e in entries)
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
