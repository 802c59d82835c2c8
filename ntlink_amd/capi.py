"""ctypes binding of the C ABI in include/ntlink_amd.h (libntlink_hip.so).

The library is the hipcc-built HIP extension that sits next to this file.  There is no CPU
fallback: importing works anywhere (so that CLIs can print --help), but opening a Device without
the library or without a GPU raises.
"""
import ctypes as C
import threading
import weakref
import os

import numpy as np
from time import perf_counter as _now

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libntlink_hip.so")

# every symbol include/ntlink_amd.h declares
SYMBOLS = [
    "ntl_ctx_create", "ntl_ctx_destroy", "ntl_last_error", "ntl_ctx_device_name", "ntl_ctx_sync", "ntl_ctx_pipelined", "ntl_ctx_set_pipeline",
    "ntl_prof_enable", "ntl_prof_reset", "ntl_prof_get",
    "ntl_batch_create", "ntl_batch_create_packed", "ntl_batch_create_packed_at", "ntl_packed_words", "ntl_batch_destroy", "ntl_batch_nseq", "ntl_batch_bases", "ntl_host_alloc", "ntl_host_free",
    "ntl_synth_genome", "ntl_synth_slices", "ntl_batch_download",
    "ntl_sketch_run", "ntl_sketch_run_indexed", "ntl_sketch_run_for_map", "ntl_sketch_has_records", "ntl_sketch_destroy", "ntl_sketch_nseq", "ntl_sketch_count", "ntl_sketch_download",
    "ntl_sketch_strips", "ntl_sketch_redo_strips", "ntl_sketch_fallback_strips", "ntl_sketch_from_lists", "ntl_sketch_wait", "ntl_mapres_wait",
    "ntl_sketch_from_host", "ntl_overlap_filter",
    "ntl_index_build", "ntl_index_destroy", "ntl_index_size",
    "ntl_map_run", "ntl_mapres_destroy", "ntl_mapres_n_mappings", "ntl_mapres_n_hits", "ntl_mapres_n_pafs",
    "ntl_mapres_n_index_hits", "ntl_mapres_download",
    "ntl_fastx_open", "ntl_fastx_open_range", "ntl_fastx_range", "ntl_fastx_close", "ntl_fastx_error", "ntl_fastx_next", "ntl_fastx_sizes", "ntl_fastx_copy", "ntl_fastx_copy_packed", "ntl_fastx_runs", "ntl_fastx_next_span", "ntl_fastx_parse_span", "ntl_fastx_copy_span", "ntl_fastx_seqs", "ntl_fastx_offsets",
    "ntl_fastx_names", "ntl_fastx_name_offsets", "ntl_write_indexlr", "ntl_write_verbose", "ntl_write_paf",
    "ntl_tsv_open", "ntl_tsv_close", "ntl_tsv_error", "ntl_tsv_next", "ntl_tsv_sizes", "ntl_tsv_copy",
    "ntl_tally_create", "ntl_tally_destroy", "ntl_tally_add", "ntl_tally_npairs", "ntl_tally_ngaps", "ntl_tally_export", "ntl_tally_merge",
    "ntl_liftover",
    "ntl_names_create", "ntl_names_destroy", "ntl_mapres_format", "ntl_text_sizes", "ntl_text_download", "ntl_text_destroy", "ntl_write_blob",
    "ntl_tally_add_ends", "ntl_tally_write", "ntl_io_errno",
]

MAPPING_DT = np.dtype([("read", "<u4"), ("ctg", "<u4"), ("n_hits", "<u4"), ("pad", "<u4"), ("hit_off", "<u8")])
HIT_DT = np.dtype([("ctg_pos", "<u4"), ("read_pos", "<u4"), ("ctg_strand", "u1"), ("read_strand", "u1"),
                   ("pad", "u1", (2,))])
PAF_DT = np.dtype([("read", "<u4"), ("ctg", "<u4"), ("q_start", "<u4"), ("q_end", "<u4"),
                   ("t_start", "<u4"), ("t_end", "<u4"), ("n_hits", "<u4"), ("strand", "<u4")])


class MapParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("z", C.c_int32), ("x", C.c_double),
                ("sensitive", C.c_int32), ("repeat_filter", C.c_int32)]


NTL_EINVAL, NTL_EDEVICE, NTL_ENOMEM, NTL_EINTERNAL, NTL_ERANGE = -1, -2, -3, -4, -5


class NtlError(RuntimeError):
    pass


_libs = {}


def load(path=None):
    """dlopen the C-ABI library and declare its prototypes."""
    # NTLINK_AMD_LIB: another build of the same C ABI (the tests run the kernels' source under a CPU mock of the HIP runtime)
    path = path or os.environ.get("NTLINK_AMD_LIB") or DEFAULT_LIB
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise NtlError(f"HIP extension not built: {path} is missing (run __graft_entry__.build())")
    if path == DEFAULT_LIB and not os.environ.get("NTL_ALLOW_STALE_LIB"):
        # what runs is what the sources beside it say (build.py: content signatures; objects and library travel to the GPU box)
        from . import build
        if not build.library_is_current():
            raise NtlError(f"{path} was not built from the sources beside it (signature mismatch): run __graft_entry__.build()")
    L = C.CDLL(path)
    vp, u64p, u32p, u8p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    L.ntl_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ntl_ctx_destroy.argtypes = [vp]
    L.ntl_ctx_destroy.restype = None
    L.ntl_last_error.argtypes = [vp]
    L.ntl_last_error.restype = C.c_char_p
    L.ntl_ctx_device_name.argtypes = [vp]
    L.ntl_ctx_device_name.restype = C.c_char_p
    L.ntl_ctx_sync.argtypes = [vp]
    L.ntl_ctx_pipelined.argtypes = [vp]
    L.ntl_ctx_set_pipeline.argtypes = [vp, C.c_int]
    L.ntl_sketch_wait.argtypes = [vp]
    L.ntl_mapres_wait.argtypes = [vp]
    L.ntl_prof_enable.argtypes = [vp, C.c_int]
    L.ntl_prof_reset.argtypes = [vp]
    L.ntl_prof_get.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), u64p]
    L.ntl_batch_create.argtypes = [vp, vp, u64p, C.c_uint64, C.POINTER(vp)]
    L.ntl_batch_create_packed.argtypes = [vp, vp, u64p, C.c_uint64, vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.ntl_batch_create_packed_at.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.ntl_fastx_next_span.argtypes = [vp, C.c_uint64, u64p, u64p]
    L.ntl_fastx_parse_span.argtypes = [vp, vp, u64p, u64p, u64p, u64p]
    L.ntl_fastx_copy_span.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, u64p]
    L.ntl_packed_words.argtypes = [C.c_uint64]
    L.ntl_packed_words.restype = C.c_uint64
    L.ntl_batch_destroy.argtypes = [vp]
    L.ntl_batch_destroy.restype = None
    L.ntl_batch_nseq.argtypes = [vp]
    L.ntl_batch_nseq.restype = C.c_uint64
    L.ntl_batch_bases.argtypes = [vp]
    L.ntl_batch_bases.restype = C.c_uint64
    L.ntl_synth_genome.argtypes = [vp, C.c_uint64, u32p, C.c_uint64, C.POINTER(vp)]
    L.ntl_synth_slices.argtypes = [vp, vp, C.c_uint64, C.c_uint64, u32p, u32p, u32p, u8p, C.c_double, C.c_double, C.c_double,
                                   C.POINTER(vp)]
    L.ntl_batch_download.argtypes = [vp, vp, u64p]
    L.ntl_sketch_run.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.ntl_sketch_run_indexed.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.ntl_sketch_run_for_map.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.ntl_sketch_has_records.argtypes = [vp]
    L.ntl_sketch_destroy.argtypes = [vp]
    L.ntl_sketch_destroy.restype = None
    L.ntl_sketch_nseq.argtypes = [vp]
    L.ntl_sketch_nseq.restype = C.c_uint64
    L.ntl_sketch_count.argtypes = [vp]
    L.ntl_sketch_count.restype = C.c_uint64
    for nm in ("ntl_sketch_strips", "ntl_sketch_redo_strips", "ntl_sketch_fallback_strips"):
        getattr(L, nm).argtypes = [vp]
        getattr(L, nm).restype = C.c_uint64
    L.ntl_sketch_from_lists.argtypes = [vp]
    L.ntl_sketch_download.argtypes = [vp, u64p, u64p, u32p, u8p]
    L.ntl_sketch_from_host.argtypes = [vp, C.c_uint64, u64p, u64p, u32p, u8p, C.POINTER(vp)]
    L.ntl_overlap_filter.argtypes = [vp, vp, u64p, u32p, u32p, C.POINTER(vp)]
    L.ntl_index_build.argtypes = [vp, vp, u32p, C.c_uint32, C.POINTER(vp)]
    L.ntl_index_destroy.argtypes = [vp]
    L.ntl_index_destroy.restype = None
    L.ntl_index_size.argtypes = [vp]
    L.ntl_index_size.restype = C.c_uint64
    L.ntl_map_run.argtypes = [vp, vp, vp, u32p, C.POINTER(MapParams), C.POINTER(vp)]
    L.ntl_mapres_destroy.argtypes = [vp]
    L.ntl_mapres_destroy.restype = None
    for nm in ("n_mappings", "n_hits", "n_pafs", "n_index_hits"):
        f = getattr(L, "ntl_mapres_" + nm)
        f.argtypes = [vp]
        f.restype = C.c_uint64
    L.ntl_mapres_download.argtypes = [vp, vp, vp, vp]
    L.ntl_host_alloc.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.ntl_host_free.argtypes = [vp, vp]
    L.ntl_host_free.restype = None
    L.ntl_fastx_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.ntl_fastx_open_range.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.ntl_fastx_range.argtypes = [vp, u64p, u64p]
    L.ntl_fastx_range.restype = None
    L.ntl_fastx_close.argtypes = [vp]
    L.ntl_fastx_close.restype = None
    L.ntl_fastx_error.argtypes = [vp]
    L.ntl_fastx_error.restype = C.c_char_p
    L.ntl_fastx_next.argtypes = [vp, C.c_uint64, u64p]
    L.ntl_fastx_sizes.argtypes = [vp, u64p, u64p, u64p]
    L.ntl_fastx_sizes.restype = None
    L.ntl_fastx_copy.argtypes = [vp, vp, vp, vp, vp]
    L.ntl_fastx_copy_packed.argtypes = [vp, vp, vp, vp, vp, u64p]
    L.ntl_fastx_runs.argtypes = [vp, vp, vp, vp]
    for nm in ("seqs", "offsets", "names", "name_offsets"):
        f = getattr(L, "ntl_fastx_" + nm)
        f.argtypes = [vp]
        f.restype = vp
    L.ntl_write_indexlr.argtypes = [C.c_int, C.c_uint64, vp, u64p, u32p, u64p, u64p, u32p, u8p]
    L.ntl_write_verbose.argtypes = [C.c_int, vp, C.c_uint64, vp, vp, u64p, vp, u64p]
    L.ntl_write_paf.argtypes = [C.c_int, vp, C.c_uint64, vp, u64p, u32p, vp, u64p, u32p]
    L.ntl_tsv_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.ntl_tsv_close.argtypes = [vp]
    L.ntl_tsv_close.restype = None
    L.ntl_tsv_error.argtypes = [vp]
    L.ntl_tsv_error.restype = C.c_char_p
    L.ntl_tsv_next.argtypes = [vp, C.c_uint64, u64p]
    L.ntl_tsv_sizes.argtypes = [vp, u64p, u64p, u64p]
    L.ntl_tsv_sizes.restype = None
    L.ntl_tsv_copy.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.ntl_tally_create.argtypes = [vp, u64p, u32p, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp)]
    L.ntl_tally_destroy.argtypes = [vp]
    L.ntl_tally_destroy.restype = None
    L.ntl_tally_add.argtypes = [vp, vp, C.c_uint64, vp, u32p]
    L.ntl_tally_add_ends.argtypes = [vp, vp, C.c_uint64, vp, u32p]
    L.ntl_names_create.argtypes = [vp, vp, u64p, u32p, C.c_uint64, C.POINTER(vp)]
    L.ntl_names_destroy.argtypes = [vp]
    L.ntl_names_destroy.restype = None
    L.ntl_mapres_format.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.ntl_text_sizes.argtypes = [vp, u64p, u64p, u64p]
    L.ntl_text_sizes.restype = None
    L.ntl_text_download.argtypes = [vp, vp, vp, vp, vp]
    L.ntl_text_destroy.argtypes = [vp]
    L.ntl_text_destroy.restype = None
    L.ntl_write_blob.argtypes = [C.c_int, vp, C.c_uint64]
    for nm in ("npairs", "ngaps"):
        f = getattr(L, "ntl_tally_" + nm)
        f.argtypes = [vp]
        f.restype = C.c_uint64
    L.ntl_tally_export.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.ntl_tally_write.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_char_p, u64p]
    L.ntl_io_errno.argtypes = []
    L.ntl_tally_merge.argtypes = [vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]
    L.ntl_liftover.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint64, vp, u64p, vp, u64p, vp, vp, vp, vp, u64p, u64p]
    _libs[path] = L
    return L


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class _Handle:
    _destroy = None

    def __init__(self, dev, ptr):
        self.dev, self.ptr = dev, ptr
        dev._live.add(self)  # closed with the device at the latest

    def close(self):
        if self.ptr:
            if self.dev.ptr:  # a handle that outlives its context (interpreter shutdown order) must not touch it any more
                getattr(self.dev.L, self._destroy)(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Batch(_Handle):
    """Sequences resident in HBM as packed 2-bit bases + ACGT-run table."""
    _destroy = "ntl_batch_destroy"

    @property
    def nseq(self):
        return int(self.dev.L.ntl_batch_nseq(self.ptr))

    @property
    def bases(self):
        return int(self.dev.L.ntl_batch_bases(self.ptr))

    def download(self):
        """(ASCII bases uint8, offsets u64[nseq+1]) of a pure-ACGT batch (the synthetic ones)."""
        buf = np.empty(max(self.bases, 1), np.uint8)
        off = np.zeros(self.nseq + 1, np.uint64)
        self.dev._chk(self.dev.L.ntl_batch_download(self.ptr, buf.ctypes.data, _ptr(off, C.c_uint64)))
        return buf[:self.bases], off


class Sketch(_Handle):
    """Device-resident minimizer lists (the output of `indexlr --long --pos --strand`)."""
    _destroy = "ntl_sketch_destroy"

    @property
    def nseq(self):
        return int(self.dev.L.ntl_sketch_nseq(self.ptr))

    def wait(self):
        """Block until the device has finished this sketch (the calls that make one only queue work)."""
        self.dev._chk(self.dev.L.ntl_sketch_wait(self.ptr))
        return self

    @property
    def count(self):
        self.wait()
        return int(self.dev.L.ntl_sketch_count(self.ptr))

    @property
    def strips(self):
        return int(self.dev.L.ntl_sketch_strips(self.ptr))

    @property
    def redo_strips(self):
        self.wait()
        return int(self.dev.L.ntl_sketch_redo_strips(self.ptr))

    @property
    def fallback_strips(self):
        self.wait()
        return int(self.dev.L.ntl_sketch_fallback_strips(self.ptr))

    @property
    def has_records(self):
        """False for a sketch made only to be mapped (Device.sketch(..., index=ix, records=False)): nothing to download."""
        return bool(self.dev.L.ntl_sketch_has_records(self.ptr))

    @property
    def from_lists(self):
        """True: the window passes wrote per-strip minimizer lists; False: the bitmask (diagnostics; same result)."""
        self.wait()
        return bool(self.dev.L.ntl_sketch_from_lists(self.ptr))

    def download(self):
        """(mx_off u64[nseq+1], hash u64, pos u32, strand u8 [1 = '+'])."""
        n, ns = self.count, self.nseq
        off = np.zeros(ns + 1, np.uint64)
        h = np.empty(n, np.uint64); p = np.empty(n, np.uint32); s = np.empty(n, np.uint8)
        self.dev._chk(self.dev.L.ntl_sketch_download(self.ptr, _ptr(off, C.c_uint64), _ptr(h, C.c_uint64),
                                                     _ptr(p, C.c_uint32), _ptr(s, C.c_uint8)))
        return off, h, p, s


class Index(_Handle):
    """Contig minimizer -> (contig, position, strand), duplicates removed."""
    _destroy = "ntl_index_destroy"

    def __len__(self):
        return int(self.dev.L.ntl_index_size(self.ptr))


class MapResult(_Handle):
    _destroy = "ntl_mapres_destroy"

    def wait(self):
        """Block until the device has finished this result (Device.map only queues work)."""
        self.dev._chk(self.dev.L.ntl_mapres_wait(self.ptr))
        return self

    @property
    def n_index_hits(self):
        self.wait()
        return int(self.dev.L.ntl_mapres_n_index_hits(self.ptr))

    def counts(self):
        self.wait()
        L = self.dev.L
        return (int(L.ntl_mapres_n_mappings(self.ptr)), int(L.ntl_mapres_n_hits(self.ptr)),
                int(L.ntl_mapres_n_pafs(self.ptr)))

    def download(self, pinned=False):
        """dict(maps=, hits=, pafs=) of structured arrays, read order.  pinned=True puts them in one
        page-locked buffer of the device's pool (a single fast DMA instead of staged copies into fresh
        pages); hand dict["_pinned"] to Device.pinned_release() when the records have been consumed."""
        nm, nh, npf = self.counts()
        if pinned:
            sizes = [nm * MAPPING_DT.itemsize, nh * HIT_DT.itemsize, npf * PAF_DT.itemsize]
            offs = [0, (sizes[0] + 63) & ~63, 0]
            offs[2] = (offs[1] + sizes[1] + 63) & ~63
            base = self.dev.pinned_empty(offs[2] + sizes[2] + 64)
            maps = base[offs[0]:offs[0] + sizes[0]].view(MAPPING_DT)
            hits = base[offs[1]:offs[1] + sizes[1]].view(HIT_DT)
            pafs = base[offs[2]:offs[2] + sizes[2]].view(PAF_DT)
            self.dev._chk(self.dev.L.ntl_mapres_download(self.ptr, maps.ctypes.data, hits.ctypes.data, pafs.ctypes.data))
            return {"maps": maps, "hits": hits, "pafs": pafs, "_pinned": base, "_owner": self.dev}
        maps = np.empty(nm, MAPPING_DT); hits = np.empty(nh, HIT_DT); pafs = np.empty(npf, PAF_DT)
        self.dev._chk(self.dev.L.ntl_mapres_download(self.ptr, maps.ctypes.data, hits.ctypes.data, pafs.ctypes.data))
        return {"maps": maps, "hits": hits, "pafs": pafs}

    def format(self, read_names, ctg_names, verbose=True, paf=True):
        """The text of this result's lines, made on the device (ntl_mapres_format): read_names / ctg_names are NameTables
        (Device.names) with lengths.  The result must stay open until the Text has been downloaded."""
        p = C.c_void_p()
        self.dev._chk(self.dev.L.ntl_mapres_format(self.ptr, read_names.ptr, ctg_names.ptr, int(bool(verbose)), int(bool(paf)), C.byref(p)))
        return Text(self.dev, p)


class NameTable(_Handle):
    """A name table (+ sequence lengths) resident on the device: the strings the text kernels copy into their lines."""
    _destroy = "ntl_names_destroy"


class Text(_Handle):
    """The lines of .verbose_mapping.tsv / .paf of one map result, formatted on the device."""
    _destroy = "ntl_text_destroy"

    def sizes(self):
        v, p, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.dev.L.ntl_text_sizes(self.ptr, C.byref(v), C.byref(p), C.byref(n))
        return int(v.value), int(p.value), int(n.value)

    def download(self):
        """dict(verbose=bytes as uint8, paf=..., maps=, ends= [2 per mapping: first and last hit]) in ONE page-locked buffer of
        the device's pool; hand dict["_pinned"] to Device.pinned_release() when it has been written out."""
        nv, npf, nm = self.sizes()
        sizes = [nv, npf, nm * MAPPING_DT.itemsize, 2 * nm * HIT_DT.itemsize]
        offs, at = [], 0
        for sz in sizes:
            offs.append(at)
            at = (at + sz + 63) & ~63
        base = self.dev.pinned_empty(at + 64)
        verbose = base[offs[0]:offs[0] + nv]
        paf = base[offs[1]:offs[1] + npf]
        maps = base[offs[2]:offs[2] + sizes[2]].view(MAPPING_DT)
        ends = base[offs[3]:offs[3] + sizes[3]].view(HIT_DT)
        self.dev._chk(self.dev.L.ntl_text_download(self.ptr, verbose.ctypes.data, paf.ctypes.data, maps.ctypes.data, ends.ctypes.data))
        return {"verbose": verbose, "paf": paf, "maps": maps, "ends": ends, "_pinned": base, "_owner": self.dev}


class Device:
    """One MI355X: context + stream.  Raises NtlError when there is no GPU or no HIP extension."""

    def __init__(self, device=0, lib_path=None):
        self.L = load(lib_path)
        self.ordinal, self.lib_path = int(device), lib_path
        p = C.c_void_p()
        rc = self.L.ntl_ctx_create(int(device), C.byref(p))
        if rc != 0:
            raise NtlError(f"ntl_ctx_create(device={device}) failed with {rc}: no usable GPU "
                           f"(this package has no CPU path)")
        self.ptr = p
        self._live = weakref.WeakSet()
        self._pinned_free, self._pinned_all, self._pinned_out = [], [], {}  # (address, capacity) of page-locked buffers
        self._pinned_lock = threading.Lock()  # the reader thread takes buffers, the device thread returns them
        self.pin_alloc_s, self.pin_alloc_bytes, self.pin_alloc_caps = 0.0, 0, []

    def workers(self, n):
        """n - 1 further contexts on the same GPU, kept for the life of this Device: the pair driver's device workers.  Their block
        caches and page-locked result buffers survive from one run_pair call to the next."""
        pool = self.__dict__.setdefault("_workers", [])
        while len(pool) < n - 1:
            pool.append(self.clone())
        return pool[:max(0, n - 1)]

    def clone(self):
        """Another context (stream, block cache, staging pool) on the same GPU: a second worker thread of the pair driver
        uploads its batch while this one's kernels run.  Device objects of one context may be used by the other."""
        return Device(self.ordinal, self.lib_path)

    def close(self):
        if self.ptr:
            for wk in self.__dict__.pop("_workers", []):
                wk.close()
            for h in list(self._live):  # device objects die before their context
                h.close()
            for addr, _cap in self._pinned_all:
                self.L.ntl_host_free(self.ptr, addr)
            self._pinned_free, self._pinned_all, self._pinned_out = [], [], {}
            self.L.ntl_ctx_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            exc = NtlError(f"error {rc}: {self.L.ntl_last_error(self.ptr).decode()}")
            exc.code = rc
            raise exc

    @property
    def name(self):
        return self.L.ntl_ctx_device_name(self.ptr).decode()

    def sync(self):
        self._chk(self.L.ntl_ctx_sync(self.ptr))

    @property
    def pipelined(self):
        return bool(self.L.ntl_ctx_pipelined(self.ptr))

    def set_pipeline(self, on):
        """Window stage on its own stream (on) or behind everything else on the one stream (off); drains the device first."""
        self._chk(self.L.ntl_ctx_set_pipeline(self.ptr, int(bool(on))))

    # ---- profiling (HIP events on the context's stream)
    def prof_enable(self, on=True):
        self._chk(self.L.ntl_prof_enable(self.ptr, int(on)))

    def prof_reset(self):
        self._chk(self.L.ntl_prof_reset(self.ptr))

    def prof_get(self, name):
        ms, n = C.c_double(), C.c_uint64()
        self._chk(self.L.ntl_prof_get(self.ptr, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)

    # ---- page-locked staging buffers for sequence bytes
    def pinned_empty(self, nbytes):
        """uint8 array of nbytes in page-locked memory from a small pool (reused buffers are already
        faulted in, and their copy to the device is one DMA).  Give it back with pinned_release();
        the memory is only valid while this Device is open."""
        nbytes = int(nbytes)
        with self._pinned_lock:
            pick = None
            for i, (addr, cap) in enumerate(self._pinned_free):
                if nbytes <= cap <= 4 * nbytes + (1 << 20) and (pick is None or cap < self._pinned_free[pick][1]):
                    pick = i
            if pick is None:
                cap = max(int(nbytes * 1.125) + 4096, 1 << 20)
                p = C.c_void_p()
                t0 = _now()
                self._chk(self.L.ntl_host_alloc(self.ptr, cap, C.byref(p)))
                self.pin_alloc_s += _now() - t0  # page-locking is the expensive part of a process's first pass
                self.pin_alloc_bytes += cap
                self.pin_alloc_caps.append(cap)
                addr = p.value
                self._pinned_all.append((addr, cap))
            else:
                addr, cap = self._pinned_free.pop(pick)
            self._pinned_out[addr] = (addr, cap)
        return np.frombuffer((C.c_uint8 * cap).from_address(addr), np.uint8)[:nbytes]

    def pinned_release(self, arr):
        """No-op for arrays that did not come from pinned_empty()."""
        if arr is None:
            return
        with self._pinned_lock:
            ent = self._pinned_out.pop(arr.ctypes.data, None)
            if ent is not None:
                self._pinned_free.append(ent)

    # ---- the path
    def batch(self, seqs, offsets=None):
        """seqs: list of bytes, or one contiguous bytes/uint8 buffer with offsets[n+1]."""
        if offsets is None:
            lens = np.fromiter((len(s) for s in seqs), np.uint64, len(seqs))
            offsets = np.zeros(len(seqs) + 1, np.uint64)
            np.cumsum(lens, out=offsets[1:])
            buf = np.frombuffer(b"".join(seqs), np.uint8)
        else:
            buf = seqs if isinstance(seqs, np.ndarray) else np.frombuffer(seqs, np.uint8)
            offsets = np.ascontiguousarray(offsets, np.uint64)
        if len(buf) == 0:
            buf = np.zeros(1, np.uint8)
        p = C.c_void_p()
        self._chk(self.L.ntl_batch_create(self.ptr, buf.ctypes.data, _ptr(offsets, C.c_uint64), len(offsets) - 1,
                                          C.byref(p)))
        return Batch(self, p)

    def batch_packed(self, ss):
        """A SeqSet read with packed=True (seqio.load): 2-bit bases + ACGT-run table made by the parser threads."""
        p = C.c_void_p()
        if ss.positions is not None:  # read in one pass: the sequences of a parser thread sit at an upper bound of their place
            lens = ss.lengths
            self._chk(self.L.ntl_batch_create_packed_at(self.ptr, ss.packed.ctypes.data, int(ss.span_positions), ss.positions.ctypes.data,
                                                        lens.ctypes.data, len(lens), ss.seq_run_first.ctypes.data, ss.run_start.ctypes.data,
                                                        ss.run_len.ctypes.data, len(ss.run_start), C.byref(p)))
            return Batch(self, p)
        self._chk(self.L.ntl_batch_create_packed(self.ptr, ss.packed.ctypes.data, _ptr(ss.offsets, C.c_uint64), len(ss.offsets) - 1,
                                                 ss.seq_run_first.ctypes.data, ss.run_start.ctypes.data, ss.run_len.ctypes.data,
                                                 len(ss.run_start), C.byref(p)))
        return Batch(self, p)

    def synth_genome(self, seed, lengths):
        """Uniform random ACGT sequences generated on the device (bench / test support)."""
        ln = np.ascontiguousarray(lengths, np.uint32)
        p = C.c_void_p()
        self._chk(self.L.ntl_synth_genome(self.ptr, int(seed), _ptr(ln, C.c_uint32), len(ln), C.byref(p)))
        return Batch(self, p)

    def synth_slices(self, src, seed, src_seq, src_start, out_len, reverse=None, sub=0.0, ins=0.0, dele=0.0):
        """Slices of the sequences of batch `src` (reverse-complemented where reverse[i]), with per-base
        error events: contigs cut from chromosomes (no errors), reads sampled from them (with errors)."""
        sq = np.ascontiguousarray(src_seq, np.uint32); st = np.ascontiguousarray(src_start, np.uint32)
        ln = np.ascontiguousarray(out_len, np.uint32)
        rv = None if reverse is None else np.ascontiguousarray(reverse, np.uint8)
        if not (len(sq) == len(st) == len(ln)) or (rv is not None and len(rv) != len(ln)):
            raise ValueError("slice arrays must have one entry per output sequence")
        p = C.c_void_p()
        self._chk(self.L.ntl_synth_slices(self.ptr, src.ptr, int(seed), len(ln), _ptr(sq, C.c_uint32), _ptr(st, C.c_uint32),
                                          _ptr(ln, C.c_uint32), None if rv is None else _ptr(rv, C.c_uint8),
                                          float(sub), float(ins), float(dele), C.byref(p)))
        return Batch(self, p)

    def sketch(self, batch, k, w, index=None, records=True):
        """index: the contig Index the sketch will be mapped against -- its minimizers are then looked up while they are emitted
        and Device.map(index, sketch, ...) skips its lookup pass (same records either way).  records=False (with an index): the
        sketch is made only for that map (ntl_sketch_run_for_map): no records to download."""
        p = C.c_void_p()
        if index is None:
            self._chk(self.L.ntl_sketch_run(self.ptr, batch.ptr, int(k), int(w), C.byref(p)))
        elif not records:
            self._chk(self.L.ntl_sketch_run_for_map(self.ptr, batch.ptr, int(k), int(w), index.ptr, C.byref(p)))
        else:
            self._chk(self.L.ntl_sketch_run_indexed(self.ptr, batch.ptr, int(k), int(w), index.ptr, C.byref(p)))
        return Sketch(self, p)

    def sketch_from_arrays(self, mx_off, mx_hash, pos, strand):
        mx_off = np.ascontiguousarray(mx_off, np.uint64)
        h = np.ascontiguousarray(mx_hash, np.uint64); q = np.ascontiguousarray(pos, np.uint32)
        s = np.ascontiguousarray(strand, np.uint8)
        p = C.c_void_p()
        self._chk(self.L.ntl_sketch_from_host(self.ptr, len(mx_off) - 1, _ptr(mx_off, C.c_uint64), _ptr(h, C.c_uint64),
                                              _ptr(q, C.c_uint32), _ptr(s, C.c_uint8), C.byref(p)))
        return Sketch(self, p)

    def overlap_filter(self, sketch, region_off, region_start, region_end):
        """Valid-region filter + per-sequence duplicate removal of the overlap stage (ntl_overlap_filter) -> new Sketch."""
        ro = np.ascontiguousarray(region_off, np.uint64)
        if len(ro) != sketch.nseq + 1:
            raise ValueError("region_off must have nseq + 1 entries")
        rs = np.ascontiguousarray(region_start, np.uint32); re = np.ascontiguousarray(region_end, np.uint32)
        if len(rs) != int(ro[-1]) or len(re) != len(rs):
            raise ValueError("region arrays must hold region_off[-1] entries")
        if len(rs) == 0:
            rs = np.zeros(1, np.uint32); re = np.zeros(1, np.uint32)
        p = C.c_void_p()
        self._chk(self.L.ntl_overlap_filter(self.ptr, sketch.ptr, _ptr(ro, C.c_uint64), _ptr(rs, C.c_uint32), _ptr(re, C.c_uint32),
                                            C.byref(p)))
        return Sketch(self, p)

    def names(self, names, lengths):
        """Device copy of a name table (seqio.Names or a list of str / bytes) with the sequences' lengths."""
        from .seqio import Names
        nm = Names.of(names)
        blob = np.ascontiguousarray(nm.blob)
        off = np.ascontiguousarray(nm.off, np.uint64)
        ln = np.ascontiguousarray(lengths, np.uint32)
        if len(ln) != len(nm):
            raise ValueError("one length per name")
        p = C.c_void_p()
        self._chk(self.L.ntl_names_create(self.ptr, blob.ctypes.data if len(blob) else None, _ptr(off, C.c_uint64),
                                          _ptr(ln, C.c_uint32) if len(ln) else None, len(nm), C.byref(p)))
        return NameTable(self, p)

    def index(self, contig_sketch, ctg_len):
        cl = np.ascontiguousarray(ctg_len, np.uint32)
        p = C.c_void_p()
        self._chk(self.L.ntl_index_build(self.ptr, contig_sketch.ptr, _ptr(cl, C.c_uint32), len(cl), C.byref(p)))
        return Index(self, p)

    def map(self, index, read_sketch, read_len, k, z=1000, x=0.0, sensitive=False, repeat_filter=False):
        rl = np.ascontiguousarray(read_len, np.uint32)
        if len(rl) != read_sketch.nseq:
            raise ValueError("read_len must have one entry per sketched read")
        P = MapParams(int(k), int(z), float(x), int(bool(sensitive)), int(bool(repeat_filter)))
        p = C.c_void_p()
        self._chk(self.L.ntl_map_run(self.ptr, index.ptr, read_sketch.ptr, _ptr(rl, C.c_uint32), C.byref(P),
                                     C.byref(p)))
        return MapResult(self, p)
