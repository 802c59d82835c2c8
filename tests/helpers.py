"""Shared test helpers: golden loading and indexlr-TSV parsing (test-side only)."""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = os.path.join(GOLD, "ref")
GEN = os.path.join(GOLD, "gen")

# (tag, target fasta, reads file, k, w, shipped-golden prefix or None)
FIXTURES = [
    ("t1_k32_w250", "scaffolds_1.fa", "long_reads_1.fa", 32, 250, "scaffolds_1.fa.k32.w250"),
    ("t2_k32_w100", "scaffolds_2.fa", "long_reads_2.fq.gz", 32, 100, "scaffolds_2.fa.k32.w100"),
    ("t3_k24_w250", "scaffolds_3.fa", "long_reads_3.fa.gz", 24, 250, "scaffolds_3.fa.k24.w250"),
    ("t4_k40_w100", "scaffolds_4.fa", "long_reads_4.fa.gz", 40, 100, "scaffolds_4.fa.k40.w100"),
    ("t7_top5_k40_w100", "scaffolds_4.fa", "long_reads_4_top5.fa", 40, 100, None),
    ("c1_k32_w100", "scaffolds_1.fa", "long_reads_1.fa", 32, 100, None),
]

SCENARIOS = ["syn_default", "syn_dense", "syn_sensitive", "syn_repeat", "syn_sens_repeat", "syn_x15",
             "syn_x03", "syn_z500_f3", "syn_a2", "syn_many_ctg"]

# the six PAF lines hard-coded in the reference's only test of the path (tests/ntlink_pytest.py:189-194)
TEST7_PAF = {
    "ERR3219854.377839\t21803\t411\t2361\t-\tscaf2\t30523\t100\t2056\t10\t1956\t255",
    "ERR3219854.377839\t21803\t2997\t11206\t-\tscaf1\t8978\t116\t8330\t19\t8214\t255",
    "ERR3219857.526030\t18128\t1182\t7927\t-\tscaf1\t8978\t2\t6781\t12\t6779\t255",
    "ERR3219854.1617584\t20496\t170\t2083\t-\tscaf2\t30523\t122\t2029\t7\t1907\t255",
    "ERR3219854.1617584\t20496\t3012\t10888\t-\tscaf1\t8978\t86\t8022\t13\t7936\t255",
    "ERR3219854.3730316\t18391\t9497\t16949\t+\tscaf1\t8978\t228\t7815\t14\t7587\t255",
}


def read_text(path):
    if path.endswith(".gz"):
        with gzip.open(path, "rt") as f:
            return f.read()
    with open(path) as f:
        return f.read()


def parse_indexlr(text, with_len):
    """indexlr TSV text -> (names, lengths or None, mx_off u64[n+1], hash u64, pos u32, strand u8)."""
    names, lens, off, hs, ps, ss = [], [], [0], [], [], []
    for line in text.split("\n"):
        if not line:
            continue
        f = line.split("\t")
        names.append(f[0])
        col = 2 if with_len else 1
        if with_len:
            lens.append(int(f[1]))
        if len(f) > col and f[col]:
            for tok in f[col].split(" "):
                a, b, c = tok.split(":")
                hs.append(int(a)); ps.append(int(b)); ss.append(1 if c == "+" else 0)
        off.append(len(hs))
    return (names, np.array(lens, np.uint32) if with_len else None, np.array(off, np.uint64),
            np.array(hs, np.uint64), np.array(ps, np.uint32), np.array(ss, np.uint8))


def contig_ids(mx_off):
    """contig id of every minimizer of a contig sketch."""
    n = len(mx_off) - 1
    return np.repeat(np.arange(n, dtype=np.uint32), np.diff(mx_off).astype(np.int64))


def load_scenario(name):
    d = os.path.join(GEN, "synthetic")
    meta = json.load(open(os.path.join(d, name + ".json")))
    ctext = read_text(os.path.join(d, name + ".contigs.tsv.gz"))
    rtext = read_text(os.path.join(d, name + ".reads.tsv.gz"))
    exp = {ext: read_text(os.path.join(d, name + ext + ".gz"))
           for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv")}
    return meta, ctext, rtext, exp


def write_bgzf(path, data, block=0xFF00, level=6):
    """`bgzip` without htslib: gzip members of at most `block` bytes of text, each with the BC extra field that holds its
    compressed size, and the empty EOF member bgzip appends."""
    import struct
    import zlib
    with open(path, "wb") as fh:
        chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]
        for ch in chunks:
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = co.compress(ch) + co.flush()
            bsize = 12 + 6 + len(body) + 8
            fh.write(b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1))
            fh.write(body + struct.pack("<II", zlib.crc32(ch) & 0xFFFFFFFF, len(ch)))
