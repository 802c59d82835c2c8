"""Deterministic synthetic assemblies and long reads.

Genome: i.i.d. uniform ACGT chromosomes cut into contigs separated by dropped gaps U[100,5000]
so that reads span contig junctions.  Reads: uniform start on the pre-gap chromosome, 50/50
strand, fixed or log-normal length, substitution / insertion / deletion errors.  Everything is a
numpy uint8 ASCII buffer + offsets, the form ntl_batch_create takes.
"""
import numpy as np

_ACGT = np.frombuffer(b"ACGT", np.uint8)
_COMP = np.zeros(256, np.uint8)
for _a, _b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    _COMP[_a] = _b


def random_bases(rng, n):
    return _ACGT[rng.integers(0, 4, n, dtype=np.uint8)]


def make_assembly(seed, n_chrom, contigs_per_chrom, contig_len, gap_lo=100, gap_hi=5000, n_run_every=0):
    """Returns (chromosomes [uint8 arrays], contig buffer uint8, contig offsets u64[n+1], names,
    contig spans [(chrom, start, end)])."""
    rng = np.random.default_rng(seed)
    chroms, parts, spans = [], [], []
    for c in range(n_chrom):
        gaps = rng.integers(gap_lo, gap_hi + 1, contigs_per_chrom)
        lens = np.maximum(1000, (contig_len * rng.uniform(0.7, 1.3, contigs_per_chrom)).astype(np.int64))
        total = int(lens.sum() + gaps.sum())
        g = random_bases(rng, total)
        chroms.append(g)
        p = 0
        for j in range(contigs_per_chrom):
            s, e = p, p + int(lens[j])
            ctg = g[s:e].copy()
            if n_run_every and (c * contigs_per_chrom + j) % n_run_every == 0 and len(ctg) > 4000:
                a = int(rng.integers(1000, len(ctg) - 2000))
                ctg[a:a + int(rng.integers(1, 300))] = ord("N")
            parts.append(ctg)
            spans.append((c, s, e))
            p = e + int(gaps[j])
    off = np.zeros(len(parts) + 1, np.uint64)
    np.cumsum([len(p) for p in parts], out=off[1:])
    names = [f"ctg{i:06d}" for i in range(len(parts))]
    return chroms, np.concatenate(parts), off, names, spans


def make_reads(seed, chroms, total_bases, mean_len, sub=0.02, ins=0.015, dele=0.015, lognormal_sigma=0.0,
               min_len=1000, max_len=100000):
    """Returns (read buffer uint8, offsets u64[n+1], names).  Error-free reads are sliced from the
    chromosomes, then sparse error events (binomially many, uniformly placed) are applied to the whole buffer."""
    rng = np.random.default_rng(seed)
    clen = np.array([len(c) for c in chroms], np.int64)
    cprob = clen / clen.sum()
    n = max(1, int(np.ceil(total_bases / mean_len)))
    if lognormal_sigma > 0:
        ln = np.clip(rng.lognormal(np.log(mean_len) - lognormal_sigma ** 2 / 2, lognormal_sigma, n),
                     min_len, max_len).astype(np.int64)
    else:
        ln = np.full(n, int(mean_len), np.int64)
    c = rng.choice(len(chroms), size=n, p=cprob) if len(chroms) > 1 else np.zeros(n, np.int64)
    ln = np.minimum(ln, clen[c])
    st = (rng.random(n) * (clen[c] - ln + 1)).astype(np.int64)
    rev = rng.random(n) < 0.5
    off = np.zeros(n + 1, np.int64)
    np.cumsum(ln, out=off[1:])
    r = np.empty(int(off[-1]), np.uint8)
    for i in range(n):
        piece = chroms[c[i]][st[i]:st[i] + ln[i]]
        r[off[i]:off[i + 1]] = _COMP[piece[::-1]] if rev[i] else piece
    tot = len(r)
    lens = ln.copy()
    if sub > 0 and tot:
        pos = rng.integers(0, tot, rng.binomial(tot, sub))
        r[pos] = _ACGT[(np.searchsorted(_ACGT, r[pos]) + rng.integers(1, 4, len(pos))) % 4]
    if dele > 0 and tot:
        pos = np.unique(rng.integers(0, tot, rng.binomial(tot, dele)))
        keep = np.ones(tot, bool)
        keep[pos] = False
        lens -= np.bincount(np.searchsorted(off, pos, side="right") - 1, minlength=n)
        r = r[keep]
        np.cumsum(lens, out=off[1:])
        tot = len(r)
    if ins > 0 and tot:
        pos = np.unique(rng.integers(0, tot, rng.binomial(tot, ins)))
        lens += np.bincount(np.searchsorted(off, pos, side="right") - 1, minlength=n)
        r = np.insert(r, pos, random_bases(rng, len(pos)))
    o = np.zeros(n + 1, np.uint64)
    np.cumsum(lens, out=o[1:])
    assert int(o[-1]) == len(r)
    names = [f"read{i}" for i in range(n)]
    return r, o, names


def workload(name, scale=1.0):
    """Named configurations of BASELINE.json (scaled for tests)."""
    if name == "C2":  # 50 Mbp assembly (100 contigs) + 10x ONT-like 10 kb reads, k32 w100
        return dict(n_chrom=1, contigs_per_chrom=100, contig_len=int(500_000 * scale), read_bases=int(500_000_000 * scale),
                    read_len=10_000, k=32, w=100, sub=0.02, ins=0.015, dele=0.015, sensitive=False)
    if name == "C3":  # 3 Gbp (5000 contigs) + 30x ONT 15 kb, k32 w250
        return dict(n_chrom=25, contigs_per_chrom=200, contig_len=int(600_000 * scale), read_bases=int(90_000_000_000 * scale),
                    read_len=15_000, k=32, w=250, sub=0.02, ins=0.015, dele=0.015, sensitive=False)
    if name == "C5":  # 3 Gbp + 60x HiFi 20 kb, k24 w100 sensitive
        return dict(n_chrom=25, contigs_per_chrom=200, contig_len=int(600_000 * scale), read_bases=int(180_000_000_000 * scale),
                    read_len=20_000, k=24, w=100, sub=0.001, ins=0.0005, dele=0.0005, sensitive=True)
    raise ValueError(name)


# ---- device-side generation (capi.Device.synth_genome / synth_slices): the host only plans ----

def plan_assembly(seed, n_chrom, contigs_per_chrom, contig_len, gap_lo=100, gap_hi=5000):
    """Chromosome lengths and the contig slices (chromosome, start, length) of make_assembly's layout,
    without generating a base: contigs of contig_len * U[0.7, 1.3] separated by dropped gaps U[gap_lo, gap_hi]."""
    rng = np.random.default_rng(seed)
    chrom_len, c_chr, c_start, c_len = [], [], [], []
    for c in range(n_chrom):
        gaps = rng.integers(gap_lo, gap_hi + 1, contigs_per_chrom)
        lens = np.maximum(1000, (contig_len * rng.uniform(0.7, 1.3, contigs_per_chrom)).astype(np.int64))
        starts = np.concatenate(([0], np.cumsum(lens + gaps)[:-1]))
        chrom_len.append(int(lens.sum() + gaps.sum()))
        c_chr.append(np.full(contigs_per_chrom, c, np.uint32))
        c_start.append(starts.astype(np.uint32))
        c_len.append(lens.astype(np.uint32))
    return dict(chrom_len=np.array(chrom_len, np.uint32), ctg_chrom=np.concatenate(c_chr), ctg_start=np.concatenate(c_start),
                ctg_len=np.concatenate(c_len), names=[f"ctg{i:06d}" for i in range(n_chrom * contigs_per_chrom)])


def slice_span(length):
    """Source bases a read of `length` output bases may consume (ntl_synth_slices, include/ntlink_amd.h)."""
    return length + length // 8 + 64


def plan_reads(seed, chrom_len, total_bases, mean_len, lognormal_sigma=0.4, min_len=1000, max_len=100000):
    """Read slices (chromosome, start, length, reverse) for about total_bases bases: start uniform on the
    chromosome, 50/50 strand, log-normal length clipped to [min_len, max_len]."""
    rng = np.random.default_rng(seed)
    clen = np.asarray(chrom_len, np.int64)
    n = max(1, int(np.ceil(total_bases / mean_len)))
    if lognormal_sigma > 0:
        ln = np.clip(rng.lognormal(np.log(mean_len) - lognormal_sigma ** 2 / 2, lognormal_sigma, n), min_len, max_len).astype(np.int64)
    else:
        ln = np.full(n, int(mean_len), np.int64)
    c = rng.choice(len(clen), size=n, p=clen / clen.sum()) if len(clen) > 1 else np.zeros(n, np.int64)
    ln = np.minimum(ln, (clen[c] - 64) * 8 // 9 - 1)
    room = clen[c] - slice_span(ln)
    st = (rng.random(n) * (room + 1)).astype(np.int64)
    rev = (rng.random(n) < 0.5).astype(np.uint8)
    return dict(chrom=c.astype(np.uint32), start=st.astype(np.uint32), length=ln.astype(np.uint32), reverse=rev)


class DeviceWorkload:
    """A named BASELINE.json configuration resident in HBM: chromosomes, contigs, and the reads as
    sub-batches of at most batch_bases bases (one ntl_batch each; a batch is bounded by 32-bit base indices)."""

    def __init__(self, dev, name, scale=1.0, read_bases=None, batch_bases=3_950_000_000, read_seed=2, with_reads=True):
        self.dev, self.name, self.W = dev, name, workload(name, scale)
        W = self.W
        self.plan = plan_assembly(1, W["n_chrom"], W["contigs_per_chrom"], W["contig_len"])
        self.genome = dev.synth_genome(1, self.plan["chrom_len"])
        self.contigs = dev.synth_slices(self.genome, 0, self.plan["ctg_chrom"], self.plan["ctg_start"], self.plan["ctg_len"])
        self.ctg_len = self.plan["ctg_len"]
        self.read_batches, self.read_lens = [], []
        total = int(W["read_bases"] if read_bases is None else read_bases)
        self.read_bases = 0
        if with_reads:
            nb = max(1, -(-total // int(batch_bases)))
            for b in range(nb):
                self.add_reads(total // nb, seed=tuple(np.atleast_1d(read_seed).tolist()) + (b,))

    def make_reads(self, bases, seed):
        W = self.W
        rp = plan_reads(seed, self.plan["chrom_len"], bases, W["read_len"])
        s64 = int(np.random.SeedSequence(seed).generate_state(1, np.uint64)[0])
        rb = self.dev.synth_slices(self.genome, s64, rp["chrom"], rp["start"], rp["length"], rp["reverse"],
                                   sub=W["sub"], ins=W["ins"], dele=W["dele"])
        return rb, rp["length"]

    def add_reads(self, bases, seed):
        rb, ln = self.make_reads(bases, seed)
        self.read_batches.append(rb)
        self.read_lens.append(ln)
        self.read_bases += int(ln.sum())

    def close(self):
        for h in self.read_batches + [self.contigs, self.genome]:
            h.close()
