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
