"""Seeded adversarial inputs for differential tests (product vs oracle): low-complexity and tandem-repeat
sequences (window ties), N/IUPAC patterns (run table, multi-run strips), lengths around the strip and
window boundaries, and random hit lists for the mapper."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", np.uint8)


def fuzz_sequences(seed, n=40, max_len=9000):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        kind = rng.integers(0, 8)
        ln = int(rng.integers(0, max_len))
        if kind == 0:      # uniform random
            s = ACGT[rng.integers(0, 4, ln)]
        elif kind == 1:    # tandem repeat of a short unit -> many equal hashes
            unit = ACGT[rng.integers(0, 4, int(rng.integers(1, 40)))]
            s = np.tile(unit, ln // len(unit) + 1)[:ln]
        elif kind == 2:    # homopolymer stretches inside random sequence
            s = ACGT[rng.integers(0, 4, ln)].copy()
            for _ in range(int(rng.integers(1, 6))):
                if ln > 10:
                    a = int(rng.integers(0, ln - 5)); b = min(ln, a + int(rng.integers(5, 600)))
                    s[a:b] = ACGT[rng.integers(0, 4)]
        elif kind == 3:    # N runs of all sizes, also at the ends
            s = ACGT[rng.integers(0, 4, ln)].copy()
            for _ in range(int(rng.integers(1, 12))):
                if ln > 2:
                    a = int(rng.integers(0, ln)); b = min(ln, a + int(rng.choice([1, 1, 2, 5, 31, 32, 33, 100, 700])))
                    s[a:b] = ord(rng.choice(list("NnRYKMxX-*")))
        elif kind == 4:    # lower case + duplicated segment (same k-mers twice within a window)
            s = ACGT[rng.integers(0, 4, ln)].copy()
            if ln > 400:
                a = int(rng.integers(0, ln - 300)); seg = s[a:a + 150].copy()
                b = min(ln - 150, a + int(rng.integers(1, 140)))
                s[b:b + 150] = seg
            s = np.where(rng.random(ln) < 0.3, s + 32, s).astype(np.uint8)
        elif kind == 5:    # two-letter alphabet
            s = np.frombuffer(b"AT", np.uint8)[rng.integers(0, 2, ln)]
        elif kind == 6:    # exactly around window / strip boundaries
            s = ACGT[rng.integers(0, 4, int(rng.choice([0, 1, 31, 32, 33, 130, 131, 132, 2046, 2047, 2048, 2079, 4095, 4127, 4128])))]
        else:              # mostly N
            s = np.full(ln, ord("N"), np.uint8)
            if ln > 50:
                a = int(rng.integers(0, ln - 40)); s[a:a + int(rng.integers(1, 40))] = ord("A")
        out.append(bytes(bytearray(s)))
    return out


def fuzz_mapping(seed, n_ctg=12, n_reads=300, k=24):
    """Random index + reads as arrays: (coff, ch, cp, cs, ctg_len, roff, rlen, rh, rp, rs)."""
    rng = np.random.default_rng(seed)
    ctg_len = rng.choice([300, 999, 1000, 5000, 60000], n_ctg).astype(np.uint32)
    ctg_len[0] = 60000
    keys, ch, cp, cs, coff = set(), [], [], [], [0]
    per = []
    for c in range(n_ctg):
        pos = np.sort(rng.choice(np.arange(0, max(1, int(ctg_len[c]) - k)), size=min(int(ctg_len[c]) // 60 + 1, 400), replace=False))
        lst = []
        for p in pos:
            key = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
            if rng.random() < 0.03 and keys:
                key = int(rng.choice(list(keys)))   # duplicate somewhere in the assembly
            keys.add(key)
            lst.append((key, int(p), int(rng.integers(0, 2))))
        per.append(lst)
        for key, p, s in lst:
            ch.append(key); cp.append(p); cs.append(s)
        coff.append(len(ch))
    rh, rp, rs, roff, rlen = [], [], [], [0], []
    for _ in range(n_reads):
        toks = []
        for _seg in range(int(rng.choice([1, 1, 2, 3, 5, 14]))):
            c = int(rng.integers(0, n_ctg))
            if not per[c]:
                continue
            a = int(rng.integers(0, len(per[c]))); b = min(len(per[c]), a + int(rng.choice([1, 2, 4, 9, 30, 200])))
            seg = per[c][a:b]
            if rng.random() < 0.5:
                seg = seg[::-1]
            seg = [t for t in seg if rng.random() > 0.2]
            mode = rng.random()
            if mode < 0.3 and len(seg) > 2:
                for _s in range(int(rng.integers(1, 4))):
                    i = int(rng.integers(0, len(seg) - 1)); seg[i], seg[i + 1] = seg[i + 1], seg[i]
            elif mode < 0.4:
                rng.shuffle(seg)
            if rng.random() < 0.2 and seg:
                seg.insert(int(rng.integers(0, len(seg) + 1)), seg[int(rng.integers(0, len(seg)))])
            toks.extend((key, int(rng.integers(0, 2)) if rng.random() < 0.1 else st) for key, _p, st in seg)
        for _m in range(int(rng.choice([0, 2, 8]))):
            toks.insert(int(rng.integers(0, len(toks) + 1)), (int(rng.integers(0, 2 ** 63)) * 2 + 1, int(rng.integers(0, 2))))
        pos = int(rng.integers(0, 200))
        for key, st in toks:
            rh.append(key); rp.append(pos); rs.append(st)
            pos += int(rng.integers(1, 400))
        roff.append(len(rh))
        rlen.append(pos + k + int(rng.integers(0, 100)))
    u = lambda a, t: np.array(a, t)
    return (u(coff, np.uint64), u(ch, np.uint64), u(cp, np.uint32), u(cs, np.uint8), ctg_len,
            u(roff, np.uint64), u(rlen, np.uint32), u(rh, np.uint64), u(rp, np.uint32), u(rs, np.uint8))
