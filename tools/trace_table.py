#!/usr/bin/env python3
"""NTL_PIPE_TRACE file -> one line per batch: when the reader handed it out, when a device worker took it and was done with it, when
it was committed and written.  usage: tools/trace_table.py trace.tsv [every]"""
import sys
ev = {}
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        continue
    t, th, what, seq = ln.rstrip("\n").split("\t")
    ev.setdefault(what, []).append((float(t), th, int(seq)))
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ro = [t for t, _, _ in ev.get("reader_out", [])]
by = {w: {s: (t, th) for t, th, s in ev.get(w, [])} for w in ("dev_start", "dev_done", "commit", "handed_over", "write_start", "write_done")}
print("seq reader_out dev_start dev_done commit write_start write_done worker")
for s in sorted(by["dev_start"]):
    if s % every:
        continue
    f = lambda w: f"{by[w][s][0]:.3f}" if s in by[w] else "-"
    print(s, f"{ro[s]:.3f}" if s < len(ro) else "-", f("dev_start"), f("dev_done"), f("commit"), f("write_start"), f("write_done"), by["dev_start"][s][1])
