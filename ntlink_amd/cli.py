"""Command-line surfaces of the pair stage, argument-compatible with the reference's executables.

  indexlr_main      `indexlr --long --pos --strand [--len] -k K -w W [-t T] FILE|-`   (ntLink:199,223)
  ntlink_pair_main  `ntlink_pair.py ...` with the reference's argparse               (bin/ntlink_pair.py:509-536)
  ntlink_main       `ntLink pair target= reads= [k= w= ...]` make-style key=value    (ntLink:8-89,165)
"""
import argparse
import os
import sys
import time

VERSION = "ntLink v1.3.11 pair stage, MI355X build 0.1.0"


def _device(ordinal=0):
    from . import capi
    return capi.Device(ordinal)


def indexlr_main(argv=None):
    ap = argparse.ArgumentParser(prog="indexlr", description="(k,w) minimizer sketch on MI355X; output format of btllib indexlr")
    ap.add_argument("-k", type=int, required=True)
    ap.add_argument("-w", type=int, required=True)
    ap.add_argument("-t", type=int, default=1, help="accepted for compatibility (the GPU does the work)")
    ap.add_argument("--long", action="store_true", help="accepted: long mode is the only mode")
    ap.add_argument("--pos", action="store_true")
    ap.add_argument("--strand", action="store_true")
    ap.add_argument("--len", action="store_true", dest="with_len")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("files", nargs="+")
    a = ap.parse_args(argv)
    if not a.pos:
        ap.error("only the `--pos [--strand]` outputs that ntLink uses are implemented")
    from . import pipeline
    dev = _device(a.device)
    try:
        pipeline.run_indexlr(dev, a.files, a.k, a.w, sys.stdout, a.with_len, with_strand=a.strand)
    finally:
        dev.close()
    return 0


def ntlink_pair_parser():
    p = argparse.ArgumentParser(description="ntLink: Scaffolding genome assemblies using long reads (pair stage on MI355X)")
    p.add_argument("FILES", nargs="+", help="Long read minimizer TSV files")
    p.add_argument("-s", help="Target scaffolds fasta file", required=True)
    p.add_argument("-m", help="Target scaffolds minimizer TSV file", required=True)
    p.add_argument("-p", help="Output prefix [out]", default="out", type=str)
    p.add_argument("-n", help="Minimum edge weight [1]", default=1, type=int)
    p.add_argument("-k", help="Kmer size used for minimizer step", required=True, type=int)
    p.add_argument("-z", help="Minimum size of contig to scaffold", default=500, type=int)
    p.add_argument("-a", help="Minimum number of anchoring long reads for an edge", type=int, default=1)
    p.add_argument("-f", help="Maximum number of contigs in a run for full transitive edge addition", default=10, type=int)
    p.add_argument("-x", help="Fudge factor allowed between mapping block lengths on read and assembly. "
                              "Set to 0 to allow mapping block to be up to read length", type=float, default=0)
    p.add_argument("-c", "--checkpoint", help="Mappings checkpoint file")
    p.add_argument("--pairs", help="Output pairs TSV file", action="store_true")
    p.add_argument("--paf", help="Output mappings in PAF-like format", action="store_true")
    p.add_argument("--sensitive", help="Run more sensitive read mapping", action="store_true")
    p.add_argument("--repeat-filter", help="Remove repetitive minimizers within a long read's sketch", action="store_true")
    p.add_argument("-v", "--version", action="version", version=VERSION)
    p.add_argument("--verbose", help="Verbose output logging", action="store_true")
    p.add_argument("--device", type=int, default=0, help="GPU ordinal")
    return p


class NtlinkPairError(Exception):
    "ntLink pair exception"


def ntlink_pair_main(argv=None):
    a = ntlink_pair_parser().parse_args(argv)
    print("Running pairing stage of ntLink ...\n")
    from . import pipeline
    try:
        dev = _device(a.device)
        try:
            pipeline.run_ntlink_pair(dev, a)
        finally:
            dev.close()
    except Exception as exc:
        raise NtlinkPairError("ntLink pairing stage encountered an error..") from exc
    return 0


_DEFAULTS = dict(target="None", reads="None", w="100", k="32", t="4", z="1000", n="1", a="1", f="10", x="0",
                 sensitive="False", repeats="False", verbose="True", ntlink_pairs_tsv="False", paf="False", v="0",
                 prefix=None, device="0")


def ntlink_main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    targets, kv, given = [], dict(_DEFAULTS), set()
    for tok in argv:
        if tok.startswith("-"):
            continue  # make flags such as -B
        if "=" in tok:
            key, val = tok.split("=", 1)
            kv[key] = val
            given.add(key)
        else:
            targets.append(tok)
    if not targets or "help" in targets:
        print("Usage: ntLink pair target=<target scaffolds> reads='List of long read files' [k= w= z= n= a= f= x= "
              "paf= verbose= sensitive= repeats= ntlink_pairs_tsv= prefix= device= t= v=]\n"
              "Only the `pair` stage (minimizer sketch + read-to-contig mapping) runs on the GPU; "
              "scaffold/gap_fill stay with the reference pipeline.")
        return 0
    if "version" in targets:
        print(VERSION)
        return 0
    if targets != ["pair"]:
        print(f"ERROR: target(s) {targets} are outside the GPU pair stage; run them with the reference ntLink", file=sys.stderr)
        return 2
    if kv["reads"] == "None":
        print("ERROR: Must set reads", file=sys.stderr)
        return 2
    if kv["target"] == "None":
        print("ERROR: Must set target", file=sys.stderr)
        return 2
    from . import pipeline
    apply_threads(kv, given)
    t0 = time.perf_counter()
    dev = _device(int(kv["device"]))
    try:
        stats = pipeline.run_pair(dev, kv["target"], kv["reads"], prefix=kv["prefix"], k=int(kv["k"]), w=int(kv["w"]), n=int(kv["n"]),
                                  a=int(kv["a"]), z=int(kv["z"]), f=int(kv["f"]), x=float(kv["x"]), paf=kv["paf"] == "True",
                                  verbose=kv["verbose"] == "True", sensitive=kv["sensitive"] == "True", repeats=kv["repeats"] == "True",
                                  pairs_tsv=kv["ntlink_pairs_tsv"] == "True")
    finally:
        dev.close()
    if kv["v"] != "0":
        write_time_file(kv, "ntLink " + " ".join(argv), time.perf_counter() - t0, stats)
    return 0


def apply_threads(kv, given):
    """`t=` (ntLink:27: threads of indexlr) sets the parser / emitter thread count of the native I/O when it was given
    on the command line; without it the library's own default (min(cores, 32)) stays."""
    if "t" in given and "NTL_IO_THREADS" not in os.environ:
        try:
            os.environ["NTL_IO_THREADS"] = str(max(1, int(kv["t"])))
        except ValueError:
            raise SystemExit(f"ERROR: t={kv['t']} is not a number")


def write_time_file(kv, command, wall_s, stats):
    """`v=1` (ntLink:100-110): the reference wraps every recipe in `time -v -o $@.time`.  The fused driver runs the
    two recipes of the pair stage (ntLink:198-199,221-225) in one process, so one file is written next to the last target,
    <prefix>.n<n>.scaffold.dot.time, with GNU time's -v lines for this process and the stage times of the driver."""
    import resource
    ru, rc = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
    user, system = ru.ru_utime + rc.ru_utime, ru.ru_stime + rc.ru_stime
    prefix = kv["prefix"] or f"{kv['target']}.k{kv['k']}.w{kv['w']}.z{kv['z']}"
    m, sec = divmod(wall_s, 60.0)
    h, m = divmod(int(m), 60)
    elapsed = f"{h}:{m:02d}:{sec:05.2f}" if h else f"{m}:{sec:05.2f}"
    lines = [f"\tCommand being timed: \"{command}\"",
             f"\tUser time (seconds): {user:.2f}",
             f"\tSystem time (seconds): {system:.2f}",
             f"\tPercent of CPU this job got: {int(100 * (user + system) / max(wall_s, 1e-9))}%",
             f"\tElapsed (wall clock) time (h:mm:ss or m:ss): {elapsed}",
             f"\tMaximum resident set size (kbytes): {max(ru.ru_maxrss, rc.ru_maxrss)}",
             f"\tMajor (requiring I/O) page faults: {ru.ru_majflt}",
             f"\tMinor (reclaiming a frame) page faults: {ru.ru_minflt}",
             f"\tVoluntary context switches: {ru.ru_nvcsw}",
             f"\tInvoluntary context switches: {ru.ru_nivcsw}",
             f"\tFile system inputs: {ru.ru_inblock}",
             f"\tFile system outputs: {ru.ru_oublock}",
             "\tExit status: 0"]
    for key in sorted(stats or {}):
        if key.startswith("t_") or key in ("read_bases", "reads", "read_minimizers", "index_hits", "index_size", "parsed_bytes",
                                           "parsed_bytes_per_rank", "pin_per_rank", "contigs_parsed_by_per_rank"):
            if key == "t_stages_per_rank":  # (a JSON list, not a number of seconds)
                lines.append(f"\tntlink_amd stage_seconds_per_rank: {stats[key]}")
                continue
            lines.append(f"\tntlink_amd {key}: {stats[key]:.3f}" if isinstance(stats[key], float) else f"\tntlink_amd {key}: {stats[key]}")
    with open(f"{prefix}.n{kv['n']}.scaffold.dot.time", "w") as fh:
        fh.write("\n".join(lines) + "\n")
