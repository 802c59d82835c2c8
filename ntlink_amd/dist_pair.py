"""`ntLink pair` on N GPUs of one node: one process per GPU, launched by torch.distributed.run.

Reads shard embarrassingly: the contig index is rebuilt on every GPU
(deterministic, <= 1 GB); every rank owns a contiguous byte range of the concatenated read files, parses
only that, maps it, writes its own text and keeps its own pair tally (pipeline.run_pair).  gloo carries two
barriers, the byte counts of the part files and the pair-tally deltas; no record array leaves its process
and no RCCL collective is on the data path.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m ntlink_amd.dist_pair pair target=asm.fa reads='r1.fq.gz r2.fq.gz' k=32 w=250 paf=True
"""
import os
import sys


def cpu_quota_cores():
    """CPU time the process's cgroup is granted, in cores, or None (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1): a
    container often sees every CPU of its host and is granted a fraction of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return q / per if q > 0 else None
        except (OSError, ValueError):
            return None


def pin_rank(local_rank, local_world):
    """Several ranks on one host: each keeps to its own contiguous slice of the cores, and its parser / emitter thread pools
    (NTL_IO_THREADS, unless the caller set it) are sized to that slice -- or to its share of the cgroup's CPU quota, one and a half
    threads per granted core as in the single-process default, when that is less: eight ranks must not start eight full-size pools
    and two device workers each on top of one another.  NTL_PIN=0 leaves the affinity alone.  -> cores of this rank, or None."""
    if local_world <= 1 or os.environ.get("NTL_PIN", "1") == "0" or not hasattr(os, "sched_getaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // local_world)
    mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    threads = min(64, len(mine))
    quota = cpu_quota_cores()
    if quota:
        threads = min(threads, int(1.5 * quota / local_world + 0.5))
    os.environ.setdefault("NTL_IO_THREADS", str(max(2, threads)))
    return len(mine)


class DistComm:
    def __init__(self, backend="gloo"):
        import torch.distributed as dist
        self.dist = dist
        if not dist.is_initialized():
            dist.init_process_group(backend)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def gather(self, obj):
        out = [None] * self.world if self.rank == 0 else None
        self.dist.gather_object(obj, out, dst=0)
        return out

    def allgather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def barrier(self):
        self.dist.barrier()


def main(argv=None):
    from . import cli, pipeline
    argv = list(sys.argv[1:] if argv is None else argv)
    kv, given = dict(cli._DEFAULTS), set()
    targets = []
    for tok in argv:
        if tok.startswith("-"):
            continue
        if "=" in tok:
            key, val = tok.split("=", 1)
            kv[key] = val
            given.add(key)
        else:
            targets.append(tok)
    if targets != ["pair"] or kv["target"] == "None" or kv["reads"] == "None":
        print("usage: ... -m ntlink_amd.dist_pair pair target=<fa> reads='<files>' [k= w= ...]", file=sys.stderr)
        return 2
    cli.apply_threads(kv, given)
    pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    import time
    t0 = time.perf_counter()
    comm = DistComm("gloo")  # a few host objects only; the device work needs no collective
    local = int(os.environ.get("LOCAL_RANK", comm.rank))
    if os.environ.get("NTL_DIST_ONE_DEVICE"):  # every rank on the GPU with this ordinal (tests on a one-GPU box)
        local = int(os.environ["NTL_DIST_ONE_DEVICE"])
    from . import capi
    dev = capi.Device(local)
    try:
        stats = pipeline.run_pair(dev, kv["target"], kv["reads"], prefix=kv["prefix"], k=int(kv["k"]), w=int(kv["w"]), n=int(kv["n"]),
                          a=int(kv["a"]), z=int(kv["z"]), f=int(kv["f"]), x=float(kv["x"]), paf=kv["paf"] == "True",
                          verbose=kv["verbose"] == "True", sensitive=kv["sensitive"] == "True", repeats=kv["repeats"] == "True",
                          pairs_tsv=kv["ntlink_pairs_tsv"] == "True", comm=comm)
        if comm.rank == 0 and kv["v"] != "0":
            cli.write_time_file(kv, "ntlink_amd.dist_pair " + " ".join(argv), time.perf_counter() - t0, stats)
    finally:
        dev.close()
        comm.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
