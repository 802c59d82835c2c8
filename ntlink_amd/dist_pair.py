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


def _sys(path):
    """/sys (and /sys/fs/cgroup) under NTL_SYSFS_ROOT when that is set: the tests hand the functions below a made-up host"""
    return os.environ.get("NTL_SYSFS_ROOT", "") + path


def cpu_quota_cores():
    """CPU time the process's cgroup is granted, in cores, or None (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1): a
    container often sees every CPU of its host and is granted a fraction of them."""
    try:
        q, per = open(_sys("/sys/fs/cgroup/cpu.max")).read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open(_sys("/sys/fs/cgroup/cpu/cpu.cfs_quota_us")).read())
            per = int(open(_sys("/sys/fs/cgroup/cpu/cpu.cfs_period_us")).read())
            return q / per if q > 0 else None
        except (OSError, ValueError):
            return None


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes():
    """NUMA node of every GPU in the order the HIP runtime numbers them, read from sysfs alone -- nothing here touches a GPU, so it
    may run before the process has decided which one is its own.  The runtime enumerates the KFD topology nodes that have SIMDs, in
    node order; a node's `drm_render_minor` names /sys/class/drm/renderD<minor>, whose PCI device carries `numa_node`.
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES (ordinals only) re-map the ordinals as the runtime does.  -> list (an entry is None
    where the kernel does not say), or [] without a KFD topology."""
    base = _sys("/sys/class/kfd/kfd/topology/nodes")
    try:
        ids = sorted(int(d) for d in os.listdir(base) if d.isdigit())
    except OSError:
        return []
    nodes = []
    for i in ids:
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, str(i), "properties")) if len(ln.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) == 0:
            continue  # a CPU node
        numa = None
        try:
            numa = int(open(_sys(f"/sys/class/drm/renderD{int(props['drm_render_minor'])}/device/numa_node")).read())
        except (OSError, ValueError, KeyError):
            pass
        nodes.append(numa if numa is not None and numa >= 0 else None)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):  # applied in this order by the runtime
        v = os.environ.get(var)
        if v and all(t.strip().isdigit() for t in v.split(",")):
            nodes = [nodes[int(t)] for t in v.split(",") if int(t) < len(nodes)]
    return nodes


LAST_PIN = {}  # what pin_rank chose for this process (bench.py and the tests report it)


def pin_rank(local_rank, local_world):
    """Several ranks on one host: each keeps to cores of its own, and sizes its parser / emitter thread pools (NTL_IO_THREADS, unless
    the caller set it) to them -- or to its share of the cgroup's CPU quota, one and a half threads per granted core as in the
    single-process default, when that is less: eight ranks must not start eight full-size pools and two device workers each on top
    of one another.  The cores are those of the NUMA node the rank's GPU hangs off (gpu_numa_nodes), shared out among the ranks
    whose GPUs sit on the same node -- page-locked staging buffers and the H2D copies then stay on the socket the GPU's PCIe root
    belongs to; contiguous slices of the affinity mask when the kernel does not say.  With fewer than four cores per rank (granted
    or present) the rank runs one reader and one device worker instead of two of each (NTL_IO_READERS, NTL_DEVICE_STREAMS, unless
    set).  NTL_PIN=0 leaves the affinity alone.  -> cores of this rank, or None."""
    LAST_PIN.clear()
    if local_world <= 1 or os.environ.get("NTL_PIN", "1") == "0" or not hasattr(os, "sched_getaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0)) if "NTL_SYSFS_ROOT" not in os.environ else \
        sorted(_cpulist(open(_sys("/sys/devices/system/cpu/online")).read()))
    per = max(1, len(cpus) // local_world)
    mine, how = cpus[local_rank * per:(local_rank + 1) * per] or cpus, "contiguous slice"
    numa = gpu_numa_nodes()
    if len(numa) >= local_world and all(n is not None for n in numa[:local_world]):
        # One policy for ALL ranks of the host (every rank computes the same answer): NUMA placement only if every rank gets at least
        # one core of its GPU's node -- a cpuset that leaves some node without cores would otherwise put those ranks on contiguous
        # slices that overlap the cores of the NUMA-placed ones.
        node_cpus, shares = {}, {}
        for node in set(numa[:local_world]):
            try:
                node_cpus[node] = [c for c in _cpulist(open(_sys(f"/sys/devices/system/node/node{node}/cpulist")).read()) if c in set(cpus)]
            except (OSError, ValueError):
                node_cpus[node] = []
            shares[node] = len(node_cpus[node]) // sum(1 for r in range(local_world) if numa[r] == node)
        if all(sh >= 1 for sh in shares.values()):
            node = numa[local_rank]
            peers = [r for r in range(local_world) if numa[r] == node]  # ranks whose GPUs share the node, in rank order
            at = peers.index(local_rank) * shares[node]
            mine, how = node_cpus[node][at:at + shares[node]], f"NUMA node {node} of GPU {local_rank}"
    if "NTL_SYSFS_ROOT" not in os.environ:
        try:
            os.sched_setaffinity(0, mine)
        except OSError:
            return None
    threads = min(64, len(mine))
    quota = cpu_quota_cores()
    cores_eff = float(len(mine))
    if quota:
        cores_eff = min(cores_eff, quota / local_world)
        threads = min(threads, int(1.5 * quota / local_world + 0.5))
    os.environ.setdefault("NTL_IO_THREADS", str(max(2, threads)))
    if cores_eff < 4:  # two readers + two device workers + the writer already outnumber the cores
        os.environ.setdefault("NTL_IO_READERS", "1")
        os.environ.setdefault("NTL_DEVICE_STREAMS", "1")
    LAST_PIN.update(cores=len(mine), first_core=mine[0], how=how, numa_node=(numa[local_rank] if local_rank < len(numa) else None),
                    cpu_quota_cores=quota, io_threads=int(os.environ["NTL_IO_THREADS"]),
                    io_readers=os.environ.get("NTL_IO_READERS"), device_streams=os.environ.get("NTL_DEVICE_STREAMS"))
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
    try:
        pin_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    except Exception as exc:  # pinning is an optimisation: an unexpected /sys layout must not cost the run
        print(f"dist_pair: pin_rank failed ({type(exc).__name__}: {exc}); running unpinned", file=sys.stderr)
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
        pins = comm.allgather(dict(LAST_PIN))
        if comm.rank == 0 and kv["v"] != "0":
            stats["pin_per_rank"] = __import__("json").dumps(pins)
            cli.write_time_file(kv, "ntlink_amd.dist_pair " + " ".join(argv), time.perf_counter() - t0, stats)
    finally:
        dev.close()
        comm.dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
