"""N > 1 path of the file-to-file driver: every rank owns a byte range of the concatenated read files (gloo, CPU, the
SIMT-mock build of the kernels standing in for the GPUs).  The sharded run must leave exactly the files of the
single-process run, and every rank must have parsed about 1/N of the input bytes."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from helpers import GEN, REF, TEST7_PAF, read_text, write_bgzf
from ntlink_amd import seqio
from ntlink_amd.pipeline import shard_range
from sim import simlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_in_order():
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 7, 100):
        off = np.zeros(n + 1, np.uint64)
        off[1:] = np.cumsum(rng.integers(0, 30000, n))
        for world in (1, 2, 3, 8):
            r = [shard_range(off, i, world) for i in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            if n >= 50:
                per = [int(off[b] - off[a]) for a, b in r]
                assert max(per) - min(per) < 2 * 30000  # balanced by bases up to one read


def _records(plan):
    out = []
    for ss in seqio.load(plan, max_bases=50_000):
        names = ss.names.tolist()
        for i, n in enumerate(names):
            out.append((n, bytes(ss.buf[int(ss.offsets[i]):int(ss.offsets[i + 1])])))
    return out


@pytest.mark.parametrize("fastq,wrap", [(False, 0), (False, 60), (True, 0)])
def test_byte_range_readers_see_every_record_once(tmp_path, fastq, wrap):
    """seqio.shard_plan + ntl_fastx_open_range: the ranks' record lists, concatenated in rank order, are the records of
    the serial reader -- for cuts anywhere (inside headers, sequences, quality lines that start with '@')."""
    rng = np.random.default_rng(7)
    paths = []
    for f in range(3):
        p = tmp_path / f"r{f}.{'fq' if fastq else 'fa'}"
        with open(p, "w") as fh:
            for i in range(int(rng.integers(1, 60))):
                n = int(rng.integers(1, 3000))
                s = "".join(rng.choice(list("ACGTN"), n))
                if fastq:
                    q = "".join(rng.choice(list("@+>I5"), n))  # qualities that look like headers
                    fh.write(f"@f{f}_{i} c\n{s}\n+\n{q}\n")
                else:
                    body = s if not wrap else "\n".join(s[j:j + wrap] for j in range(0, n, wrap))
                    fh.write(f">f{f}_{i} comment\n{body}\n")
        paths.append(str(p))
    gz = tmp_path / "z.fa.gz"
    import gzip
    with gzip.open(gz, "wt") as fh:
        fh.write(">gz1\nACGT\n>gz2\nGGGTTT\n")
    paths.insert(2, str(gz))  # a file that cannot be cut goes whole to one rank
    want = _records(paths)
    total = sum(os.path.getsize(p) for p in paths)
    for world in (2, 3, 5, 64):
        got, parsed = [], []
        for r in range(world):
            plan = seqio.shard_plan(paths, r, world)
            st = {}
            recs = []
            for ss in seqio.load(plan, max_bases=50_000, stats=st):
                for i, n in enumerate(ss.names.tolist()):
                    recs.append((n, bytes(ss.buf[int(ss.offsets[i]):int(ss.offsets[i + 1])])))
            got += recs
            parsed.append(st.get("parsed_bytes", 0))
        assert got == want, world
        assert sum(parsed) == total
        if world <= 3:
            assert max(parsed) < total / world + 8000  # a rank's share + at most one record and the small gzip file


def _run_ranks(tmp_path, world, port, args, extra_env=None):
    os.makedirs(tmp_path / "shm", exist_ok=True)  # where a host's shared copy of the packed contigs lives for a moment (pipeline.shared_contigs)
    env = dict(os.environ, OMP_NUM_THREADS="1", NTLINK_AMD_LIB=simlib.build(), NTL_IO_THREADS="2", NTL_SHM_DIR=str(tmp_path / "shm"))
    for key, val in (extra_env or {}).items():
        if val is None:
            env.pop(key, None)
        else:
            env[key] = val
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), "pair"] + args
    assert subprocess.call(cmd, cwd=tmp_path, env=env, timeout=900) == 0


@pytest.mark.parametrize("world,port", [(2, 29571), (3, 29572), (8, 29574)])
def test_ranks_equal_single_process(tmp_path, world, port):
    """top-5 reads of test 7 in one plain FASTA: five reads over 2 / 3 / 8 ranks, every cut falls inside a read (with eight
    ranks -- the node size of BASELINE.json configs[3] -- some ranks own no read start at all)."""
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    _run_ranks(tmp_path, world, port, ["target=scaffolds_4.fa", "reads=long_reads_4_top5.fa", "k=40", "w=100", "paf=True",
                                       "ntlink_pairs_tsv=True", "v=1"])
    pre = str(tmp_path / "scaffolds_4.fa.k40.w100.z1000")
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    assert read_text(pre + ".verbose_mapping.tsv") == read_text(d + ".verbose_mapping.tsv")
    assert read_text(pre + ".paf") == read_text(d + ".paf")
    assert set(read_text(pre + ".paf").splitlines()) == TEST7_PAF
    assert read_text(pre + ".pairs.tsv") == read_text(d + ".pairs.tsv")
    assert os.path.exists(pre + ".n1.scaffold.dot")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    # v=1: GNU-time style report with the driver's counters; every rank parsed about its share of the read file
    rep = dict(line.strip().split(": ", 1) for line in open(pre + ".n1.scaffold.dot.time") if ": " in line)
    assert rep["Exit status"] == "0" and float(rep["User time (seconds)"]) > 0
    per = json.loads(rep["ntlink_amd parsed_bytes_per_rank"])
    total = os.path.getsize(tmp_path / "long_reads_4_top5.fa")
    assert sum(per) == total == int(rep["ntlink_amd parsed_bytes"]) and len(per) == world
    longest = max(len(s) for _, s in __import__("oracle").read_fastx(str(tmp_path / "long_reads_4_top5.fa"))) + 200
    assert all(abs(b - total / world) <= longest for b in per)
    # one rank of the host parsed the target FASTA, the others mapped its packed copy (pipeline.shared_contigs)
    assert json.loads(rep["ntlink_amd contigs_parsed_by_per_rank"]) == [0] * world
    assert os.listdir(tmp_path / "shm") == []


def test_three_ranks_many_reads_and_files(tmp_path):
    """test 1 of the reference (plain FASTA reads, k32 w250) cut into two files + a gzip file, three ranks: pairs gathered
    from all ranks (gap lists in read order), outputs byte-identical to the single-process goldens."""
    shutil.copy(os.path.join(REF, "scaffolds_1.fa"), tmp_path / "scaffolds_1.fa")
    recs = list(__import__("oracle").read_fastx(os.path.join(REF, "long_reads_1.fa")))
    import gzip
    cuts = [0, len(recs) // 2, len(recs) * 3 // 4, len(recs)]
    names = ["a.fa", "b.fa.gz", "c.fa"]
    for (a, b), n in zip(zip(cuts, cuts[1:]), names):
        opener = gzip.open if n.endswith(".gz") else open
        with opener(tmp_path / n, "wt") as fh:
            for name, seq in recs[a:b]:
                fh.write(f">{name}\n{seq.decode() if isinstance(seq, bytes) else seq}\n")
    _run_ranks(tmp_path, 3, 29573, ["target=scaffolds_1.fa", "reads=" + " ".join(names), "k=32", "w=250", "paf=True", "ntlink_pairs_tsv=True"])
    pre = str(tmp_path / "scaffolds_1.fa.k32.w250.z1000")
    d = os.path.join(GEN, "fixtures", "t1_k32_w250")
    for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
        assert read_text(pre + ext) == read_text(d + ext), ext
    assert read_text(pre + ".pairs.tsv") == read_text(os.path.join(REF, "expected_outputs", "scaffolds_1.fa.k32.w250.z1000.pairs.tsv"))


@pytest.mark.parametrize("env,want,port", [({"NTL_TEST_FAKE_HOSTS": "2"}, [0, 1, 0], 29577), ({"NTL_SHARE_CONTIGS": "0"}, [0, 1, 2], 29578)],
                         ids=["two_hosts", "no_sharing"])
def test_contigs_are_parsed_once_per_host(tmp_path, env, want, port):
    """pipeline.shared_contigs: with the three ranks dealt out over two (pretended) hosts the lowest rank of each host parses the target
    and the third rank maps its host's copy; NTL_SHARE_CONTIGS=0: every rank parses, as before round 5.  Same files either way."""
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    _run_ranks(tmp_path, 3, port, ["target=scaffolds_4.fa", "reads=long_reads_4_top5.fa", "k=40", "w=100", "paf=True", "ntlink_pairs_tsv=True", "v=1"],
               extra_env=env)
    pre = str(tmp_path / "scaffolds_4.fa.k40.w100.z1000")
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
        assert read_text(pre + ext) == read_text(d + ext), ext
    rep = dict(line.strip().split(": ", 1) for line in open(pre + ".n1.scaffold.dot.time") if ": " in line)
    assert json.loads(rep["ntlink_amd contigs_parsed_by_per_rank"]) == want
    assert os.listdir(tmp_path / "shm") == []


def _fake_host(root, gpus_per_node=(4, 4), cpus_per_node=64, cpu_max="1600000 100000"):
    """A made-up /sys for dist_pair.pin_rank: NUMA nodes with `cpus_per_node` CPUs each, GPUs hanging off them (KFD topology
    nodes behind one CPU node per NUMA node, render minors 128..), and the cgroup's cpu.max."""
    def put(path, text):
        os.makedirs(os.path.dirname(root + path), exist_ok=True)
        with open(root + path, "w") as fh:
            fh.write(text)
    n_nodes = len(gpus_per_node)
    put("/sys/fs/cgroup/cpu.max", cpu_max + "\n")
    put("/sys/devices/system/cpu/online", f"0-{n_nodes * cpus_per_node - 1}\n")
    kfd, minor = 0, 128
    for nd in range(n_nodes):
        put(f"/sys/devices/system/node/node{nd}/cpulist", f"{nd * cpus_per_node}-{(nd + 1) * cpus_per_node - 1}\n")
        put(f"/sys/class/kfd/kfd/topology/nodes/{kfd}/properties", "cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        kfd += 1
    for nd, n in enumerate(gpus_per_node):
        for _ in range(n):
            put(f"/sys/class/kfd/kfd/topology/nodes/{kfd}/properties", f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\n")
            put(f"/sys/class/drm/renderD{minor}/device/numa_node", f"{nd}\n")
            kfd += 1
            minor += 1


def test_eight_ranks_many_reads_and_files(tmp_path):
    """BASELINE.json configs[3] in miniature: eight ranks (one node's worth) share test 1's reads, given as three plain files;
    every rank parses about an eighth of the bytes, the outputs are those of one process, no part file is left behind.
    The ranks see a made-up host -- two NUMA nodes of 64 CPUs with four GPUs each, 16 cores granted by cpu.max, as on the GPU
    boxes of this project: every rank takes cores of its GPU's node and, at two granted cores per rank, one reader, one device
    worker and three parser threads."""
    _fake_host(str(tmp_path / "fakesys"))
    shutil.copy(os.path.join(REF, "scaffolds_1.fa"), tmp_path / "scaffolds_1.fa")
    recs = list(__import__("oracle").read_fastx(os.path.join(REF, "long_reads_1.fa")))
    cuts = [0, len(recs) // 3, len(recs) * 2 // 3, len(recs)]
    names = ["a.fa", "b.fa", "c.fa"]
    for (a, b), n in zip(zip(cuts, cuts[1:]), names):
        with open(tmp_path / n, "wt") as fh:
            for name, seq in recs[a:b]:
                fh.write(f">{name}\n{seq.decode() if isinstance(seq, bytes) else seq}\n")
    _run_ranks(tmp_path, 8, 29575, ["target=scaffolds_1.fa", "reads=" + " ".join(names), "k=32", "w=250", "paf=True", "ntlink_pairs_tsv=True", "v=1"],
               extra_env={"NTL_SYSFS_ROOT": str(tmp_path / "fakesys"), "NTL_IO_THREADS": None, "NTL_IO_READERS": None, "NTL_DEVICE_STREAMS": None})
    pre = str(tmp_path / "scaffolds_1.fa.k32.w250.z1000")
    d = os.path.join(GEN, "fixtures", "t1_k32_w250")
    for ext in (".verbose_mapping.tsv", ".paf", ".pairs.tsv"):
        assert read_text(pre + ext) == read_text(d + ext), ext
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or f.endswith(".assembling")]
    rep = dict(line.strip().split(": ", 1) for line in open(pre + ".n1.scaffold.dot.time") if ": " in line)
    per = json.loads(rep["ntlink_amd parsed_bytes_per_rank"])
    total = sum(os.path.getsize(tmp_path / n) for n in names)
    assert len(per) == 8 and sum(per) == total
    longest = max(len(s) for _, s in recs) + 200
    assert all(abs(b - total / 8) <= longest for b in per)
    pins = json.loads(rep["ntlink_amd pin_per_rank"])
    assert [p["numa_node"] for p in pins] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert [p["first_core"] for p in pins] == [0, 16, 32, 48, 64, 80, 96, 112] and all(p["cores"] == 16 for p in pins)
    assert all(p["cpu_quota_cores"] == 16.0 and p["io_threads"] == 3 and p["io_readers"] == "1" and p["device_streams"] == "1" for p in pins)


def test_a_dead_run_leaves_no_checkpoint(tmp_path):
    """Stale part files and a half-assembled output of a run that died must neither survive nor be taken for a checkpoint."""
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    pre = str(tmp_path / "scaffolds_4.fa.k40.w100.z1000")
    for stale in (".verbose_mapping.tsv.part1", ".paf.part5", ".verbose_mapping.tsv.assembling"):
        with open(pre + stale, "w") as fh:
            fh.write("junk\tfrom\ta\tdead:+_run:+\n")
    _run_ranks(tmp_path, 2, 29576, ["target=scaffolds_4.fa", "reads=long_reads_4_top5.fa", "k=40", "w=100", "paf=True", "ntlink_pairs_tsv=True"])
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    assert read_text(pre + ".verbose_mapping.tsv") == read_text(d + ".verbose_mapping.tsv")
    assert read_text(pre + ".paf") == read_text(d + ".paf")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or f.endswith(".assembling")]


@pytest.mark.parametrize("fastq,block", [(False, 0xFF00), (True, 0xFF00), (False, 700), (True, 333)])
def test_bgzf_member_ranges_see_every_record_once(tmp_path, fastq, block):
    """A BGZF file is cut below file granularity: ranks own ranges of its members (ranges in the compressed file, cut at member
    starts, records cut where the plain-file ranges cut them: the first record start behind the first line end).  The ranks'
    record lists in rank order are the serial reader's, with tiny members too (records spanning dozens of members), and every
    rank's share of the compressed bytes is about total / world."""
    import gzip
    rng = np.random.default_rng(11)
    lines = []
    for i in range(300):
        n = int(rng.integers(1, 9000))
        s = "".join(rng.choice(list("ACGTN"), n))
        if fastq:
            q = "".join(rng.choice(list("@+>I5"), n))
            lines.append(f"@q{i} c\n{s}\n+\n{q}\n")
        else:
            lines.append(f">q{i} comment\n{s}\n")
    data = "".join(lines).encode()
    p = tmp_path / ("r.fq.gz" if fastq else "r.fa.gz")
    write_bgzf(str(p), data, block=block)
    assert gzip.open(p, "rb").read() == data  # a valid multi-member gzip file
    plain = tmp_path / "plain.fa"
    plain.write_bytes(b">x\nACGT\n")
    paths = [str(plain), str(p)]
    want = _records(paths)
    assert len(want) == 301
    total = sum(os.path.getsize(x) for x in paths)
    for world in (1, 2, 3, 8, 50):
        got, parsed = [], []
        for r in range(world):
            plan = seqio.shard_plan(paths, r, world)
            st = {}
            for ss in seqio.load(plan, max_bases=40_000, stats=st):
                for i, n in enumerate(ss.names.tolist()):
                    got.append((n, bytes(ss.buf[int(ss.offsets[i]):int(ss.offsets[i + 1])])))
            parsed.append(st.get("parsed_bytes", 0))
        assert got == want, (world, len(got))
        assert sum(parsed) == total
        if 1 < world <= 8:
            assert max(parsed) < total / world + 0x10000 + 64  # a rank's share + at most one member


def test_pin_rank_choices_on_made_up_hosts(tmp_path, monkeypatch):
    """dist_pair.pin_rank without any process group: NUMA-local cores when the kernel names the GPUs' nodes, contiguous slices
    when it does not, two readers / device workers kept where a rank has four cores or more, HIP_VISIBLE_DEVICES re-mapping."""
    from ntlink_amd import dist_pair
    for var in ("NTL_IO_THREADS", "NTL_IO_READERS", "NTL_DEVICE_STREAMS", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)

    def pin(root, rank, world):
        monkeypatch.setenv("NTL_SYSFS_ROOT", root)
        for var in ("NTL_IO_THREADS", "NTL_IO_READERS", "NTL_DEVICE_STREAMS"):
            monkeypatch.delenv(var, raising=False)
        dist_pair.pin_rank(rank, world)
        return dict(dist_pair.LAST_PIN)

    a = str(tmp_path / "a")  # an uneven host: 6 GPUs on node 0, 2 on node 1, no CPU quota
    _fake_host(a, gpus_per_node=(6, 2), cpus_per_node=48, cpu_max="max 100000")
    assert dist_pair.gpu_numa_nodes.__call__() is not None
    pins = [pin(a, r, 8) for r in range(8)]
    assert [p["numa_node"] for p in pins] == [0] * 6 + [1] * 2
    assert [p["cores"] for p in pins] == [8] * 6 + [24] * 2 and pins[6]["first_core"] == 48 and pins[7]["first_core"] == 72
    assert all(p["io_readers"] is None and p["device_streams"] is None and p["cpu_quota_cores"] is None for p in pins)
    assert pins[0]["io_threads"] == 8 and pins[7]["io_threads"] == 24
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "7,0")  # two ranks on the last and the first GPU
    assert [pin(a, r, 2)["numa_node"] for r in range(2)] == [1, 0]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    b = str(tmp_path / "b")  # the kernel does not say (numa_node -1): contiguous slices of the CPUs
    _fake_host(b, gpus_per_node=(8,), cpus_per_node=64, cpu_max="3200000 100000")
    for m in range(128, 136):
        open(f"{b}/sys/class/drm/renderD{m}/device/numa_node", "w").write("-1\n")
    pins = [pin(b, r, 8) for r in range(8)]
    assert all(p["how"] == "contiguous slice" and p["cores"] == 8 for p in pins) and [p["first_core"] for p in pins] == list(range(0, 64, 8))
    assert all(p["io_threads"] == 6 and p["io_readers"] is None for p in pins)  # 32 granted cores / 8 ranks = 4 each: both pairs kept
    assert dist_pair.pin_rank(0, 1) is None and dist_pair.LAST_PIN == {}
    # a cpuset that holds node 0's cores only, GPUs 4-7 on node 1 -- no rank of node 1 can be NUMA-placed, so NO rank is:
    # eight disjoint slices instead of four NUMA shares overlapped by four fallback slices
    c = str(tmp_path / "c")
    _fake_host(c, gpus_per_node=(4, 4), cpus_per_node=16, cpu_max="1600000 100000")
    with open(c + "/sys/devices/system/cpu/online", "w") as fh:
        fh.write("0-15\n")
    pins = [pin(c, r, 8) for r in range(8)]
    assert all(p["how"] == "contiguous slice" and p["cores"] == 2 for p in pins)
    assert [p["first_core"] for p in pins] == list(range(0, 16, 2))


def test_a_copy_that_cannot_be_published_costs_only_the_sharing(tmp_path):
    """pipeline.shared_contigs: no room or no directory for the host's shared copy of the packed contigs is not a failure of
    the run -- the ranks are told and parse the target themselves, as they did before round 5; same outputs."""
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    _run_ranks(tmp_path, 2, 29581, ["target=scaffolds_4.fa", "reads=long_reads_4_top5.fa", "k=40", "w=100", "paf=True", "v=1"],
               extra_env={"NTL_SHM_DIR": str(tmp_path / "no_such_directory")})
    pre = str(tmp_path / "scaffolds_4.fa.k40.w100.z1000")
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    assert read_text(pre + ".verbose_mapping.tsv") == read_text(d + ".verbose_mapping.tsv")
    assert read_text(pre + ".paf") == read_text(d + ".paf")
    rep = dict(line.strip().split(": ", 1) for line in open(pre + ".n1.scaffold.dot.time") if ": " in line)
    assert json.loads(rep["ntlink_amd contigs_parsed_by_per_rank"]) == [0, 1]


def test_a_target_that_does_not_parse_fails_every_rank(tmp_path):
    """pipeline.shared_contigs: only one rank of a host parses the target; when it fails, the ranks that wait for its copy must fail
    with it -- not hang in the collective -- and nothing stays under /dev/shm."""
    shutil.copy(os.path.join(REF, "long_reads_4_top5.fa"), tmp_path / "long_reads_4_top5.fa")
    os.makedirs(tmp_path / "shm")
    env = dict(os.environ, OMP_NUM_THREADS="1", NTLINK_AMD_LIB=simlib.build(), NTL_IO_THREADS="2", NTL_SHM_DIR=str(tmp_path / "shm"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(ROOT, "tests", "dist_worker.py"), "pair", "target=missing.fa", "reads=long_reads_4_top5.fa",
           "k=40", "w=100"]
    assert subprocess.call(cmd, cwd=tmp_path, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0
    assert os.listdir(tmp_path / "shm") == []
    assert not [f for f in os.listdir(tmp_path) if "verbose_mapping" in f or f.endswith(".paf")]
