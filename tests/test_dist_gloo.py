"""N > 1 path: reads sharded over ranks (world_size 2, gloo, CPU), records gathered in rank order.
The sharded run must leave exactly the files of the single-process run."""
import os
import shutil
import subprocess
import sys

import numpy as np

from helpers import GEN, REF, TEST7_PAF, read_text
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


def test_two_ranks_equal_single_process(tmp_path):
    simlib.build()
    for n in ("scaffolds_4.fa", "long_reads_4_top5.fa"):
        shutil.copy(os.path.join(REF, n), tmp_path / n)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "tests", "dist_worker.py"), "pair", "target=scaffolds_4.fa",
           "reads=long_reads_4_top5.fa", "k=40", "w=100", "paf=True", "ntlink_pairs_tsv=True"]
    assert subprocess.call(cmd, cwd=tmp_path, env=env, timeout=600) == 0
    pre = str(tmp_path / "scaffolds_4.fa.k40.w100.z1000")
    d = os.path.join(GEN, "fixtures", "t7_top5_k40_w100")
    assert read_text(pre + ".verbose_mapping.tsv") == read_text(d + ".verbose_mapping.tsv")
    assert read_text(pre + ".paf") == read_text(d + ".paf")
    assert set(read_text(pre + ".paf").splitlines()) == TEST7_PAF
    assert read_text(pre + ".pairs.tsv") == read_text(d + ".pairs.tsv")
    assert os.path.exists(pre + ".n1.scaffold.dot")
