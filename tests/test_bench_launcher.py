"""bench.py --gpus N must run N ranks when it is started without a launcher (the driver's SCALE command shape
is `python bench.py --gpus N ...` as well as torch.distributed.run): CPU check of the spawn path with gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_2_spawns_two_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # one JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world"] == 2


def test_bench_line_on_the_simt_mock(tmp_path):
    """The bench's own logic (device-side workload, contig stage once, steps over resident sub-batches, JSON contract)
    on the SIMT mock at a tiny scale; the numbers mean nothing here."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sim import simlib
    lib = simlib.build()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", lib, "--workload", "C2", "--scale", "0.0004",
                        "--steps", "1", "--warmup", "1", "--no-e2e", "--batch-bases", "100000"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["workload"].startswith("C2:") and "2 distinct HBM-resident sub-batches" in out["config"]["workload"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "roofline", "cpu_baseline"):
        assert key in out
    assert out["n_gpus"] == 1 and out["cpu_baseline"]["kind"] == "port" and "stages_s" in out["cpu_baseline"]
    assert out["config"]["index_size"] > 0 and out["config"]["mappings_hits_pafs_per_step"][0] > 0
    # round 6: the roofline block says what binds, keeps the contract's HBM fields, and the line says that no scaling curve was measured
    roof = out["roofline"]
    assert roof["bound"] == "valu" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-4
    assert "scaling_note" in out and "No 1 -> 8 GPU curve" in out["scaling_note"]


def test_valu_roofline_fraction_is_priced_on_the_floor(monkeypatch):
    """roofline.valu.frac = the floor's instructions at their calibrated issue cost / the SIMD cycles of the launch -- it must fall when
    the kernel issues more instructions for the same bases, which the kernel's own mix priced against its own cycles
    (issue_slots_filled) cannot show."""
    sys.path.insert(0, ROOT)
    import bench
    floor = sum(n * bench.valu_class_cycles(c) for c, n in bench.VALU_FLOOR_CLASSES)
    assert 27.0 < floor < 31.0 and sum(n for _, n in bench.VALU_FLOOR_CLASSES) == bench.VALU_FLOOR_PER_KMER
    monkeypatch.setattr(bench, "priced_cycles", lambda pm, n: {"cycles": 3.3, "how": "test"})
    bases = 3_900_000_000
    lean = bench.valu_roofline({"valu_wave_instr_per_launch": 1.2e9, "measured_cycles_per_wave_instr": 3.3, "clock_ghz": 2.1}, 1.8, bases)
    fat = bench.valu_roofline({"valu_wave_instr_per_launch": 2.4e9, "measured_cycles_per_wave_instr": 3.3, "clock_ghz": 2.1}, 3.6, bases)
    assert abs(lean["frac"] - floor * (bases / 64) / (3.3 * 1.2e9)) < 1e-3
    assert abs(fat["frac"] - lean["frac"] / 2) < 1e-3                      # twice the instructions for the same bases: half the fraction
    assert lean["issue_slots_filled"] == fat["issue_slots_filled"] == 1.0  # ... which the self-priced figure does not see
    assert abs(lean["useful_frac"] - 9.0 / (1.2e9 * 64 / bases)) < 1e-3


def test_device_state_poller_reads_the_busiest_card(tmp_path, monkeypatch):
    """config.device_state_during_timed_steps: shader clock / socket power / junction temperature from the amdgpu hwmon files while the
    timed steps run (the step follows the clock, and the boxes differ in the clock they sustain).  No hwmon: no field, never an error;
    no PCI address from the runtime: the card that draws the most power."""
    import glob
    import time
    sys.path.insert(0, ROOT)
    import bench
    with bench.DeviceStatePoller(0) as p:
        pass
    assert p.summary() is None or "sclk_mhz" in p.summary()  # (None here: no amdgpu hwmon in the CPU container)
    dirs = []
    for i, (mhz, watt) in enumerate(((2400, 290), (2130, 1190))):
        d = tmp_path / f"card{i}" / "device" / "hwmon" / "hwmon0"
        d.mkdir(parents=True)
        for name, v in (("freq1_label", "sclk"), ("freq1_input", mhz * 10**6), ("power1_input", watt * 10**6), ("temp2_input", 58000), ("power1_cap", 1400 * 10**6)):
            (d / name).write_text(f"{v}\n")
        dirs.append(str(d))
    monkeypatch.setattr(glob, "glob", lambda pat: dirs)
    with bench.DeviceStatePoller(0, period=0.01) as p:
        time.sleep(0.08)
    st = p.summary()
    assert st["samples"] >= 2 and st["source"] == dirs[1] and "most power" in st["picked"]
    assert st["sclk_mhz"]["median"] == 2130.0 and st["socket_power_w"]["max"] == 1190.0 and st["junction_c"]["min"] == 58.0 and st["power_cap_w"] == 1400.0


def test_two_ranks_strong_scaling_line_on_the_simt_mock():
    """`bench.py --gpus 2` (strong scaling is the default for N > 1: BASELINE.json configs[3]) end to end on CPU: the launcher starts two gloo ranks, each maps its half of the read
    set on the SIMT mock, rank 0 prints one line with n_gpus = 2, the max-over-ranks time and the summed bases."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sim import simlib
    lib = simlib.build()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--lib", lib, "--workload", "C2", "--scale", "0.0004", "--steps", "1",
            "--warmup", "1", "--no-e2e", "--no-cpu-baseline", "--batch-bases", "100000"]
    p = subprocess.run(base + ["--gpus", "2", "--min-batches", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)  # bare --gpus N = configs[3]
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["world_size"] == 2 and two["rccl_ranks_seen"] == 2
    assert [(r["rank"], r["local_rank"], r["backend"]) for r in two["ranks"]] == [(0, 0, "gloo"), (1, 1, "gloo")] and all(r["device"] for r in two["ranks"])
    per_rank = int(two["config"]["workload"].split(" read bases per GPU")[0].split("+ ")[-1])
    assert abs(2 * per_rank - 200_000) < 40_000  # C2 at scale 0.0004 is 200 kbases of reads: split two ways (a few 10-kb reads each)
    assert two["ms_per_step"] > 0 and two["config"]["index_size"] > 0 and two["config"]["mappings_hits_pafs_per_step"][0] > 0


def test_eight_ranks_strong_scaling_line_on_the_simt_mock():
    """configs[3]'s shape: a bare `bench.py --gpus 8` -- eight gloo ranks, each with an eighth of the read set; the line
    carries every rank's time and bases so that an imbalance would show."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sim import simlib
    lib = simlib.build()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--lib", lib, "--workload", "C2", "--scale", "0.0008", "--steps", "1",
                        "--warmup", "1", "--no-e2e", "--no-cpu-baseline", "--batch-bases", "100000", "--serial-steps", "0", "--gpus", "8", "--min-batches", "4"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["world_size"] == 8 and out["rccl_ranks_seen"] == 8
    assert [r["local_rank"] for r in out["ranks"]] == list(range(8))
    assert out["sub_batches_per_rank"] >= 4  # a rank's share is cut into sub-batches of its own (--min-batches), not the N = 1 size
    assert len(out["per_rank_ms_per_step"]) == 8 and len(out["per_rank_bases_per_step"]) == 8
    # C2 at scale 0.0008: 400 kbases of reads over all ranks; a sub-batch of 12.5 kbases is rounded up to two whole 10-kb reads
    assert 400_000 <= sum(out["per_rank_bases_per_step"]) <= 700_000
    assert max(out["per_rank_ms_per_step"]) == pytest_approx(out["ms_per_step"])


def pytest_approx(v):
    import pytest
    return pytest.approx(v, rel=1e-3, abs=1e-3)
