#!/usr/bin/env python3
"""Benchmark of the `ntLink pair` hot path on MI355X (BASELINE.json metric).

One step = one pass of the device-resident path over one batch: packed contigs and reads are
already in HBM; the step sketches the contigs, builds the index, sketches the reads, probes, maps
(accepted contigs + PAF blocks) and compacts the results in HBM.  N > 1: one rank per GPU, reads
sharded (every rank gets its own synthetic read set of the same size = weak scaling), contig index
replicated, no data-path collective; torch.distributed (RCCL) only for the barrier and the max.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ntlink_amd import capi, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", help="C2 (default: the metric's 1-GPU config), C3, C5")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the workload (debugging only; reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    torch.cuda.set_device(local_rank)
    if "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU, RCCL for barrier/max only
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when the communicator comes up: keep stdout for the one JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            warm = torch.zeros(1, device="cuda")
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    W = synth.workload(args.workload, args.scale)
    if args.workload != "C2":
        # C3/C5 at full size do not fit a default run: cap the read set per GPU (stated in config)
        W["read_bases"] = min(W["read_bases"], int(4_000_000_000 * args.scale))
    k, w = W["k"], W["w"]
    t0 = time.time()
    chroms, cbuf, coff, cnames, _ = synth.make_assembly(1, W["n_chrom"], W["contigs_per_chrom"], W["contig_len"])
    rbuf, roff, _ = synth.make_reads(2 + rank, chroms, W["read_bases"], W["read_len"], W["sub"], W["ins"], W["dele"],
                                     lognormal_sigma=0.4)
    ctg_len = np.diff(coff).astype(np.uint32)
    read_len = np.diff(roff).astype(np.uint32)
    gen_s = time.time() - t0
    read_bases = int(roff[-1])
    contig_bases = int(coff[-1])

    dev = capi.Device(local_rank)
    t0 = time.time()
    cb = dev.batch(cbuf, coff)
    rb = dev.batch(rbuf, roff)
    upload_s = time.time() - t0
    params = dict(k=k, z=1000, x=0.0, sensitive=W["sensitive"], repeat_filter=False)
    stats = {}

    def step():
        csk = dev.sketch(cb, k, w)
        ix = dev.index(csk, ctg_len)
        rsk = dev.sketch(rb, k, w)
        res = dev.map(ix, rsk, read_len, **params)
        stats.update(read_mx=rsk.count, contig_mx=csk.count, index=len(ix), index_hits=res.n_index_hits,
                     counts=res.counts())
        for h in (res, rsk, ix, csk):
            h.close()

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()
        dev.sync()

    for _ in range(args.warmup):
        step()
    dev.prof_enable(True)
    dev.prof_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {nm: dev.prof_get(nm) for nm in ("sketch_meta", "sketch_mask", "sketch_emit", "index", "probe", "map", "compact")}
    dev.prof_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([float(read_bases)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_bases = float(tb.item())
    else:
        total_bases = float(read_bases)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bases * args.steps / elapsed / 1e9
        d = 2.0 / (w + 1)
        # dominant kernel: sketch_mask_kernel.  Algorithmic bytes: 0.25 + 16 d per base.
        mask_ms, mask_n = prof["sketch_mask"]
        bytes_per_step = (0.25 + 16.0 * d) * (read_bases + contig_bases)
        launches_per_step = mask_n / max(args.steps, 1)
        avg_launch_ms = mask_ms / max(mask_n, 1)
        bytes_per_launch = bytes_per_step / max(launches_per_step, 1)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        hfrac = stats["index_hits"] / max(stats["read_mx"], 1)
        traffic, traffic_src = pmc_traffic(args.workload, args.scale)
        out = {
            "metric": "read Gbases/s mapped (ntLink pair, paf=True)",
            "value": round(value, 4), "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W['n_chrom'] * W['contigs_per_chrom']} contigs / {contig_bases} bp assembly + "
                                   f"{read_bases} read bases per GPU ({len(read_len)} reads, mean {W['read_len']} bp, lognormal), "
                                   f"k={k} w={w} z=1000 x=0 sensitive={W['sensitive']} paf=True verbose=True",
                       "scale": args.scale, "hit_fraction": round(hfrac, 4),
                       "read_minimizers": stats["read_mx"], "contig_minimizers": stats["contig_mx"],
                       "index_size": stats["index"], "mappings_hits_pafs": list(stats["counts"]),
                       "device": dev.name, "gen_s": round(gen_s, 1), "upload_s": round(upload_s, 2),
                       "stage_ms_per_step": {nm: round(v[0] / args.steps, 3) for nm, v in prof.items()}},
            "roofline": {"bound": "hbm", "kernel": "sketch_mask_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_base": round(0.25 + 16.0 * d, 4), "avg_launch_ms": round(avg_launch_ms, 4),
                         "launches": mask_n,
                         "note": "integer/VALU-bound kernel (SURVEY 7): see DESIGN.md for the VALU roofline",
                         "valu": valu_roofline(args.workload, args.scale, avg_launch_ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cbuf, coff, ctg_len, rbuf, roff, read_len, k, w, params, stats)
        print(json.dumps(out), flush=True)
    for h in (cb, rb):
        h.close()
    dev.close()
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic(workload, scale):
    """HBM bytes per sketch_mask_kernel launch from the rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE in separate --pmc runs, tools/gpu_round.sh); PMC counters cannot be read
    from inside the process, so the committed summary of the latest profiled run is quoted."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if scale != 1.0 or not os.path.exists(path):
        return None, None
    t = json.load(open(path)).get(workload)
    if not t:
        return None, None
    return t["bytes_per_launch"], t["source"]


def valu_roofline(workload, scale, avg_launch_ms):
    """The roofline that actually binds the kernel: VALU issue.  Peak measured with tools/valu_calib.hip
    (profiles/r01_valu_calibration.txt): one wave64 VALU instruction per 4 cycles per SIMD.  Instructions per
    launch come from the committed PMC pass (SQ_INSTS_VALU), the launch time is the live one."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if scale != 1.0 or not os.path.exists(path) or avg_launch_ms <= 0:
        return None
    t = json.load(open(path)).get(workload, {})
    if "valu_wave_instr_per_launch" not in t:
        return None
    peak = 1024 * 2.2e9 / 4.0  # wave-instructions per second (256 CUs x 4 SIMDs, ~2.2 GHz under load)
    ach = t["valu_wave_instr_per_launch"] / (avg_launch_ms * 1e-3)
    return {"achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "G wave-instr/s", "frac": round(ach / peak, 3),
            "lane_instr_per_base": t.get("valu_lane_instr_per_base"), "source": t.get("valu_source")}


def cpu_baseline(cbuf, coff, ctg_len, rbuf, roff, read_len, k, w, params, stats):
    """The oracle (C restatement of indexlr + ntlink_pair mapping, parity-pinned) timed on this host's
    cores on a bounded sample of the same workload.  kind = "port": the literal reference cannot run
    here (btllib absent; its Python does not travel)."""
    import oracle
    cores = os.cpu_count() or 1
    # sample: all contigs + as many reads as ~2 Gbases of single-thread-equivalent work allows
    budget = int(40e6 * 25 * max(1, min(cores, 64)) ** 0.9)
    n = int(np.searchsorted(roff, min(int(roff[-1]), budget), side="right")) - 1
    n = max(1, min(n, len(read_len)))
    sub_off = roff[:n + 1]
    sub_buf = rbuf[:int(sub_off[-1])]
    t0 = time.perf_counter()
    co, ch, cp, cs = oracle.sketch_batch(cbuf, coff, k, w, threads=cores)
    cid = np.repeat(np.arange(len(co) - 1, dtype=np.uint32), np.diff(co).astype(np.int64))
    ix = oracle.Index(ch, cid, cp, cs)
    ro, rh, rp, rs = oracle.sketch_batch(sub_buf, sub_off, k, w, threads=cores)
    res = oracle.map_reads(ix, ctg_len, ro, read_len[:n], rh, rp, rs, k=params["k"], z=params["z"], x=params["x"],
                           sensitive=params["sensitive"], repeat_filter=params["repeat_filter"], threads=cores)
    dt = time.perf_counter() - t0
    bases = int(sub_off[-1])
    return {"value": round(bases / dt / 1e9, 4), "unit": "Gbases/s", "cores": cores, "kind": "port",
            "sample": f"all {len(ctg_len)} contigs + first {n} reads ({bases} bases) of the same workload; "
                      f"sketch with {cores} OpenMP threads (indexlr -t), map read-parallel; {dt:.1f} s",
            "mappings": int(len(res["maps"]))}


if __name__ == "__main__":
    main()
