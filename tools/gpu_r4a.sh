#!/bin/bash
# round 4, A/B of the window pass: build, sketch parity tests, sketch stage alone at the C3 / C5 / w=150 points for the variants of
# NTL_SKETCH_WAVE, SQ counters of the default
TAG=${1:-r04a}
VARIANTS=${VARIANTS:-"0 1 4 16"}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketch or fast or fuzz or edge or golden or thresh" 2>&1 | tail -4 | tee gpurun_out/$TAG/pytest_sketch.log
for v in $VARIANTS; do
  NTL_SKETCH_WAVE=$v python tools/sketch_bench.py | tee -a gpurun_out/$TAG/sk_c3.jsonl
done
for v in 0 1 4; do
  NTL_SKETCH_WAVE=$v python tools/sketch_bench.py --w 100 --k 24 --read-len 20000 --bases 3.9e9 | tee -a gpurun_out/$TAG/sk_c5.jsonl
done
for v in 0 1; do
  NTL_SKETCH_WAVE=$v python tools/sketch_bench.py --w 150 --k 32 --read-len 15000 --bases 3.9e9 | tee -a gpurun_out/$TAG/sk_w150.jsonl
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -o p -- python3 $R/tools/sketch_bench.py --reps 2 > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVES --output-format csv -d $R/gpurun_out/$TAG/pmc_sq2 -o p -- python3 $R/tools/sketch_bench.py --reps 2 > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq2.err
cd $R
python3 - <<PY | tee gpurun_out/$TAG/pmc_summary.txt
import csv, collections
for d in ("pmc_sq", "pmc_sq2"):
    try:
        rows = list(csv.DictReader(open("gpurun_out/$TAG/%s/p_counter_collection.csv" % d)))
    except OSError as e:
        print(d, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "sketch_" in r["Kernel_Name"] and ("wave" in r["Kernel_Name"] or "thresh" in r["Kernel_Name"] or "fast_list" in r["Kernel_Name"]):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(d, k, {c: round(sum(x[1:]) / max(len(x) - 1, 1)) for c, x in v.items()})
PY
