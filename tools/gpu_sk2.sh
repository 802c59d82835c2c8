#!/bin/bash
# quick A/B of the sketch stage: build, sketch parity tests, timing at the C3 and C2 points, VALU/SALU/LDS counters at C3
TAG=${1:-r02s}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sketch or fast or fuzz or edge or golden" 2>&1 | tail -4 | tee gpurun_out/$TAG/pytest_sketch.log
python tools/sketch_bench.py | tee gpurun_out/$TAG/sk_c3.json
python tools/sketch_bench.py --w 100 --read-len 10000 --bases 2e9 | tee gpurun_out/$TAG/sk_c2.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -o p -- python3 $R/tools/sketch_bench.py --reps 2 > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq.err
cd $R
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/$TAG/pmc_sq/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "sketch_fast" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {c: round(sum(x[1:]) / max(len(x) - 1, 1)) for c, x in v.items()})
PY
