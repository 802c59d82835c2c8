#!/bin/bash
# sketch stage alone: timing at the C3 and C2 points (fast and exact pass), then SQ counters of the C3 point
TAG=${1:-r02s}
mkdir -p gpurun_out/$TAG
python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
python tools/sketch_bench.py | tee gpurun_out/$TAG/sk_c3.json
NTL_SKETCH_FAST=0 python tools/sketch_bench.py | tee gpurun_out/$TAG/sk_c3_exact.json
python tools/sketch_bench.py --w 100 --read-len 10000 --bases 2e9 | tee gpurun_out/$TAG/sk_c2.json
NTL_SKETCH_FAST=0 python tools/sketch_bench.py --w 100 --read-len 10000 --bases 2e9 | tee gpurun_out/$TAG/sk_c2_exact.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -o p -- python3 $R/tools/sketch_bench.py --reps 2 > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/$TAG/pmc_sq2 -o p -- python3 $R/tools/sketch_bench.py --reps 2 > /dev/null 2> $R/gpurun_out/$TAG/pmc_sq2.err
cd $R
python3 - <<PY
import csv, collections
for d in ("pmc_sq", "pmc_sq2"):
    try:
        rows = list(csv.DictReader(open("gpurun_out/$TAG/%s/p_counter_collection.csv" % d)))
    except Exception as e:
        print(d, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "sketch" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {c: round(sum(x[1:]) / max(len(x) - 1, 1)) for c, x in v.items()})
PY
tail -3 gpurun_out/$TAG/pmc_sq2.err
