#!/bin/bash
# phase ablation of sketch_fast_kernel (NTL_SKETCH_ABLATE bits: 1 search, 2 window pass, 4 rolling, 8 init): time + VALU instructions
TAG=${1:-r02t}
mkdir -p gpurun_out/$TAG
NTL_EXTRA_HIPCC_FLAGS=-DNTL_SKETCH_ABLATION python __graft_entry__.py > gpurun_out/$TAG/build.log 2>&1 || { tail -20 gpurun_out/$TAG/build.log; exit 1; }
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ab in 0 1 3 7 15; do
  export NTL_SKETCH_ABLATE=$ab
  python3 $R/tools/sketch_bench.py $SKARGS | tee -a $R/gpurun_out/$TAG/ablate.jsonl | cut -c1-230
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/$TAG/ab$ab -o p -- python3 $R/tools/sketch_bench.py --reps 2 $SKARGS > /dev/null 2> /dev/null
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("$R/gpurun_out/$TAG/ab$ab/p_counter_collection.csv")):
    if "sketch_fast" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("ablate=$ab", {c: round(sum(x[1:]) / max(len(x) - 1, 1) / 1e6) for c, x in agg.items()}, "(millions per launch)")
PY
done
