#!/bin/bash
# Round 5: settings sweeps of how the two streams share the device (tools/share_sweep.py).  usage: tools/gpu_r5b.sh <tag>
set -x
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for P in 2 0; do
NTL_PIPELINE_PRIO=$P timeout 1200 python tools/share_sweep.py --workload C3 --steps 3 '' 'NTL_EMIT_WGS_PER_CU=4' 'NTL_EMIT_WGS_PER_CU=6' 'NTL_EMIT_WGS_PER_CU=8' 'NTL_EMIT_WGS_PER_CU=0' 'NTL_EMIT_WGS_PER_CU=4 NTL_SKW_WGS_PER_CU=2' 'NTL_EMIT_WGS_PER_CU=0 NTL_SKW_WGS_PER_CU=2' 2>$O/sweep_C3.err | sed "s/^{/{\"prio\": $P, /" | tee -a $O/sweep_C3.jsonl
done
tail -n 3 $O/sweep_C3.err 
