#!/bin/bash
# Round 5: settings sweeps of how the two streams share the device (tools/share_sweep.py).  usage: tools/gpu_r5b.sh <tag>
set -x
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
[ -f "$R/bench.py" ] || { echo "no bench.py under $R"; exit 1; }
cd "$R"
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
timeout 1200 python tools/share_sweep.py --workload C3 --steps 3 '' 'NTL_SKW_RETRY=0' '' 'NTL_SKW_RETRY=0' 'NTL_EMIT_WGS_PER_CU=3' 'NTL_EMIT_WGS_PER_CU=4' 'NTL_EMIT_WGS_PER_CU=3 NTL_SKW_RETRY=0' 2>$O/sweep_C3.err | tee $O/sweep_C3.jsonl
timeout 900 python tools/share_sweep.py --workload C5 --steps 2 '' 'NTL_SKW_RETRY=0' '' 'NTL_SKW_RETRY=0' 2>>$O/sweep_C3.err | tee $O/sweep_C5.jsonl
tail -n 3 $O/sweep_C3.err 
