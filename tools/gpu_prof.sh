#!/bin/bash
# Full-size bench line + rocprofv3 kernel trace of the same command.  Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out/prof
python __graft_entry__.py > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 3000 gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name '*stats*' | head; 
for f in $(find gpurun_out/prof -name '*kernel_stats.csv'); do head -30 $f; done
# keep the big trace out of the merge
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
