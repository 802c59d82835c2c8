#!/bin/bash
# kernel trace of the pipelined bench (2 steps) for a timeline: gpurun_out/<tag>/trace_<W>/..kernel_trace.csv (trimmed to the columns a timeline needs)
# usage: [ENVS="A=1"] tools/gpu_trace.sh <tag> [workload, default C3]
TAG=${1:-r06t}; W=${2:-C3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; O=$R/gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for kv in $ENVS; do export $kv; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$W -o kt -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-others --serial-steps 0 --workload $W > $O/bench_trace_$W.json 2> $O/trace_$W.err
F=$(find $O/trace_$W -name '*kernel_trace.csv' | head -1)
python3 - "$F" "$O/timeline_$W.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# the last ~40 % of the run: the timed steps
with open(sys.argv[2], "w") as fh:
    fh.write("start_us,end_us,queue,kernel\n")
    for r in rows[int(len(rows) * 0.55):]:
        nm = r["Kernel_Name"].split("(")[0][:60]
        fh.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:.1f},{(int(r["End_Timestamp"]) - t0) / 1e3:.1f},{r.get("Queue_Id", "")},{nm}\n')
print(len(rows), "dispatches")
PY
find $O -name '*kernel_trace.csv' -delete
ls -la $O
