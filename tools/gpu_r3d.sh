#!/bin/bash
TAG=${1:-r03d}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
NTL_POOL_TRACE=1 NTL_BENCH_TRACE=1 NTL_SKETCH_LANES=0 timeout 600 python bench.py --no-cpu-baseline --no-e2e --no-others --workload C5 --steps 2 --warmup 1 --serial-steps 0 > $O/c5.json 2> $O/c5.err
echo "mallocs $(grep -c hipMalloc $O/c5.err) frees $(grep -c hipFree $O/c5.err)"; grep "hipMalloc" $O/c5.err | awk '{for(i=1;i<=NF;i++) if($i=="in") s+=$(i+1)} END {print "malloc ms total", s}'; grep "bench trace" $O/c5.err | tail -60 | head -24; grep "ntl pool" $O/c5.err | tail -12
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]; c=j["config"]
    print(sys.argv[2], j["value"], "Gbases/s", j["ms_per_step"], "ms", c["stage_ms_per_step"], "SERIAL", c.get("serial_pass",{}).get("ms_per_step"), c.get("serial_pass",{}).get("stage_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  tail -2 $O/bench_$name.err | cut -c1-300; }
run c5 X=1 -- --workload C5 --steps 3 --warmup 1
run c2 X=1 -- --workload C2 --steps 50 --warmup 3
run c3_lanes0 NTL_SKETCH_LANES=0 -- --steps 6 --warmup 1
run c3_lanes1 NTL_SKETCH_LANES=1 -- --steps 6 --warmup 1
