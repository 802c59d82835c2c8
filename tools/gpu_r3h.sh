#!/bin/bash
# emit kernel with two minimizers in flight per thread (NTL_EMIT_U=2) against one; pair-driver timeline
TAG=${1:-r03h}
O=gpurun_out/$TAG; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
NTL_EMIT_U=2 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-others "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    j=[json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"metric')][-1]; c=j["config"]
    print(sys.argv[2], j["value"], "Gbases/s", j["ms_per_step"], "ms", "SERIAL", c.get("serial_pass",{}).get("ms_per_step"), c.get("serial_pass",{}).get("stage_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  tail -2 $O/bench_$name.err | cut -c1-300; }
run c3_u1 NTL_EMIT_U=1 -- --steps 6 --warmup 1
run c3_u2 NTL_EMIT_U=2 -- --steps 6 --warmup 1
run c5_u1 NTL_EMIT_U=1 -- --workload C5 --steps 3 --warmup 1
run c5_u2 NTL_EMIT_U=2 -- --workload C5 --steps 3 --warmup 1
run c2_u1 NTL_EMIT_U=1 -- --workload C2 --steps 40 --warmup 3
run c2_u2 NTL_EMIT_U=2 -- --workload C2 --steps 40 --warmup 3
NTL_E2E_TRACE_ONLY=$PWD/$O/pipe_trace timeout 900 python tools/e2e_diag.py --bases 16e9 > $O/e2e_trace.jsonl 2> $O/e2e_trace.err
cut -c1-400 $O/e2e_trace.jsonl; ls $O
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/valu_calib3.hip -o /tmp/valu_calib3 && timeout 300 /tmp/valu_calib3 20000 | tee $O/valu_calib3.txt
