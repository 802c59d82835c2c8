#!/bin/bash
P='import json,sys; d=json.loads(sys.stdin.read()); print(d["config"]["stage_ms_per_step"]["sketch_mask"], d["roofline"]["avg_launch_ms"])'
for v in 0 1 2 4 6 7 3; do echo -n "ablate=$v mask_ms: "; NTL_ABLATE=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
