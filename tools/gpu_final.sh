#!/bin/bash
# Round-end visit: full round (tests, smoke, bench, rocprof trace + PMC passes) and the end-to-end runs.
TAG=${1:-r01g}
bash tools/gpu_round.sh $TAG
bash tools/gpu_e2e.sh
cp gpurun_out/e2e/e2e.jsonl gpurun_out/$TAG/e2e.jsonl
