for wg in 1 2 3; do
NTL_SKW_WGS_PER_CU=$wg python bench.py --no-e2e --no-cpu-baseline --no-others --workload C5 --steps 3 --serial-steps 0 > gpurun_out/r04u/bench_C5_$wg.json 2> gpurun_out/r04u/bench_C5_$wg.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04u/bench_C5_$wg.json") if l.startswith("{")][-1])
print("C5 wgs", $wg, d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"])
PY
done
NTL_SKETCH_WAVE=4 NTL_SKW_WGS_PER_CU=3 python bench.py --no-e2e --no-cpu-baseline --no-others --workload C5 --steps 3 --serial-steps 0 > gpurun_out/r04u/bench_C5_w4_3.json 2>/dev/null
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04u/bench_C5_w4_3.json") if l.startswith("{")][-1])
print("C5 wave4 wgs 3 (12 waves)", d["value"], d["ms_per_step"], d["config"]["stage_ms_per_step"])
PY
