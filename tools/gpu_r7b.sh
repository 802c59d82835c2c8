set -x
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r07b; mkdir -p $O
python __graft_entry__.py > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for V in default wpe5 wpe6 wpe8; do
  unset NTLINK_AMD_LIB
  [ $V != default ] && export NTLINK_AMD_LIB=$PWD/ntlink_amd/build/var_$V/libntlink_hip.so
  echo "== $V"; timeout 300 python tools/small_window_bench.py 15,5 20,10 20,15 15,3 2> $O/swb_$V.err | grep '"sketch_small_kernel": true' | tee $O/swb_$V.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['k'], d['w'], d['Gbases_per_s'], d['window_pass_Gbases_per_s'], d['stage_ms'])
"
done
unset NTLINK_AMD_LIB
for E in "" "NTL_EMIT_WGS_PER_CU=3" "NTL_EMIT_WGS_PER_CU=4" "NTL_EMIT_WGS_PER_CU=0" "NTL_SKW_WGS_PER_CU=2" "NTL_SKW_WGS_PER_CU=4 NTL_EMIT_WGS_PER_CU=3" "NTL_SKETCH_WAVE=8"; do
  echo "== C2 [$E]"
  env $E timeout 300 python bench.py --workload C2 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-others 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']
print(d['value'], 'Gbases/s', d['ms_per_step'], 'ms/step', c['stage_ms_per_step'])
"
done
