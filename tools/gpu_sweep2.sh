mkdir -p gpurun_out/r04aj
E=NTL_EMIT_WGS_PER_CU; B=NTL_SKW_BUDGET; G=NTL_SKW_WGS_PER_CU
timeout 600 python tools/share_sweep.py --workload C5 --steps 3 "" \
 "$E=3" "$E=3 $B=12" "$E=3 $B=24" "$E=4 $B=12" "$E=3 $G=2" "$E=3 $G=4" "$E=3 $G=5" "$E=3 NTL_SKETCH_WAVE=8 $G=1" "$E=3 NTL_SKETCH_WAVE=8 $G=2" \
 "$E=5" "$E=3 NTL_EMIT_U=2" "$E=2 $G=4" "$E=4 $G=4" "$E=3" "" > gpurun_out/r04aj/sweep_C5.jsonl 2> gpurun_out/r04aj/sweep_C5.err
cut -c1-330 gpurun_out/r04aj/sweep_C5.jsonl
tail -2 gpurun_out/r04aj/sweep_C5.err
