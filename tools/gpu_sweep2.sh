mkdir -p gpurun_out/r04ap
T=NTL_SKETCH_THRESH
timeout 400 python tools/share_sweep.py --workload C3 --steps 4 "" "$T=7" "$T=8" "$T=9" "$T=9.5" "$T=10.5" "$T=11" "$T=12" "" > gpurun_out/r04ap/thresh_C3.jsonl 2> gpurun_out/r04ap/thresh_C3.err
cut -c1-330 gpurun_out/r04ap/thresh_C3.jsonl
timeout 400 python tools/share_sweep.py --workload C5 --steps 2 "" "$T=7" "$T=8" "$T=9" "$T=11" "$T=12" "" > gpurun_out/r04ap/thresh_C5.jsonl 2> gpurun_out/r04ap/thresh_C5.err
cut -c1-330 gpurun_out/r04ap/thresh_C5.jsonl
tail -2 gpurun_out/r04ap/thresh_C5.err
