#!/bin/bash
mkdir -p gpurun_out/calib
hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o /tmp/valu_calib || exit 1
/tmp/valu_calib | tee gpurun_out/calib/run.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/calib/pmc -o p -- /tmp/valu_calib > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/calib/pmc.err
cd $GRAFT_REPO_ROOT
/opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading /tmp/valu_calib 2>/dev/null | head -5
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open("gpurun_out/calib/pmc/p_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, {c:f"{max(x):.4e}" for c,x in v.items()})
PY
