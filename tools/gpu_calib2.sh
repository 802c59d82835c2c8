#!/bin/bash
# Per-instruction VALU/LDS issue calibration (tools/valu_calib2.hip) -> gpurun_out/<tag>/valu_calib2.txt
TAG=${1:-r02a}
mkdir -p gpurun_out/$TAG
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/valu_calib2.hip -o /tmp/valu_calib2 || exit 1
timeout 600 /tmp/valu_calib2 ${2:-20000} | tee gpurun_out/$TAG/valu_calib2.txt
