#!/bin/bash
# The two shortest soaks that showed round 3's box-dependent mismatches (mirrored / masked per-ray-origin packets), with the differences printed:
# run first in a GPU call; output under gpurun_out/canary_*.txt
mkdir -p gpurun_out
timeout -k 10 200 python tests/soak_rays.py 3000 ${1:-77} perray1 > gpurun_out/canary_rays.txt 2>&1
timeout -k 10 300 python tests/soak_fuzz.py 3000 ${1:-77} refl > gpurun_out/canary_fuzz.txt 2>&1
rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | head -1 >> gpurun_out/canary_boxes.txt
echo "canary: rays $(grep -c MISMATCH gpurun_out/canary_rays.txt) fuzz $(grep -c MISMATCH gpurun_out/canary_fuzz.txt) mismatches"
