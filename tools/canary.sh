#!/bin/bash
# The two soaks narrowed to mirrored / masked per-ray-origin packets (the packets a work-in-progress library once got wrong: profiles/r3_final_soak.txt),
# with the differences printed:
# run first in a GPU call; output under gpurun_out/canary_*.txt
mkdir -p gpurun_out
timeout -k 10 200 python tests/soak_rays.py 3000 ${1:-77} perray1 > gpurun_out/canary_rays.txt 2>&1
timeout -k 10 300 python tests/soak_fuzz.py 3000 ${1:-77} refl > gpurun_out/canary_fuzz.txt 2>&1
rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | head -1 >> gpurun_out/canary_boxes.txt
# the same two runs on the variant with the forced six-wave budget of the generic-packet kernels (spills -> scratch), when it has been built
# (tools/variant.sh rays6 -DSNAIL_RAYS_WAVES=6): the same packets through kernels that spill
if [ -f snail_amd/exp/lib_rays6.so ]; then
  SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_rays6.so timeout -k 10 200 python tests/soak_rays.py 3000 ${1:-77} perray1 > gpurun_out/canary_rays6_rays.txt 2>&1
  SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_rays6.so timeout -k 10 300 python tests/soak_fuzz.py 3000 ${1:-77} refl > gpurun_out/canary_rays6_fuzz.txt 2>&1
  echo "canary (six-wave variant): rays $(grep -c MISMATCH gpurun_out/canary_rays6_rays.txt) fuzz $(grep -c MISMATCH gpurun_out/canary_rays6_fuzz.txt) mismatches"
fi
echo "canary: rays $(grep -c MISMATCH gpurun_out/canary_rays.txt) fuzz $(grep -c MISMATCH gpurun_out/canary_fuzz.txt) mismatches"
