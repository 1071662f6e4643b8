#!/bin/bash
# tools/variant.sh NAME [-DFLAG ...]: build snail_amd/exp/lib_NAME.so = the product sources with extra preprocessor flags (experiment
# hooks of snail_hip.hip / snail_dev.inc).  Run on the build host; the .so travels to the GPU box with the snapshot and is loaded with
# SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_NAME.so (snail_amd/_lib.py) -- the product library is never overwritten.
set -eu
name=$1; shift
cd "$(dirname "$0")/../snail_amd/csrc"
mkdir -p ../exp
FLAGS=$(grep '^FLAGS' Makefile | sed 's/^FLAGS *?= *//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -shared snail_hip.hip bvh_build.cpp host_sse.cpp -o ../exp/lib_$name.so
echo "built snail_amd/exp/lib_$name.so ($*)"
