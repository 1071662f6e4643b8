#!/bin/bash
# Builds the padded variants of libsnailhip.so that tools/history/exp_pad.sh times (run here, on the build host; the .so files travel to the
# GPU box with the snapshot): the hand-written node loop with 20 extra instructions of one kind per node visit (SNAIL_EXP_PAD).
set -eu
cd "$(dirname "$0")/../snail_amd/csrc"
mkdir -p ../exp
FLAGS=$(grep '^FLAGS' Makefile | sed 's/^FLAGS *?= *//')
mk() { /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "-DSNAIL_EXP_PAD=\"$2\"" -shared snail_hip.hip bvh_build.cpp -o ../exp/lib_$1.so; echo "built lib_$1.so"; }
rep() { python3 -c "import sys; print(sys.argv[1] * 20)" "$1"; }
mk mulchain "$(rep ' v_mul_f32 %[t1], %[t1], %[t1]\n')"
mk mul2     "$(rep ' v_mul_f32 %[t1], %[t2], %[t4]\n')"
mk max3     "$(rep ' v_max3_f32 %[t1], %[t2], %[t4], %[t5]\n')"
mk subs     "$(rep ' v_sub_f32 %[t1], s84, %[t2]\n')"
mk mov      "$(rep ' v_mov_b32 %[t1], %[t2]\n')"
mk salu20   "$(rep ' s_add_u32 %[off], %[off], 1\n')"
mk nop20    "$(rep ' s_nop 0\n')"
mk vcmp     "$(rep ' v_cmp_le_f32 vcc, 0, %[t1]\n')"
mk rfl      "$(rep ' v_readfirstlane_b32 %[off], %[t1]\n')"
