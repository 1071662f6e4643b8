#!/bin/bash
# round 4, VERDICT item 9: the mirrored packets' kernel k_rays<false,true> at different register budgets, WITH the staged pipeline's
# dispatch-order feedback in place.  Variants (tools/variant.sh; product sources + flags):
#   ship      95 VGPRs, 5 waves / SIMD: one- and two-rays-per-lane narrow leaves
#   rays85    -DSNAIL_PERRAY_NARROW2=0: one-ray tier only, 85 VGPRs (still 5 waves: the step is at 80)
#   rays80    the same under amdgpu_waves_per_eu(6): 80 VGPRs, 6 spilled, 6 waves
#   rays80n2  both tiers under the six-wave budget
# per variant: bench.py --config 3 --reflections (throughput, lone frame) twice, and the SQ counter passes of six lone frames.
set -u
O=gpurun_out/r4rays; mkdir -p $O
export TMPDIR=/tmp
for v in ship rays85 rays80 rays80n2; do
  if [ $v = ship ]; then unset SNAIL_LIB_PATH; else export SNAIL_LIB_PATH=$PWD/snail_amd/exp/lib_$v.so; fi
  for i in 1 2; do timeout -k 10 300 python bench.py --config 3 --reflections --steps 300 --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err || echo "bench $v FAILED"; done
  OUT=$PWD/gpurun_out/pmc_rays_$v; mkdir -p $OUT
  for g in sq1 sq2; do
    if [ $g = sq1 ]; then C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; else C="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; fi
    ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$g -- python3 $GRAFT_REPO_ROOT/tools/whitted_once.py atrium 1 refl > $OUT/$g.log 2>&1 ); echo "$v $g rc=$?"
  done
  python tools/pmc_summary.py $OUT "k_rays<" > $O/pmc_$v.txt; find $OUT -name "*.csv" -size +200k -delete 2>/dev/null
done
python - <<'PY'
import json, glob, re
O = 'gpurun_out/r4rays'
out = []
for v in ('ship', 'rays85', 'rays80', 'rays80n2'):
    vals = []
    for f in sorted(glob.glob('%s/bench_%s_*.json' % (O, v))):
        try:
            d = json.loads([l for l in open(f) if l.startswith('{')][-1]); vals.append((d['value'], d['ms_per_step'], d['roofline']['lone_frame_ms'], d.get('verified')))
        except Exception as e: vals.append(('ERR', str(e)))
    c = {}
    for line in open('%s/pmc_%s.txt' % (O, v)):
        m = re.match(r'\S+\s+(\S+)\s+n=\d+\s+mean=(\S+)', line)
        if m: c[m.group(1)] = float(m.group(2))
    lanes = c.get('SQ_THREAD_CYCLES_VALU', 0) / c['SQ_INSTS_VALU'] if c.get('SQ_INSTS_VALU') else None
    wait = c.get('SQ_WAIT_INST_ANY', 0) / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else None
    out.append('%-9s bench (Mrays/s, ms/frame, lone ms, verified): %s | k_rays<false,true>: VALU %.1f M, SALU %.1f M, lanes live %.1f, wait share of wave cycles %.3f' % (
        v, vals, c.get('SQ_INSTS_VALU', 0) / 1e6, c.get('SQ_INSTS_SALU', 0) / 1e6, lanes or 0, wait or 0))
open(O + '/summary.txt', 'w').write('\n'.join(out) + '\n'); print('\n'.join(out))
PY
