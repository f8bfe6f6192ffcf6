#!/bin/bash
# A/B of several builds of the HIP library on ONE GPU box (box-to-box noise is ~1.5 %): every library named on the command line is benchmarked in
# turn, REPS rounds interleaved, through SIGMAENV_LIB (capi.py: an alternative build of the same HIP library, never a fallback).
# Usage: tools/ab_libs.sh <tag> <lib.so> [<lib.so> ...]   (paths relative to sigmarl_amd/csrc; REPS=3, STEPS=256 by default)
tag=$1; shift
export SIGMAENV_ALLOW_STALE=1 SIGMAENV_LIB_OLD_ABI=1  # (kept older builds are the point of an A/B)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
REPS=${REPS:-3}; STEPS=${STEPS:-256}
for r in $(seq 1 $REPS); do
  for lib in "$@"; do
    SIGMAENV_LIB=$R/sigmarl_amd/csrc/$lib python $R/bench.py --cpu-seconds 0 --steps $STEPS --warmup 32 $BENCH_EXTRA 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', $r, '%.4e' % d.get('value_sustained', d['value']), '%.5f' % d['ms_per_step'], '%.5f' % d['config'].get('per_step_launch',{}).get('ms_per_step',0))" | tee -a $out/ab.txt
  done
done
python - <<PY
import collections
v=collections.defaultdict(list)
for l in open("$out/ab.txt"):
    p=l.split(); v[p[0]].append(float(p[2]))
for k,x in v.items(): print(f"{k:40s} best {max(x):.4e} mean {sum(x)/len(x):.4e}")
PY
