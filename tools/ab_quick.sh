#!/bin/bash
# A/B of two trees on the same GPU box: bench lines + the SQ instruction counters of the step kernel.  Usage: tools/ab_quick.sh <outdir>
out=$GRAFT_REPO_ROOT/$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pmc() { tree=$1; name=$2; shift 2; (cd "$tree" && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$out/$name" -o "$name" -- python bench.py "$@" > "$out/$name.json" 2> "$out/$name.err"); }
(cd $GRAFT_REPO_ROOT/_ab/old && python bench.py --cpu-seconds 0 --steps 256 --warmup 32 > $out/old_256.json 2>/dev/null; python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > $out/old_20.json 2>/dev/null)
(cd $GRAFT_REPO_ROOT && python bench.py --cpu-seconds 0 --steps 256 --warmup 32 > $out/new_256.json 2>/dev/null; python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > $out/new_20.json 2>/dev/null)
pmc $GRAFT_REPO_ROOT/_ab/old pmc_old --cpu-seconds 0 --steps 32 --warmup 8 --no-one-stream --streams 1
pmc $GRAFT_REPO_ROOT pmc_new1 --cpu-seconds 0 --steps 32 --warmup 8 --no-compare --streams 1 --chunk 1
pmc $GRAFT_REPO_ROOT pmc_newT --cpu-seconds 0 --steps 32 --warmup 32 --no-compare
cd $GRAFT_REPO_ROOT
for n in pmc_old pmc_new1 pmc_newT; do echo "== $n"; python tools/pmc_summary.py $out/$n step_wave; done > $out/pmc_summary.txt
