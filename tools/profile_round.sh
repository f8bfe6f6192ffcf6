#!/bin/bash
# The round's profile set for the headline workload (run on the GPU box; outputs under gpurun_out/<tag>, copy into profiles/).
# Usage: tools/profile_round.sh <tag> [bench args...]      e.g. tools/profile_round.sh r02_c --streams 1
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py --steps 256 --warmup 32 --cpu-seconds 12 "$@" > $out/${tag}_bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python bench.py --steps 256 --warmup 32 --cpu-seconds 0 --no-one-stream "$@" > $out/${tag}_bench_under_rocprof.json 2> $out/trace.err
db=$(ls $out/trace/*.db 2>/dev/null | head -1)
if [ -n "$db" ]; then python tools/rocpd_summary.py "$db" $out/${tag}_kernel_stats.csv > /dev/null; else cp $(ls $out/trace/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv; fi
bash tools/pmc_passes.sh $out/pmc --steps 32 --warmup 8 --cpu-seconds 0 --no-one-stream "$@" > /dev/null 2>&1
python tools/pmc_summary.py $out/pmc step_wave > $out/${tag}_pmc_step_kernel.txt
