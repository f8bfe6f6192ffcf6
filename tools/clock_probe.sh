#!/bin/bash
# Shader-clock estimate under load: GRBM_GUI_ACTIVE (GPU cycles while busy) and the matrix pipe's busy cycles per launch of the fp32 MLP and of the step kernel, beside
# the launch durations of a kernel trace of the same commands (run on the GPU box).  Usage: tools/clock_probe.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for w in policy head; do
  if [ $w = policy ]; then A="--policy --streams 1 --steps 32 --warmup 8 --cpu-seconds 0 --no-compare"; K=mlp32; else A="--steps 32 --warmup 32 --cpu-seconds 0 --no-compare"; K=step_wave; fi
  mkdir -p $out/$w
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES --output-format csv -d $out/$w/sq1 -o p -- python $R/bench.py $A > /dev/null 2> $out/$w.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${w}_trace -o t -- python $R/bench.py $A > /dev/null 2>> $out/$w.err
  (cd $R; echo "== $w"; python tools/pmc_summary.py $out/$w $K; grep -i "$K" $(ls $out/${w}_trace/*kernel_stats.csv | head -1)) | tee -a $out/clock_probe.txt
done
