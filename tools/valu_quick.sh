#!/bin/bash
# One SQ pass of the headline's 32-step launches: VALU / SALU instructions per launch of the step kernel (run on the GPU box).  Usage: tools/valu_quick.sh <tag> [bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag; mkdir -p $out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc/sq1 -o p -- python $R/bench.py --steps 32 --warmup 32 --cpu-seconds 0 --no-compare "$@" > /dev/null 2> $out/pmc.err
cd $R
python tools/pmc_summary.py $out/pmc step_wave | tee $out/pmc_step_kernel.txt
