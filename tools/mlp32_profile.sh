#!/bin/bash
# Kernel trace + matrix-core counters of the exact-fp32 actor kernel (run on the GPU box).  Usage: tools/mlp32_profile.sh <outdir>
out=$GRAFT_REPO_ROOT/$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- python $GRAFT_REPO_ROOT/tools/actor_timing.py > "$out/actor_timing.txt" 2> "$out/trace.err"
cp $(ls "$out"/trace/*kernel_stats.csv | head -1) "$out/kernel_stats.csv"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d "$out/pmc1" -o p -- python $GRAFT_REPO_ROOT/tools/actor_timing.py > /dev/null 2> "$out/pmc1.err"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d "$out/pmc2" -o p -- python $GRAFT_REPO_ROOT/tools/actor_timing.py > /dev/null 2> "$out/pmc2.err"
cd $GRAFT_REPO_ROOT
{ python tools/pmc_summary.py "$out/pmc1" mlp32; python tools/pmc_summary.py "$out/pmc2" mlp32; } > "$out/pmc_mlp32.txt"
