#!/bin/bash
# two quick SQ passes for the step kernel.  Usage: tools/pmc_quick.sh <outdir> <bench args...>
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_ARGS=("$@")
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- python bench.py "${BENCH_ARGS[@]}" > "$out/$name.json" 2> "$out/$name.err"; }
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
pass sq5 SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
