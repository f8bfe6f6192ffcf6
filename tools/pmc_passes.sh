#!/bin/bash
# Collects PMC counters for the bench command in separate rocprofv3 passes (counters only, no tracing domains), as the
# MI355X guide prescribes.  Usage: tools/pmc_passes.sh <outdir> <bench args...>
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_ARGS=("$@")
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- python bench.py "${BENCH_ARGS[@]}" > "$out/$name.json" 2> "$out/$name.err"; }
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
pass sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
pass sq4 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VSKIPPED
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
