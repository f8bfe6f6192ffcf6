#!/bin/bash
# PMC passes (counters only) over tools/qp_once.py.  Usage: tools/pmc_qp.sh <outdir under gpurun_out> [kernel substring]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- python tools/qp_once.py > "$out/$name.log" 2> "$out/$name.err"; }
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
[ -n "$PMC_MORE" ] && pass sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
[ -n "$PMC_MORE" ] && pass sq4 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VSKIPPED
python tools/pmc_summary.py "$out" "${2:-cbf}" | tee "$out/summary.txt"
