#!/bin/bash
# Diagnostic (profile build): SQ instruction counts of the step kernel with one phase ablated at a time (results of such runs are invalid;
# only the counters are read; an ablated phase can leave garbage that sends later phases into very long loops: every run has its own timeout).
# Usage: tools/pmc_ablation.sh <outdir>
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SIGMAENV_LIB=$GRAFT_REPO_ROOT/sigmarl_amd/csrc/libsigmaenv_prof.so
for skip in 0 1 2 4 8 16 32 64 127; do
  SIGMAENV_DEBUG_SKIP=$skip timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d "$out/s$skip" -o s$skip -- python bench.py --steps 16 --warmup 4 --cpu-seconds 0 --streams 1 > /dev/null 2> "$out/s$skip.err"
  echo "skip=$skip" >> "$out/summary.txt"
  python tools/pmc_summary.py "$out/s$skip" step_wave | grep -E "SQ_" >> "$out/summary.txt"
done
cat "$out/summary.txt"
