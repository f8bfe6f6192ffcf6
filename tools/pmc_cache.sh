#!/bin/bash
# L2 / L1 hit-rate counters for the bench command.  Usage: tools/pmc_cache.sh <outdir> <bench args...>
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_ARGS=("$@")
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o "$name" -- python bench.py "${BENCH_ARGS[@]}" > "$out/$name.json" 2> "$out/$name.err"; }
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcc2 TCC_WRITE_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
