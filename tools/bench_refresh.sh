#!/bin/bash
# The bench lines of a profile set again, AFTER its PMC summaries (profiles/valu_*.json, traffic_latest.json) have been collected: bench.py reads those files for the
# issue-side / traffic figures of its JSON line, so the lines written during tools/profile_rNN.sh carry the PREVIOUS set's counters.  Usage: tools/bench_refresh.sh <tag>
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
B="python $R/bench.py"
$B --cpu-seconds 12 > $out/bench.json 2> $out/bench.err
$B --steps 20 --warmup 5 > $out/bench_driver_style.json 2> /dev/null
$B --cbf-qp --steps 64 --warmup 16 --cpu-seconds 10 > $out/bench_cbf_qp.json 2> /dev/null
$B --cbf --steps 64 --warmup 16 --cpu-seconds 0 > $out/bench_cbf.json 2> /dev/null
cd $R; tools/bench_lines_r04.sh $tag/lines > /dev/null 2>&1
ls $out $out/lines | head -40
