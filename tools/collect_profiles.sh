#!/bin/bash
# Copies the summaries of a tools/profile_rNN.sh run from gpurun_out/<tag> into profiles/ under the round's prefix.  Usage: tools/collect_profiles.sh <tag> <prefix, e.g. r05>
tag=$1; pre=$2; src=gpurun_out/$tag
for f in bench.json bench_driver_style.json bench_sweep.json bench_emulate_ranks8.json bench_config4.json bench_cpm_32x8192.json bench_cbf_qp.json phase_cycles.txt phase_cycles_obs_bird.txt \
         qp_phase_cycles.txt pmc_step_kernel.txt pmc_config4_step_kernel.txt pmc_cbf_qp_kernel.txt pmc_cbf_kernel.txt bench_cbf.json traffic_config4.json; do
  [ -f $src/$f ] && cp $src/$f profiles/${pre}_$f
done
for n in head config4 cbf_qp cbf mtv obs_bird; do
  [ -f $src/${n}_kernel_stats.csv ] && cp $src/${n}_kernel_stats.csv profiles/${pre}_${n}_kernel_stats.csv
  [ -f $src/${n}_bench_under_rocprof.json ] && cp $src/${n}_bench_under_rocprof.json profiles/${pre}_${n}_bench_under_rocprof.json
done
for f in $src/lines/*.json; do [ -f $f ] && cp $f profiles/${pre}_lines_$(basename $f); done
[ -f $src/traffic_latest.json ] && cp $src/traffic_latest.json profiles/traffic_latest.json
[ -f $src/valu_latest.json ] && cp $src/valu_latest.json profiles/valu_latest.json
[ -f $src/valu_dominant_latest.json ] && cp $src/valu_dominant_latest.json profiles/valu_dominant_latest.json
[ -f $src/valu_cbf_margin_latest.json ] && cp $src/valu_cbf_margin_latest.json profiles/valu_cbf_margin_latest.json
ls profiles | grep "^${pre}_" | wc -l
