#!/bin/bash
# Round-4 profile set of the policy-in-the-loop path (run on the GPU box): the fp32 actor in both arithmetic modes and the bf16 variant in front of the fused step.
# Usage: tools/profile_r04_policy.sh <tag>   (outputs under gpurun_out/<tag>; the summaries are then copied into profiles/r04_policy_*)
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag; mkdir -p $out
B="python $R/bench.py"
cd $R
$B --policy --cpu-seconds 8 > $out/bench_policy_fp32_split.json 2> $out/bench.err
$B --policy --policy-mode exact --cpu-seconds 0 > $out/bench_policy_fp32_exact.json 2>> $out/bench.err
$B --policy --policy-precision bf16 --cpu-seconds 0 > $out/bench_policy_bf16.json 2>> $out/bench.err
$B --policy --streams 2 --cpu-seconds 0 > $out/bench_policy_fp32_split_two_streams.json 2>> $out/bench.err
MODE=split python tools/actor_timing.py > $out/actor_timing_split.txt 2>&1
MODE=exact python tools/actor_timing.py > $out/actor_timing_exact.txt 2>&1
python tools/fuzz_mlp32.py --cases 60 > $out/fuzz_mlp32.txt 2>&1
python tools/mlp_phase_cycles.py > $out/mlp32s_phase_cycles.txt 2>&1
tools/mlp32_profile.sh gpurun_out/$tag/mlp32s > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_policy -o t -- $B --policy --steps 256 --warmup 32 --cpu-seconds 0 --no-compare > $out/policy_bench_under_rocprof.json 2> $out/trace_policy.err
cp $(ls $out/trace_policy/*kernel_stats.csv | head -1) $out/policy_kernel_stats.csv
ls $out
