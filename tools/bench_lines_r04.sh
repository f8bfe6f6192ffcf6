#!/bin/bash
# The measurement lines SURVEY.md section 8(d) names beside the headline (run on the GPU box): mtv, the reference's own defaults, the drop-in surface, and
# the non-default observation rows inside the T-step launch.  Usage: tools/bench_lines_r04.sh <tag>
tag=$1; R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
B="python $R/bench.py --steps 256 --warmup 32"
$B > $out/bench.json 2> $out/bench.err
$B --distance mtv > $out/bench_mtv.json 2>> $out/bench.err
$B --defaults > $out/bench_reference_defaults.json 2>> $out/bench.err
$B --surface --steps 64 --warmup 8 > $out/bench_surface.json 2>> $out/bench.err
$B --surface --defaults --steps 64 --warmup 8 > $out/bench_surface_defaults.json 2>> $out/bench.err
$B --param is_ego_view=false --cpu-seconds 4 > $out/bench_obs_bird.json 2>> $out/bench.err
$B --param is_ego_view=false --param is_apply_mask=true --param is_obs_steering=true --cpu-seconds 4 > $out/bench_obs_bird_mask_steer.json 2>> $out/bench.err
$B --param is_obs_steering=true --param is_observe_ref_path_other_agents=true --param is_apply_mask=true --cpu-seconds 4 > $out/bench_obs_steer_ref.json 2>> $out/bench.err
$B --param is_observe_distance_to_boundaries=false --cpu-seconds 4 > $out/bench_obs_boundary_points.json 2>> $out/bench.err
$B --param is_ego_view=false --param is_partial_observation=false --cpu-seconds 4 > $out/bench_obs_full.json 2>> $out/bench.err
tail -3 $out/bench.err
