#!/bin/bash
# Round-6 profile set (the passes of rounds 4 and 5, comparable file by file, plus the wait-counter and margin-kernel passes of the CBF kernels) (run on the GPU box; outputs under gpurun_out/<tag>, the summaries are then copied into profiles/).
# Usage: tools/profile_r06.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py"
trace() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -o t -- $B "$@" > $out/${name}_bench_under_rocprof.json 2> $out/trace_$name.err; cp $(ls $out/trace_$name/*kernel_stats.csv | head -1) $out/${name}_kernel_stats.csv; }
pmc() { name=$1; shift; ctrs=$1; shift; rocprofv3 --pmc $ctrs --output-format csv -d $out/pmc_$name -o p -- $B "$@" > /dev/null 2> $out/pmc_$name.err; }
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"
SQ3="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
SQ4="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VSKIPPED"
# ---- headline (BASELINE config 2): plain lines, kernel trace, PMC passes (launches of 32 steps each), phase stamps
$B --cpu-seconds 12 > $out/bench.json 2> $out/bench.err
$B --steps 20 --warmup 5 > $out/bench_driver_style.json 2> /dev/null   # the driver's own command line (incl. config.lines and cpu_baseline)
$B --cpu-seconds 0 --no-compare --sweep > $out/bench_sweep.json 2> /dev/null
$B --cpu-seconds 0 --no-compare --force-dist --emulate-ranks 8 > $out/bench_emulate_ranks8.json 2> /dev/null
trace head --steps 256 --warmup 32 --cpu-seconds 0 --no-compare
A="--steps 32 --warmup 32 --cpu-seconds 0 --no-compare"
pmc head_sq1 "$SQ1" $A; pmc head_sq2 "$SQ2" $A; pmc head_sq3 "$SQ3" $A; pmc head_sq4 "$SQ4" $A; pmc head_fetch FETCH_SIZE $A; pmc head_write WRITE_SIZE $A
mkdir -p $out/pmc_head; for d in sq1 sq2 sq3 sq4 fetch write; do cp -r $out/pmc_head_$d $out/pmc_head/$d; done
cd $R
python tools/pmc_summary.py $out/pmc_head step_wave > $out/pmc_step_kernel.txt
STEPS_PER_LAUNCH=32 python tools/make_traffic_json.py $out/pmc_head 16 4096 c2c $out/traffic_latest.json > /dev/null
STEPS_PER_LAUNCH=32 python tools/make_valu_json.py $out/pmc_head 16 4096 $out/valu_latest.json cpm_entire "profiles/valu_latest.json (rocprofv3 --pmc SQ passes of 32-step launches, tools/profile_r06.sh)" > /dev/null
python tools/phase_timestamps.py > $out/phase_cycles.txt 2> /dev/null
cd /tmp
# ---- config 4: on-ramp 32 x 8192 (injected start; every env restarts every step) and its shape on a map that holds 32 vehicles
C4="--scenario on_ramp_1 --agents 32 --envs-per-gpu 8192 --steps 128 --warmup 32"
$B $C4 --cpu-seconds 8 > $out/bench_config4.json 2> /dev/null
trace config4 $C4 --cpu-seconds 0 --no-compare
A4="--scenario on_ramp_1 --agents 32 --envs-per-gpu 8192 --steps 32 --warmup 32 --cpu-seconds 0 --no-compare"
pmc c4_sq1 "$SQ1" $A4; pmc c4_sq4 "$SQ4" $A4; pmc c4_fetch FETCH_SIZE $A4; pmc c4_write WRITE_SIZE $A4
mkdir -p $out/pmc_c4; for d in sq1 sq4 fetch write; do cp -r $out/pmc_c4_$d $out/pmc_c4/$d; done
$B --agents 32 --envs-per-gpu 8192 --steps 128 --warmup 32 --cpu-seconds 0 --no-compare > $out/bench_cpm_32x8192.json 2> /dev/null
cd $R
python tools/pmc_summary.py $out/pmc_c4 step_wave > $out/pmc_config4_step_kernel.txt
STEPS_PER_LAUNCH=32 python tools/make_traffic_json.py $out/pmc_c4 32 8192 c2c $out/traffic_config4.json on_ramp_1 > /dev/null
cd /tmp
# ---- config 5: centralized CBF-QP before every step (the LEAN instantiation solves the envs; the deferred one is a launch of 128 workgroups that find nothing)
C5="--cbf-qp --steps 64 --warmup 16"
$B $C5 --cpu-seconds 10 > $out/bench_cbf_qp.json 2> /dev/null
trace cbf_qp $C5 --cpu-seconds 0 --no-compare
A5="--cbf-qp --steps 16 --warmup 8 --cpu-seconds 0 --no-compare"
pmc c5_sq1 "$SQ1" $A5; pmc c5_sq2 "$SQ2" $A5; pmc c5_sq3 "$SQ3" $A5; pmc c5_sq4 "$SQ4" $A5
mkdir -p $out/pmc_c5; for d in sq1 sq2 sq3 sq4; do cp -r $out/pmc_c5_$d $out/pmc_c5/$d; done
cd $R
python tools/pmc_summary.py $out/pmc_c5 "cbf_qp_kernel<false, true" > $out/pmc_cbf_qp_kernel.txt
KERNEL_SUBSTR="sigmaenv_cbf_qp_kernel<false, true" python tools/make_valu_json.py $out/pmc_c5 16 2048 $out/valu_dominant_latest.json cpm_entire "profiles/valu_dominant_latest.json (rocprofv3 --pmc SQ passes of bench.py --cbf-qp, LEAN instantiation, tools/profile_r06.sh)" > /dev/null
python - <<PY
import json
d = json.load(open("$out/valu_dominant_latest.json")); d["kernel"] = "sigmaenv_cbf_qp_kernel"; json.dump(d, open("$out/valu_dominant_latest.json", "w"), indent=1)
PY
cd /tmp
# ---- the QP-free margin rewards
CM="--cbf --steps 64 --warmup 16"
$B $CM --cpu-seconds 0 > $out/bench_cbf.json 2> /dev/null
trace cbf $CM --cpu-seconds 0 --no-compare
AM="--cbf --steps 16 --warmup 8 --cpu-seconds 0 --no-compare"
pmc cm_sq1 "$SQ1" $AM; pmc cm_sq2 "$SQ2" $AM; pmc cm_sq3 "$SQ3" $AM; pmc cm_sq4 "$SQ4" $AM
mkdir -p $out/pmc_cm; for d in sq1 sq2 sq3 sq4; do cp -r $out/pmc_cm_$d $out/pmc_cm/$d; done
cd $R
python tools/pmc_summary.py $out/pmc_cm "sigmaenv_cbf_kernel" > $out/pmc_cbf_kernel.txt
KERNEL_SUBSTR="sigmaenv_cbf_kernel" python tools/make_valu_json.py $out/pmc_cm 16 2048 $out/valu_cbf_margin_latest.json cpm_entire "profiles/valu_cbf_margin_latest.json (rocprofv3 --pmc SQ passes of bench.py --cbf, tools/profile_r06.sh)" > /dev/null
# ---- round 4: the measurement lines beside the headline (mtv, reference defaults, drop-in surface, observation rows inside the T-step launch), QP / variant phase stamps
cd $R
tools/bench_lines_r04.sh $tag/lines > /dev/null 2>&1
python tools/qp_phase_cycles.py > $out/qp_phase_cycles.txt 2> /dev/null
PARAMS='{"is_ego_view": false}' python tools/phase_timestamps.py > $out/phase_cycles_obs_bird.txt 2> /dev/null
cd /tmp
trace mtv --steps 256 --warmup 32 --cpu-seconds 0 --no-compare --distance mtv
trace obs_bird --steps 256 --warmup 32 --cpu-seconds 0 --no-compare --param is_ego_view=false
ls $out | head -100
# the raw traces / counter dumps stay on the box (gpurun copies at most 64 MiB back): only the summaries above travel
cd $R; find $out -maxdepth 1 -type d \( -name "trace_*" -o -name "pmc_*" \) -exec rm -rf {} +; find $out -name "*.err" -size +64k -delete; du -sh $out
