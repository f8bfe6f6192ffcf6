cd /root/repo
bash tools/profile_round.sh r02_d > /dev/null 2>&1
cat gpurun_out/r02_d/r02_d_kernel_stats.csv | head -5
python tools/make_traffic_json.py gpurun_out/r02_d/pmc > gpurun_out/r02_d/traffic_latest.json 2>/dev/null; cat gpurun_out/r02_d/traffic_latest.json | head -20
python tools/make_valu_json.py gpurun_out/r02_d/pmc > gpurun_out/r02_d/valu_latest.json 2>/dev/null; cat gpurun_out/r02_d/valu_latest.json
timeout 300 python tools/phase_timestamps.py > gpurun_out/r02_d/r02_d_phase_cycles.txt 2>&1; tail -10 gpurun_out/r02_d/r02_d_phase_cycles.txt
bash tools/pmc_ablation.sh gpurun_out/r02_d/ablation > gpurun_out/r02_d/r02_d_pmc_phase_ablation.txt 2>&1
grep -E "skip=|SQ_INSTS_VALU " gpurun_out/r02_d/r02_d_pmc_phase_ablation.txt | head -40
