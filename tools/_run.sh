cd /root/repo
rm -rf gpurun_out/r02_g/trace gpurun_out/r02_g/pmc
bash tools/profile_round.sh r02_g > /dev/null 2>&1
cat gpurun_out/r02_g/r02_g_kernel_stats.csv | head -3
timeout 300 python tools/phase_timestamps.py > gpurun_out/r02_g/r02_g_phase_cycles.txt 2>&1; tail -9 gpurun_out/r02_g/r02_g_phase_cycles.txt
