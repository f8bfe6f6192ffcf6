mkdir -p gpurun_out/r2q
python -m pytest tests -m gpu -x -q > gpurun_out/r2q/pytest.log 2>&1; tail -5 gpurun_out/r2q/pytest.log
bash tools/profile_round.sh r02_c
python tools/make_traffic_json.py gpurun_out/r02_c/pmc 16 2048 c2c gpurun_out/r02_c/traffic_latest.json > /dev/null
python tools/make_valu_json.py gpurun_out/r02_c/pmc 16 2048 gpurun_out/r02_c/valu_latest.json > /dev/null
cat gpurun_out/r02_c/r02_c_kernel_stats.csv | head -5; cat gpurun_out/r02_c/traffic_latest.json | head -12; cat gpurun_out/r02_c/valu_latest.json
python tools/phase_timestamps.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_c/r02_c_phase_cycles.txt; cat gpurun_out/r02_c/r02_c_phase_cycles.txt
