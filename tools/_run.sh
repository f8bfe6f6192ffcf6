cd /root/repo
mkdir -p gpurun_out/r02_e
timeout 600 python bench.py --sweep > gpurun_out/r02_e/bench_default_sweep.json 2> gpurun_out/r02_e/bench.err; tail -c 3000 gpurun_out/r02_e/bench_default_sweep.json
timeout 120 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-like', d['value'], d['ms_per_step'], d['config'].get('one_stream'), d['roofline']['kernel_launches'])"
for v in "--policy" "--policy --policy-precision bf16" "--cbf" "--cbf-qp" "--distance mtv" "--scenario on_ramp_1 --agents 32 --envs-per-gpu 8192"; do timeout 300 python bench.py --steps 128 --warmup 16 --cpu-seconds 0 --no-one-stream $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"; done
