cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 512 --warmup 64 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['one_stream']['ms_per_step'])"; done
