cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or variants" 2>&1 | tail -5
timeout 300 python bench.py --steps 512 --warmup 64 --cpu-seconds 0 --no-one-stream 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
