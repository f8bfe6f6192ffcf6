cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 512 --warmup 64 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
timeout 300 python bench.py --scenario on_ramp_1 --agents 32 --envs-per-gpu 8192 --steps 64 --warmup 8 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config4', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --agents 32 --envs-per-gpu 8192 --steps 64 --warmup 8 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('32x8192 cpm', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --agents 4 --envs-per-gpu 16384 --steps 64 --warmup 8 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('4x16384', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --agents 8 --envs-per-gpu 8192 --steps 64 --warmup 8 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8x8192', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --distance mtv --steps 256 --warmup 32 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mtv', d['value'], d['ms_per_step'])"
