set -x
cd /root/repo
mkdir -p gpurun_out/r02_d
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for i in 1 2; do timeout 300 python bench.py --steps 512 --warmup 64 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('shards'))"; done
timeout 300 python bench.py --streams 1 --steps 512 --warmup 64 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('shards'))"
timeout 300 python tools/phase_timestamps.py 2>&1 | tail -12
