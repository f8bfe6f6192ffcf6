cd /root/repo
for gs in 0 2 4; do timeout 300 python bench.py --steps 64 --warmup 8 --cpu-seconds 0 --cbf-qp --cbf-group-size $gs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group', $gs, d['value'], d['ms_per_step'])"; done
