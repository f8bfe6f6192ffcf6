cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
for ex in alltoall gather; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 64 --warmup 8 --cpu-seconds 0 --force-dist --exchange $ex 2>gpurun_out/dist_$ex.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$ex', d['value'], d['ms_per_step'], d['config']['rollout_gather'][:90])"
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun N=1', d['value'], d['ms_per_step'])"
