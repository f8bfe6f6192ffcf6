"""The rollout exchange over RCCL on the one GPU a test box has: world size 1, ``force_collective=True``, both modes, with the step kernel
writing the record rows on two env-shard streams (the 8-GPU driver run goes through exactly this code with world size 8)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from sigmarl_amd import capi
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
from sigmarl_amd.shard import RolloutExchange, slab_width, unpack_slab
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
N, B, S, T = 16, 128, 2, 4
Bs = B // S
streams = [torch.cuda.Stream(dev) for _ in range(S)]
envs = []
for k in range(S):
    with torch.cuda.stream(streams[k]):
        e = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_apply_mask=False, is_obs_noise=False, is_use_mtv_distance=False), n_envs=Bs, device=dev)
        e.reset_random(seed=3 + k)
        envs.append(e)
torch.cuda.synchronize()
D = envs[0].D
W = slab_width(N, D)
pf, pc = envs[0].map.list_first[0], envs[0].map.list_count[0]
gen = torch.Generator(device=dev).manual_seed(0)
for mode in ("gather", "alltoall"):
    ex = RolloutExchange(B, N, D, T, dev, force_collective=True, mode=mode)
    assert ex.collective
    want = []
    for t in range(T + 2):  # one full chunk + a partial one
        act = torch.rand((B, N, 2), generator=gen, device=dev) * torch.tensor([1.0, 0.5], device=dev) - torch.tensor([0.0, 0.25], device=dev)
        slot = ex.slot(streams)
        for k, e in enumerate(envs):
            with torch.cuda.stream(streams[k]):
                e.set_slab(slot[k * Bs:(k + 1) * Bs])
                e.step(act[k * Bs:(k + 1) * Bs].contiguous())
        torch.cuda.synchronize()
        want.append((torch.cat([e.buffer(capi.BUF_OBS) for e in envs]).clone(), torch.cat([e.buffer(capi.BUF_REWARD) for e in envs]).clone(),
                     torch.cat([e.buffer(capi.BUF_DONE) for e in envs]).bool().clone()))
        ex.advance(streams)
        for k, e in enumerate(envs):
            with torch.cuda.stream(streams[k]):
                e.auto_reset(seed=3 + k, counter=t, path_first=pf, path_count=pc)
    ex.flush(streams)
    ex.wait_all()
    torch.cuda.synchronize()
    assert ex.completed == [(0, T), (1, 2)], ex.completed
    t = 0
    for k, n_valid in ex.completed:
        chunk = ex.gathered(k)[0] if mode == "gather" else ex.time_slice(k)
        assert tuple(chunk.shape) == (T, B, W)
        obs, rew, done = unpack_slab(chunk, N, D)
        for q in range(n_valid):
            assert torch.equal(obs[q], want[t][0]) and torch.equal(rew[q], want[t][1]) and torch.equal(done[q], want[t][2]), (mode, k, q)
            t += 1
    assert t == T + 2
for e in envs:
    e.close()
dist.destroy_process_group()
print("EXCHANGE_OK")
'''


@pytest.mark.gpu
def test_rollout_exchange_over_rccl_world_1(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "EXCHANGE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
