#!/usr/bin/env python
"""Diagnostic: T-step launches of the whole batch on one stream against S shards of B / S envs on S streams (does the other shard's launch fill the end of a launch, where the
SIMDs run out of wavefronts?).  No rollout record.  Usage: python tools/shard_tstep_probe.py [B] [T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = 16
dev = torch.device("cuda:0")
def run(S, launches=64, offset=False):
    Bs = B // S
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    envs, acts = [], []
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            e = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=Bs, device="cuda:0", env_index_base=k * Bs)
            e.reset_random(seed=1)
            envs.append(e)
            acts.append((torch.rand((T, Bs, N, 2), device=dev) * torch.tensor([1.0, 0.5], device=dev) - torch.tensor([0.0, 0.25], device=dev)).contiguous())
    torch.cuda.synchronize()
    pf, pc = envs[0].map.list_first[0], envs[0].map.list_count[0]
    def go(n, c0):
        for i in range(n):
            for k, e in enumerate(envs):
                with torch.cuda.stream(streams[k]):
                    if offset and i == 0 and k > 0:  # start shard k a fraction of a launch late: the ends of the shards' launches never coincide
                        e.step_autoreset_n(acts[k][: max(1, (T * k) // S)], seed=1, counter0=c0, path_first=pf, path_count=pc)
                    e.step_autoreset_n(acts[k], seed=1, counter0=c0 + i * T, path_first=pf, path_count=pc)
    go(24, 0)  # sustained clocks
    torch.cuda.synchronize()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        go(launches, 1000 * (rep + 1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, launches * T * B * N / dt)
    for e in envs:
        e.close()
    return best
for S, off in ((1, False), (2, False), (2, True), (4, True)):
    print("B %d T %d: %d shard(s)%s: %.4g agent-env-steps/s" % (B, T, S, " staggered" if off else "", run(S, offset=off)), flush=True)
