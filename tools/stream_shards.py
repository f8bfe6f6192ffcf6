#!/usr/bin/env python
"""Experiment: the envs of one GPU stepped as S independent shards on S HIP streams (same total work per step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters

B, N = int(os.environ.get("B", 4096)), 16
for S in [int(x) for x in os.environ.get("S", "1,2,4").split(",")]:
    Bs = B // S
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    envs = []
    for k, st in enumerate(streams):
        with torch.cuda.stream(st):
            e = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="distance",
                                    is_apply_mask=False, is_obs_noise=False, max_steps=128), n_envs=Bs, device="cuda:0")
            e.reset_random(seed=100 + k)
            envs.append(e)
    gen = torch.Generator(device="cuda").manual_seed(0)
    acts = torch.empty((16, B, N, 2), device="cuda")
    acts[..., 0] = torch.rand((16, B, N), generator=gen, device="cuda")
    acts[..., 1] = torch.rand((16, B, N), generator=gen, device="cuda") * 0.5 - 0.25
    shard_acts = [[acts[t, k * Bs:(k + 1) * Bs].contiguous() for t in range(16)] for k in range(S)]
    slabs = [torch.empty((Bs, N * 33 + 1), device="cuda") for _ in range(S)]
    pf, pc = envs[0].map.list_first[0], envs[0].map.list_count[0]
    torch.cuda.synchronize()

    def run(n, t0):
        for t in range(n):
            for k, e in enumerate(envs):
                e.set_slab(slabs[k])
                e.step_autoreset(shard_acts[k][(t0 + t) % 16], seed=100 + k, counter=t0 + t + 1, path_first=pf, path_count=pc)

    if S > 1 and os.environ.get("SKEW", "1") == "1":  # start the shards out of phase
        for k, e in enumerate(envs):
            with torch.cuda.stream(streams[k]):
                torch.cuda._sleep(int(40000 * k))
    run(64, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(512, 64)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"S={S}: {el / 512 * 1e3:.4f} ms/step  {N * B * 512 / el:.4g} agent-env-steps/s")
    for e in envs:
        e.close()
