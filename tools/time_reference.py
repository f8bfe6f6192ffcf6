#!/usr/bin/env python
"""BASELINE.md section 3, step 1 (BUILD CONTAINER ONLY): times the imported reference (PyTorch-CPU, under the import shims of
tests/golden/gen/refshim.py) and the repo's C oracle on the SAME host at N = 16, B in {256, 1024, 4096}, cpm_entire, c2c and mtv, and prints
the table + the oracle / reference ratio that translates the GPU box's `cpu_baseline` ("port") into "times the original PyTorch-CPU path".

Usage: python tools/time_reference.py [--envs 256,1024,4096] [--steps 3] [--json out.json]
The reference's step time is flat in the batch size per env (per-env Python loops), so B = 4096 takes ~20 s per step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "gen"))

import numpy as np
import torch


def time_reference(B, N, mtv, steps):
    import refshim
    refshim.install()
    from sigmarl.helper_common import Parameters

    torch.manual_seed(0)
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=B, dt=0.05,
                   is_use_mtv_distance=mtv, rew_method="distance", is_challenging_initial_state_buffer=False, is_testing_mode=False, max_steps=128)
    env = refshim.RefEnv(p, B)
    gen = torch.Generator().manual_seed(1)

    def one():
        act = torch.stack([torch.rand(B, N, generator=gen), torch.rand(B, N, generator=gen) * 0.5 - 0.25], dim=-1)
        _, _, done, _ = env.step(act)
        for e in torch.where(done)[0].tolist():
            env.reset_env(e)

    one()  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    return (time.perf_counter() - t0) / steps


def time_oracle(B, N, mtv, seconds=4.0):
    import oracle_binding as ob
    from sigmarl_amd import capi
    from sigmarl_amd.maps import load_map
    from sigmarl_amd.params import Parameters, make_config

    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, dt=0.05, is_use_mtv_distance=mtv,
                   rew_method="distance", max_steps=128)
    mp = load_map("cpm_entire")
    env = ob.OracleEnv(make_config(p, mp, B), mp)
    pf, pc = mp.list_first[0], mp.list_count[0]
    env.get(capi.BUF_DONE, copy=False)[:] = 1
    env.auto_reset(0, 0, pf, pc)
    rng = np.random.default_rng(0)
    act = np.stack([rng.uniform(0, 1, (B, N)), rng.uniform(-0.25, 0.25, (B, N))], axis=-1).astype(np.float32)
    env.step(act); env.auto_reset(0, 1, pf, pc)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < seconds:
        env.step(act); env.auto_reset(0, k + 2, pf, pc); k += 1
    dt = (time.perf_counter() - t0) / k
    env.close()
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", default="256,1024,4096")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--agents", type=int, default=16)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    N = args.agents
    rows = []
    host = dict(cpus=os.cpu_count(), torch_threads=torch.get_num_threads(), torch=torch.__version__)
    try:
        host["cpu_model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:  # noqa: BLE001
        pass
    print(host)
    for mtv in (False, True):
        for B in [int(x) for x in args.envs.split(",")]:
            t_ref = time_reference(B, N, mtv, args.steps)
            t_ora = time_oracle(B, N, mtv)
            r = dict(distance="mtv" if mtv else "c2c", n_agents=N, n_envs=B, reference_s_per_step=t_ref, reference_agent_env_steps_per_s=N * B / t_ref,
                     oracle_s_per_step=t_ora, oracle_agent_env_steps_per_s=N * B / t_ora, oracle_over_reference=t_ref / t_ora)
            rows.append(r)
            print(json.dumps(r))
    if args.json:
        json.dump(dict(host=host, rows=rows), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
