#!/usr/bin/env python
"""Randomised differential run: HIP path vs the CPU oracle over random configurations (map, agents, envs, distance type, reward method,
testing mode, observation switches, fixed-duration resets), every buffer compared after every fused step / reset launch.
Diagnostic tool for the GPU box (uses tests/' helpers):   python tools/fuzz_parity.py [--seconds 300] [--seed 0]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

import oracle_binding as ob
import test_gpu_parity as tp
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config

MAPS = ["cpm_entire", "cpm_entire", "cpm_entire", "intersection_1", "on_ramp_1", "roundabout_1", "interchange_2", "intersection_5", "on_ramp_2_multilane", "cpm_mixed"]
REW = ["distance", "ttc", "sparse", "distance_sparse", "ttc_sparse"]


def one_case(rng, k):
    scen = MAPS[rng.integers(len(MAPS))]
    mp = load_map(scen)
    n_max = 20 if scen == "cpm_entire" else 6
    N = int(rng.integers(1, n_max + 1))
    if scen == "cpm_entire" and rng.integers(8) == 0:  # now and then a crowded map: up to 64 agents (one env per wavefront, no lane pairing)
        N = int(rng.integers(21, 65))
    B = int(rng.integers(3, 160))
    kw = dict(n_agents=N, scenario_type=scen, is_use_mtv_distance=bool(rng.integers(2)), rew_method=REW[rng.integers(len(REW))], dt=float(rng.choice([0.05, 0.1])),
              is_testing_mode=bool(rng.integers(4) == 0), is_apply_mask=bool(rng.integers(3) == 0), is_obs_noise=bool(rng.integers(4) == 0), max_steps=int(rng.integers(6, 40)),
              reset_agent_fixed_duration=float(rng.choice([0, 0, 0.5])), n_points_short_term=int(rng.choice([3, 3, 3, 2, 5])),
              is_using_opponent_modeling=bool(rng.integers(6) == 0))
    if rng.integers(3) == 0:  # observation switches
        kw.update(is_obs_steering=bool(rng.integers(2)), is_observe_ref_path_other_agents=bool(rng.integers(2)), is_observe_vertices=bool(rng.integers(2)),
                  is_observe_distance_to_agents=bool(rng.integers(2)), is_observe_distance_to_center_line=bool(rng.integers(2)),
                  is_observe_distance_to_boundaries=bool(rng.integers(2)))
        if rng.integers(3) == 0:  # bird view, with or without the (lanelet-relation) mask; a third of those with the full observation where the reference's reshape allows it
            kw.update(is_ego_view=False)
            if rng.integers(3) == 0:
                kw.update(is_partial_observation=False, n_nearing_agents_observed=int(rng.integers(1, 5)))
                try:
                    from sigmarl_amd.params import obs_flags
                    capi.obs_dim(min(kw["n_nearing_agents_observed"], N - 1), obs_flags(Parameters(**kw)), kw["n_points_short_term"], N)
                except ValueError:
                    kw.update(is_partial_observation=True, n_nearing_agents_observed=2)
    p = Parameters(**kw)
    cfg = make_config(p, mp, B)
    dev, ora = tp._hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    lst = int(rng.integers(len(mp.list_first)))
    if mp.list_count[lst] < 1:
        lst = 0
    pf, pc = int(mp.list_first[lst]), int(mp.list_count[lst])
    mixed = scen == "cpm_mixed" and bool(rng.integers(2))  # the sub-scenario lists instead of one path range (sigmaenv_set_scenario_lists)
    if mixed:
        probs = [float(v) for v in rng.uniform(0.0, 1.0, 3)]
        dev.set_scenario_lists(probs)
        ora.set_scenario_lists(probs)
        pf, pc = 0, capi.SCENARIO_LISTS
    seed = int(rng.integers(1 << 30))
    dev.auto_reset(seed, 0, pf, pc)
    ora.auto_reset(seed, 0, pf, pc)
    tag = f"case {k}: {scen} N={N} B={B} " + " ".join(f"{a}={b}" for a, b in kw.items() if a not in ("n_agents", "scenario_type"))
    n_diff = tp._compare_all(dev, ora, tag + " | initial reset")
    T = int(rng.integers(8, 40))
    for t in range(T):
        mode = rng.integers(4)
        lo, hi, s = [(0, 1, 0.25), (0.1, 0.4, 0.03), (-0.5, 1.6, 0.9), (0.0, 0.0, 0.0)][mode]
        act = np.stack([rng.uniform(lo, hi, (B, N)) if hi > lo else np.zeros((B, N)), rng.uniform(-s, s, (B, N)) if s > 0 else np.zeros((B, N))], axis=-1).astype(np.float32)
        if rng.integers(2):
            dev.step_autoreset(act, seed, t + 1, pf, pc)
        elif rng.integers(3) == 0:  # the in-kernel step loop with a chunk of one step (longer chunks: tests/test_gpu_nstep.py)
            import torch

            dev.env.step_autoreset_n(torch.as_tensor(act[None]).to(dev.env.device).contiguous(), seed=seed, counter0=t + 1, path_first=pf, path_count=pc)
            dev.env.sync()
        else:
            dev.step(act)
            n_diff += 0
            dev.auto_reset(seed, t + 1, pf, pc)
        ora.step(act)
        ora.auto_reset(seed, t + 1, pf, pc)
        n_diff += tp._compare_all(dev, ora, tag + f" | step {t}")
        if rng.integers(4) == 0 and not mixed:  # host-driven resets (sigmaenv_reset): single agents of some envs, or whole envs, onto random centre-line points
            full = bool(rng.integers(2))
            envs = rng.choice(B, size=int(rng.integers(1, min(B, 6) + 1)), replace=False)
            ei, ai, ids, st8 = [], [], [], []
            for e in envs:
                agents = range(N) if full else rng.choice(N, size=int(rng.integers(1, N + 1)), replace=False)
                for i in agents:
                    gp = pf + int(rng.integers(pc))
                    pt = int(rng.integers(1, max(2, int(mp.n_center[gp]) - 2)))
                    x, y = [float(v) for v in mp.center[gp, pt]]
                    yaw = float(mp.yaw[gp, min(pt, int(mp.n_yaw[gp]) - 1)])
                    sp = float(rng.uniform(0, 1))
                    ei.append(int(e)); ai.append(int(i)); ids.append((gp, lst, gp - pf, pt))
                    st8.append((x, y, yaw, sp, 0.0, sp * np.cos(np.float32(yaw)), sp * np.sin(np.float32(yaw)), 0.0))
            for env_ in (dev, ora):
                env_.reset(np.asarray(ei, np.int32), np.asarray(ai, np.int32), np.asarray(ids, np.int32), np.asarray(st8, np.float32), int(full))
                env_.observe()
            n_diff += tp._compare_all(dev, ora, tag + f" | host reset after step {t}")
    dev.close()
    ora.close()
    return tag, T * B * N, n_diff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    k = n_steps = n_diff = 0
    while time.time() - t0 < args.seconds:
        tag, steps, diff = one_case(rng, k)
        n_steps += steps
        n_diff += diff
        if diff:
            print("differing non-observation words:", diff, tag)
        k += 1
    print(f"fuzz: {k} configurations, {n_steps} agent-steps compared, {n_diff} differing non-observation fp32 words, no mask / index mismatch, all floats within {tp.FTOL}")


if __name__ == "__main__":
    main()
