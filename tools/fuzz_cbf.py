#!/usr/bin/env python
"""Randomised differential run of the CBF module: margin reward channels and the (centralized / grouped) CBF-QP of the HIP path vs the CPU
oracle over random configurations (map, agents, circles, nominal controller, lambda penalty, grouping) on sampled and stepped states.
Diagnostic tool for the GPU box:   python tools/fuzz_cbf.py [--seconds 240] [--seed 0]"""
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
from sigmarl_amd import capi, cbf
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config

MAPS = ["cpm_entire", "cpm_entire", "intersection_1", "on_ramp_1", "roundabout_1"]


STATS = {"iter_mismatch": 0, "solves": 0, "repeat_checked": 0}  # iteration counts HIP vs oracle; bitwise repeats of the HIP solve (CHECK_REPEAT)
CHECK_REPEAT = [False]  # tests/test_gpu_fuzz.py: every QP is solved twice on the device and the two results must be the same bits (minimiser, safe action, iteration counts)
FORCE_N = [0]           # tests/test_gpu_fuzz.py: the vehicle count of the next CPM case (odd counts beyond 32: the packed Hessian's last word)
UNCONVERGED = [0]  # solves at the iteration limit on both sides (only beyond 16 vehicles)
CPM_MAX_AGENTS = 16  # (--cpm-agents: up to 64 vehicles on the CPM map -- the two / one lanes per vehicle layouts of the QP kernel's register path and its <BIG> instantiation)


def one_case(rng, k):
    scen = MAPS[rng.integers(len(MAPS))]
    mp = load_map(scen)
    N = int(rng.integers(1, (CPM_MAX_AGENTS if scen.startswith("cpm") else 5) + 1))
    if CPM_MAX_AGENTS > 16 and scen.startswith("cpm"):
        N = int(rng.integers(17, CPM_MAX_AGENTS + 1))
        if FORCE_N[0]:
            N = FORCE_N[0]
    B = int(rng.integers(4, 40))
    solve = bool(rng.integers(3) > 0)
    grouping = solve and N >= 3 and bool(rng.integers(3) == 0)
    kw = dict(n_agents=N, scenario_type=scen, dt=float(rng.choice([0.05, 0.1])), rew_method=str(rng.choice(["cbf", "cbf_sparse"])), is_solve_qp=solve,
              is_using_cbf_training=True, is_obs_noise=False, is_apply_mask=False, nom_controller_type=str(rng.choice(["rl", "clf"])),
              adaptive_lambda=bool(rng.integers(2)) or grouping, n_circles_approximate_vehicle=int(rng.integers(1, 5)), is_use_mtv_distance=bool(rng.integers(2)),
              is_apply_cbf_action=bool(rng.integers(2)), max_steps=20)
    if grouping:
        kw.update(is_grouping_agents=True, max_group_size=int(rng.integers(1, max(2, N))), observation_range=float(rng.choice([0.3, 0.5, 1.0])))
    p = Parameters(**kw)
    cfg = make_config(p, mp, B)
    dev, ora = tp._hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    cc = cbf.make_cbf_config(p)
    for e in (dev, ora):
        e.cbf_attach(cc, seg_l, seg_r)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    pf, pc = int(mp.list_first[0]), int(mp.list_count[0])
    seed = int(rng.integers(1 << 30))
    dev.auto_reset(seed, 0, pf, pc)
    ora.auto_reset(seed, 0, pf, pc)
    tag = f"case {k}: {scen} N={N} B={B} " + " ".join(f"{a}={b}" for a, b in kw.items() if a not in ("n_agents", "scenario_type"))
    if os.environ.get("FUZZ_VERBOSE"):
        print(tag, flush=True)
    worst_u = worst_m = 0.0
    for t in range(int(rng.integers(2, 6))):
        act = np.stack([rng.uniform(-0.3, 1.2, (B, N)), rng.uniform(-0.5, 0.5, (B, N))], axis=-1).astype(np.float32)
        if solve:
            sd, ud, idv = dev.cbf_qp(act)
            if CHECK_REPEAT[0]:
                sd2, ud2, idv2 = dev.cbf_qp(act)
                assert np.array_equal(idv, idv2) and np.array_equal(ud.view(np.uint64), ud2.view(np.uint64)) and np.array_equal(sd.view(np.uint32), sd2.view(np.uint32)), \
                    tag + f" | a second solve of the same problem differs at step {t}: iterations {idv[:, 0].tolist()} vs {idv2[:, 0].tolist()}"
                STATS["repeat_checked"] += B
            so, uo, io = ora.cbf_qp(act)[:3]
            STATS["solves"] += B
            STATS["iter_mismatch"] += int((idv[:, 0] != io[:, 0]).sum())
            if not np.array_equal(idv[:, 1], io[:, 1]) or (CPM_MAX_AGENTS <= 16 and not idv[:, 1].all()):
                # up to 16 vehicles every solve converges; beyond (--cpm-agents), crammed scenes hit the iteration limit -- on BOTH sides, for the same envs, or it is a finding
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_cbf_fail.npz"), state=ora.get(capi.BUF_STATE), path=ora.get(capi.BUF_PATH), short=ora.get(capi.BUF_SHORT_TERM),
                         act=act, info_hip=idv, info_ora=io, u_hip=ud, u_ora=uo, kw=np.asarray(repr(kw)))
                raise AssertionError(tag + f" | not converged at step {t}: hip {np.flatnonzero(idv[:, 1] == 0).tolist()} iters {idv[:, 0].max()}, oracle {np.flatnonzero(io[:, 1] == 0).tolist()} iters {io[:, 0].max()}")
            UNCONVERGED[0] += int((idv[:, 1] == 0).sum())
            keep = idv[:, 1] != 0
            ud, uo = ud[keep], uo[keep]  # (an env at the iteration limit: wherever the iteration happened to stand -- the safe action compared below is the nominal one on both sides)
            if len(ud) == 0:
                ud, uo = np.zeros((1, 1, 2)), np.zeros((1, 1, 2))
            if grouping:
                assert np.array_equal(dev.cbf_groups(), ora.cbf_groups()), tag + " | groups"
            du = float(np.abs(ud - uo).max())
            if du > 5e-7:  # (the unit tests hold 1e-7 on their instances; over ~1e5 random solves the worst seen is 1.4e-7: the minimiser is only determined to
                           #  that level by the 1e9-weighted terms' rounding noise, and the two sides sum in different orders)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_cbf_fail.npz"), state=ora.get(capi.BUF_STATE), path=ora.get(capi.BUF_PATH), short=ora.get(capi.BUF_SHORT_TERM),
                         act=act, info_hip=idv, info_ora=io, u_hip=ud, u_ora=uo, kw=np.asarray(repr(kw)))
            assert du <= 5e-7, tag + f" | u differs by {du} in envs {np.flatnonzero(np.abs(ud - uo).max(axis=(1, 2)) > 1e-7).tolist()}"
            assert np.abs(sd - so).max() <= 1e-6, tag + " | safe action"
            worst_u = max(worst_u, du)
            step_act = sd if (kw["is_apply_cbf_action"] or grouping) else act
        else:
            md = dev.cbf_rewards(act)
            mo = ora.cbf_rewards(act)
            for x, y in zip(md, mo):
                m = ~np.isnan(y)
                assert np.array_equal(np.isnan(x), np.isnan(y)), tag + " | margin set"
                if m.any():
                    same_inf = np.isinf(x[m]) & np.isinf(y[m]) & (np.sign(x[m]) == np.sign(y[m]))
                    with np.errstate(invalid="ignore"):
                        d = float(np.where(same_inf, 0.0, np.abs(x[m] - y[m]) / np.maximum(1.0, np.abs(y[m]))).max())
                    assert d <= 1e-11, tag + f" | margins differ by {d}"
                    worst_m = max(worst_m, d)
            step_act = act
        dev.step(step_act)
        ora.step(step_act)
        tp._compare_all(dev, ora, tag + f" | step {t}")
        dev.auto_reset(seed, t + 1, pf, pc)
        ora.auto_reset(seed, t + 1, pf, pc)
    dev.close()
    ora.close()
    return B, solve, grouping, worst_u, worst_m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpm-agents", type=int, default=16, help="largest vehicle count on the CPM map (17 .. 64: only such cases are drawn there)")
    args = ap.parse_args()
    global CPM_MAX_AGENTS
    CPM_MAX_AGENTS = args.cpm_agents
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    k = n_qp = n_grp = n_margin = 0
    wu = wm = 0.0
    while time.time() - t0 < args.seconds:
        B, solve, grouping, du, dm = one_case(rng, k)
        k += 1
        n_qp += int(solve and not grouping); n_grp += int(grouping); n_margin += int(not solve)
        wu, wm = max(wu, du), max(wm, dm)
    conv = "all converged" if UNCONVERGED[0] == 0 else str(UNCONVERGED[0]) + " solves at the iteration limit on both sides alike"
    print(f"fuzz_cbf: {k} configurations ({n_qp} centralized QP, {n_grp} grouped QP, {n_margin} margin reward): {conv}, max |u_hip - u_oracle| {wu:.2e}, "
          f"max relative margin difference {wm:.2e}, env buffers identical")


if __name__ == "__main__":
    main()
