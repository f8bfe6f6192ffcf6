"""GPU parity of the QP-free CBF margin reward through the C-ABI: HIP path vs the CPU oracle on the same
seeded inputs and vs goldens produced by running the reference (sigmarl/cbf_qp.py:2534-2804, sigmarl/pseudo_distance.py).

Bars: the three reward channels (float32) identical to the oracle's up to one float32 ulp; the float64 margins within 1e-11 relative
(the fp32 / fp16 part of the path is bit-identical by construction, the float64 part only differs through the device's double-precision
trigonometric functions).  Against the reference goldens: see tests/traj_replay.py (fp16 rounding flips).
"""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
import traj_replay as tr
from sigmarl_amd import capi, cbf
from sigmarl_amd.maps import MapTable, load_map
from sigmarl_amd.params import Parameters, make_config
from test_oracle_golden import _cbf_fixture, cbf_case_env

pytestmark = pytest.mark.gpu

MARGIN_RTOL = 1e-11
REW_TOL = 1.2e-7


def _hip_env(cfg, mp):
    from sigmarl_amd.env import NumpyAdapter, SigmaEnv

    return NumpyAdapter(SigmaEnv(cfg=cfg, map_table=mp, device="cuda:0"))


def _cmp_margins(a, b, tag):
    for name, x, y in zip(("lane_left", "lane_right", "pair"), a, b):
        m = ~np.isnan(y)
        assert np.array_equal(np.isnan(x), np.isnan(y)), f"{tag}: {name}: different set of written entries"
        d = np.abs(x[m] - y[m]) / np.maximum(1.0, np.abs(y[m]))
        if d.size == 0:  # a single agent has no pairs
            continue
        assert d.max() <= MARGIN_RTOL, f"{tag}: {name}: max rel err {d.max():.3e}"


def _cmp_rewards(dev, ora, tag):
    a, b = dev.get(capi.BUF_REWARD_INFO)[4:7], ora.get(capi.BUF_REWARD_INFO)[4:7]
    assert np.abs(a.astype(np.float64) - b).max() <= REW_TOL, f"{tag}: reward channels differ by {np.abs(a - b).max()}"
    return int((a != b).sum())


def test_cbf_set_states_vs_oracle_and_reference():
    z, meta = _cbf_fixture()
    dev = cbf_case_env(_hip_env, z, meta)
    ora = cbf_case_env(ob.OracleEnv, z, meta)
    md, mo = dev.cbf_rewards(z["p2_act"]), ora.cbf_rewards(z["p2_act"])
    _cmp_margins(md, mo, "set states")
    assert _cmp_rewards(dev, ora, "set states") <= 2
    rep = tr.Report()
    rep.cbf("lane_left", md[0], z["p2_lane_left"])
    rep.cbf("lane_right", md[1], z["p2_lane_right"])
    rep.cbf("pair", md[2], z["p2_pair"])
    ri = dev.get(capi.BUF_REWARD_INFO)
    rep.cbf("rew", np.stack([ri[5], ri[6], ri[4]]), z["p2_rew"])
    for e in (dev, ora):
        e.cbf_inject_centers(z["p2_centers"])
    md, mo = dev.cbf_rewards(z["p2_act"]), ora.cbf_rewards(z["p2_act"])
    _cmp_margins(md, mo, "set states, injected centres")
    rep.cbf("inj_lane_left", md[0], z["p2_lane_left"])
    rep.cbf("inj_lane_right", md[1], z["p2_lane_right"])
    rep.cbf("inj_pair", md[2], z["p2_pair"])
    ri = dev.get(capi.BUF_REWARD_INFO)
    rep.cbf("inj_rew", np.stack([ri[5], ri[6], ri[4]]), z["p2_rew"])
    print(f"cbf_functions (HIP): {rep}")
    assert rep.cbf_ok("cbf_functions"), str(rep)
    assert all(rep.cbf_bad[k] == 0 for k in rep.cbf_bad if k.startswith("inj_")), str(rep)  # with the reference's own centres: no exception at all
    dev.close()
    ora.close()


CASES = [
    # scenario, N, B, circles, rew_method, dt, steps, nominal controller
    ("cpm_entire", 16, 40, 3, "cbf", 0.05, 8, "rl"),
    ("cpm_entire", 5, 33, 2, "cbf_sparse", 0.1, 8, "rl"),     # ragged sizes, two circles
    ("intersection_1", 4, 24, 4, "cbf", 0.1, 10, "rl"),        # non-loop map, four circles
    ("on_ramp_1", 6, 17, 1, "cbf_sparse", 0.05, 8, "rl"),      # one circle: the centre only
    ("cpm_entire", 32, 6, 3, "cbf", 0.05, 4, "rl"),
    ("cpm_entire", 8, 21, 3, "cbf", 0.05, 8, "clf"),           # margins at the CLF controller's action
    ("cpm_entire", 64, 3, 4, "cbf", 0.05, 2, "rl"),            # the maxima: 64 agents, 4 circles (42 KB of LDS per workgroup)
    ("cpm_entire", 1, 9, 3, "cbf", 0.05, 4, "rl"),             # one agent: no pairs
]


@pytest.mark.parametrize("scen,N,B,Cc,rew,dt,steps,nom", CASES)
def test_cbf_rollout_vs_oracle(scen, N, B, Cc, rew, dt, steps, nom):
    """Seeded rollouts with the margin rewards computed before every step and consumed by the step's reward (rew_method "cbf...")."""
    from test_gpu_parity import _compare_all

    p = Parameters(n_agents=N, scenario_type=scen, rew_method=rew, dt=dt, is_solve_qp=False, is_using_cbf_training=True, is_apply_mask=False,
                   is_obs_noise=False, max_steps=6, n_circles_approximate_vehicle=Cc, is_use_mtv_distance=False, nom_controller_type=nom)
    mp = load_map(scen)
    cfg = make_config(p, mp, B)
    assert cfg.rew_flags & capi.REW_CBF
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    cc = cbf.make_cbf_config(p)
    dev.cbf_attach(cc, seg_l, seg_r)
    ora.cbf_attach(cc, seg_l, seg_r)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    pf, pc = mp.list_first[0], mp.list_count[0]
    dev.auto_reset(9, 0, pf, pc)
    ora.auto_reset(9, 0, pf, pc)
    rng = np.random.default_rng(77)
    n_diff = n_neg = 0
    for t in range(steps):
        act = np.stack([rng.uniform(-0.7, 1.3, (B, N)), rng.uniform(-0.7, 0.7, (B, N))], axis=-1).astype(np.float32)
        md, mo = dev.cbf_rewards(act), ora.cbf_rewards(act)
        _cmp_margins(md, mo, f"{scen} step {t}")
        n_diff += _cmp_rewards(dev, ora, f"{scen} step {t}")
        n_neg += int((mo[0] < 0).sum() + (mo[1] < 0).sum() + (np.nan_to_num(mo[2]) < 0).sum())
        # both sides continue from the ORACLE's channels so that a one-ulp reward difference cannot leak into the state comparison
        dev.env.buffer(capi.BUF_REWARD_INFO)[4:7].copy_(__import__("torch").from_numpy(ora.get(capi.BUF_REWARD_INFO)[4:7]))
        dev.step(act)
        ora.step(act)
        _compare_all(dev, ora, f"{scen} step {t}")
        dev.auto_reset(9, t + 1, pf, pc)
        ora.auto_reset(9, t + 1, pf, pc)
    assert n_neg > 0  # violated constraints occurred
    assert n_diff <= 4, n_diff
    dev.close()
    ora.close()


def test_cbf_full_size_sample_vs_oracle():
    """BASELINE size (16 agents x 4096 envs): after a few steps the reward channels of the first and the last 96 envs equal the oracle's
    on the same states; every channel of every env is finite and in [-1, 0]."""
    import torch
    from sigmarl_amd.env import SigmaEnv

    B, N, S = 4096, 16, 96
    p = Parameters(n_agents=N, scenario_type="cpm_entire", rew_method="cbf", dt=0.05, is_solve_qp=False, is_using_cbf_training=True,
                   is_apply_mask=False, is_obs_noise=False, is_use_mtv_distance=False)
    env = SigmaEnv(p, n_envs=B, device="cuda:0")
    env.reset_random(seed=11)
    env.cbf_attach()
    g = torch.Generator(device="cuda").manual_seed(3)
    for t in range(6):
        act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.4 - 0.2, torch.rand(B, N, generator=g, device="cuda") * 1.0 - 0.5], -1).contiguous()
        env.cbf_rewards(act)
        if t < 5:
            env.step_autoreset(act, seed=2)
    env.sync()
    ch = env.buffer(capi.BUF_REWARD_INFO)[4:7]
    assert torch.isfinite(ch).all() and float(ch.min()) >= -1.0 and float(ch.max()) <= 0.0
    assert int((ch < 0).sum()) > 1000
    mp = env.map
    seg_l, seg_r = cbf.load_segment_tables(mp)
    for lo in (0, B - S):
        ora = ob.OracleEnv(make_config(p, mp, S), mp)
        ora.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        st = env.buffer(capi.BUF_STATE)[lo:lo + S].cpu().numpy()
        pa = env.buffer(capi.BUF_PATH)[lo:lo + S].cpu().numpy()
        ora.reset(np.repeat(np.arange(S), N), np.tile(np.arange(N), S), pa.reshape(-1, 4), st.reshape(-1, 8), 1)
        ora.cbf_rewards(act[lo:lo + S].cpu().numpy(), want_margins=False)
        d = np.abs(ch[:, lo:lo + S].cpu().numpy() - ora.get(capi.BUF_REWARD_INFO)[4:7])
        assert d.max() <= REW_TOL, (lo, d.max())
        ora.close()
    env.close()


def test_cbf_degenerate_states_vs_oracle():
    """Vehicles metres off their lane and outside the map (large pseudo distances, bounds that prune nothing), coincident vehicles, zero
    speed, reversed heading: margins, channels and the QP agree with the oracle."""
    z, meta = _cbf_fixture()
    N, B = 16, 12
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True,
                   is_obs_noise=False, is_apply_mask=False)
    cfg = make_config(p, mp, B)
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    st8 = np.zeros((B, N, 8), np.float32)
    st8[..., :5] = z["p2_state"][:B]
    rng = np.random.default_rng(4)
    st8[:, 2, 0:2] += rng.uniform(-3.0, 3.0, (B, 2)).astype(np.float32)      # metres away from the own lane
    st8[:, 3, 0:2] = np.float32([40.0, -25.0])                                # outside the map altogether
    st8[:, 5, 0:3] = st8[:, 4, 0:3]                                           # coincident with vehicle 4
    st8[:, 6, 3] = 0.0                                                        # standing still
    st8[:, 7, 2] += np.float32(np.pi)                                         # facing backwards
    ids = np.zeros((B, N, 4), np.int32)
    ids[..., 0] = z["p2_path"][:B]
    ids[..., 2] = ids[..., 0]
    for e in (dev, ora):
        e.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        e.reset(np.repeat(np.arange(B), N), np.tile(np.arange(N), B), ids.reshape(-1, 4), st8.reshape(-1, 8), 1)
    act = z["p2_act"][:B]
    md, mo = dev.cbf_rewards(act), ora.cbf_rewards(act)
    _cmp_margins(md, mo, "degenerate")
    _cmp_rewards(dev, ora, "degenerate")
    assert np.isfinite(mo[0]).all() and np.abs(mo[0]).max() > 10  # far-away vehicles: margins of tens of metres
    safe_d, u_d, info_d = dev.cbf_qp(act)
    safe_o, u_o, info_o = ora.cbf_qp(act)
    assert info_d[:, 1].all() and info_o[:, 1].all()
    assert np.abs(u_d - u_o).max() <= 1e-6 and np.abs(safe_d - safe_o).max() <= 1e-6
    dev.close()
    ora.close()


def test_cbf_own_segment_tables_on_compiled_map():
    """A map without a shipped table asset (compiled by sigmarl_amd.mapc): the tables come from sigmarl_amd.cbf.segment_tables."""
    from sigmarl_amd import mapc

    table = mapc.compile_scenario("intersection_2")
    mp = MapTable("intersection_2", table=table)
    assert not mp.from_asset
    N, B = 4, 9
    p = Parameters(n_agents=N, scenario_type="intersection_2", rew_method="cbf", dt=0.1, is_solve_qp=False, is_using_cbf_training=True,
                   is_apply_mask=False, is_obs_noise=False, is_use_mtv_distance=False)
    cfg = make_config(p, mp, B)
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    cc = cbf.make_cbf_config(p)
    dev.cbf_attach(cc, seg_l, seg_r)
    ora.cbf_attach(cc, seg_l, seg_r)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    dev.auto_reset(3, 0, mp.list_first[0], mp.list_count[0])
    ora.auto_reset(3, 0, mp.list_first[0], mp.list_count[0])
    act = np.random.default_rng(5).uniform(-0.5, 1.0, (B, N, 2)).astype(np.float32)
    _cmp_margins(dev.cbf_rewards(act), ora.cbf_rewards(act), "compiled map")
    _cmp_rewards(dev, ora, "compiled map")
    dev.close()
    ora.close()


def test_cbf_full_scan_fallback_matches(monkeypatch):
    """Without the chunk boxes (SIGMAENV_PRUNE=0) the kernel scans every segment: same margins as the pruned scan and the oracle."""
    z, meta = _cbf_fixture()
    monkeypatch.setenv("SIGMAENV_PRUNE", "0")
    dev = cbf_case_env(_hip_env, z, meta)
    monkeypatch.delenv("SIGMAENV_PRUNE")
    ora = cbf_case_env(ob.OracleEnv, z, meta)
    _cmp_margins(dev.cbf_rewards(z["p2_act"]), ora.cbf_rewards(z["p2_act"]), "full scan")
    _cmp_rewards(dev, ora, "full scan")
    dev.close()
    ora.close()


def test_cbf_requires_attach_and_rejects_grouping_without_qp():
    mp = load_map("cpm_entire")
    with pytest.raises(NotImplementedError):  # the reference's grouped update raises without is_solve_qp (lam=None in its coefficient builders)
        make_config(Parameters(n_agents=4, rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True, is_grouping_agents=True, is_apply_mask=False,
                               is_obs_noise=False), mp, 2)
    p = Parameters(n_agents=4, rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True, is_apply_mask=False, is_obs_noise=False)
    dev = _hip_env(make_config(p, mp, 2), mp)
    with pytest.raises(RuntimeError, match="cbf_attach"):
        dev.cbf_rewards(np.zeros((2, 4, 2), np.float32))
    dev.close()


@pytest.mark.parametrize("nominal,adaptive,N", [("rl", False, 16), ("rl", True, 16), ("clf", False, 16), ("clf", True, 8), ("rl", False, 32), ("rl", False, 3), ("rl", True, 1),
                                                ("rl", True, 9), ("rl", False, 13), ("clf", True, 5),  # (2 N not a power of two: the factor's LDS area)
                                                ("rl", False, 33), ("clf", True, 40), ("rl", True, 64)])  # beyond 32 vehicles: packed Hessian, candidate list in HBM
def test_cbf_qp_vs_oracle_and_kkt(nominal, adaptive, N):
    """The centralized CBF-QP (sigmarl/cbf_qp.py:733-1400): HIP minimiser == the oracle's, and it satisfies the KKT conditions of the
    original problem (checked in numpy on the constraint data; cvxpy / OSQP are absent, see tests/test_cbf_qp.py)."""
    from test_cbf_qp import check_kkt, qp_case

    B = 48 if N <= 16 else (12 if N <= 32 else 6)
    dev, act = qp_case(_hip_env, N=N if N <= 16 else 16, B=B, nominal=nominal, adaptive_lambda=adaptive) if N <= 16 else (None, None)
    if N > 16:  # 32 agents: sampled starts (the set-state fixture has 16 agents per env)
        mp = load_map("cpm_entire")
        p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True,
                       is_obs_noise=False, is_apply_mask=False, nom_controller_type=nominal, adaptive_lambda=adaptive)
        cfg = make_config(p, mp, B)
        dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
        seg_l, seg_r = cbf.load_segment_tables(mp)
        for e in (dev, ora):
            e.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        dev.env.buffer(capi.BUF_DONE).fill_(1)
        ora.get(capi.BUF_DONE, copy=False)[:] = 1
        dev.auto_reset(4, 0, mp.list_first[0], mp.list_count[0])
        ora.auto_reset(4, 0, mp.list_first[0], mp.list_count[0])
        act = np.random.default_rng(8).uniform(-0.6, 1.2, (B, N, 2)).astype(np.float32)
    else:
        ora, _ = qp_case(ob.OracleEnv, N=N, B=B, nominal=nominal, adaptive_lambda=adaptive)
    safe_d, u_d, info_d = dev.cbf_qp(act)
    safe_o, u_o, info_o, con, unom = ora.cbf_qp(act, with_data=True)
    assert info_d[:, 1].all() and info_o[:, 1].all(), (info_d[:, 1].sum(), info_o[:, 1].sum())
    assert info_d[:, 0].max() <= 60
    assert np.abs(u_d - u_o).max() <= 1e-7, np.abs(u_d - u_o).max()
    assert np.abs(safe_d - safe_o).max() <= 1e-6
    check_kkt(ora, u_d, con, unom, nominal, tol=1e-8)  # the HIP minimiser against the (oracle-computed) problem data
    assert N == 1 or (np.abs(u_d - unom).max(axis=(1, 2)) > 1e-6).sum() >= 3
    dev.close()
    ora.close()


@pytest.mark.parametrize("apply,nominal", [(False, "rl"), (True, "rl"), (True, "clf")])
def test_cbf_qp_rollout_reward_vs_oracle(apply, nominal):
    """rew_method "cbf_sparse" with is_solve_qp=True: the QP before every step, the step penalises |applied - nominal| action
    (road_traffic.py:1112-1139); with is_apply_cbf_action the safe action is what the env steps with."""
    import torch
    from test_gpu_parity import _compare_all

    N, B = 8, 20
    p = Parameters(n_agents=N, scenario_type="cpm_entire", rew_method="cbf_sparse", dt=0.05, is_solve_qp=True, is_using_cbf_training=True,
                   is_apply_cbf_action=apply, nom_controller_type=nominal, is_apply_mask=False, is_obs_noise=False, max_steps=6, is_use_mtv_distance=False)
    mp = load_map("cpm_entire")
    cfg = make_config(p, mp, B)
    assert cfg.rew_flags & capi.REW_CBF_QP and not cfg.rew_flags & capi.REW_CBF
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    cc = cbf.make_cbf_config(p)
    assert cc.is_apply_cbf_action == int(apply)
    for e in (dev, ora):
        e.cbf_attach(cc, seg_l, seg_r)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    pf, pc = mp.list_first[0], mp.list_count[0]
    dev.auto_reset(6, 0, pf, pc)
    ora.auto_reset(6, 0, pf, pc)
    rng = np.random.default_rng(21)
    changed = 0
    for t in range(6):
        act = np.stack([rng.uniform(-0.7, 1.3, (B, N)), rng.uniform(-0.7, 0.7, (B, N))], axis=-1).astype(np.float32)
        safe_d, u_d, info_d = dev.cbf_qp(act)
        safe_o, u_o, info_o = ora.cbf_qp(act)
        assert info_d[:, 1].all() and info_o[:, 1].all()
        assert np.abs(u_d - u_o).max() <= 1e-7 and np.abs(safe_d - safe_o).max() <= 1e-6
        assert np.abs(dev.get(capi.BUF_CBF_NOMINAL) - ora.get(capi.BUF_CBF_NOMINAL)).max() <= 1e-6
        changed += int((np.abs(safe_o - np.clip(act, [-0.5, -0.5411], [1.0, 0.5411])).max(axis=(1, 2)) > 1e-5).sum())
        # continue both sides from the ORACLE's QP result so that a last-bit difference cannot leak into the state comparison
        dev.env.buffer(capi.BUF_CBF_NOMINAL).copy_(torch.from_numpy(ora.get(capi.BUF_CBF_NOMINAL)))
        step_act = safe_o if apply else act
        dev.step(step_act)
        ora.step(step_act)
        _compare_all(dev, ora, f"step {t}")
        dev.auto_reset(6, t + 1, pf, pc)
        ora.auto_reset(6, t + 1, pf, pc)
    assert changed > 0
    dev.close()
    ora.close()


@pytest.mark.parametrize("N", [24, 40, 41])
def test_cbf_qp_dense_cluster_uses_the_full_system(N):
    """24 (40, 41: the packed-Hessian variant, with an even and an odd number of words) vehicles piled within half a metre: (almost) every vehicle is coupled to others through active pair rows, more than
    the 16 the compacted register factorisation takes -- the full-system LDS path; same minimiser as the oracle, KKT conditions hold."""
    from test_cbf_qp import check_kkt

    z, meta = _cbf_fixture()
    B = 6
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True,
                   is_obs_noise=False, is_apply_mask=False)
    cfg = make_config(p, mp, B)
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    rng = np.random.default_rng(9)
    st8 = np.zeros((B, N, 8), np.float32)
    ids = np.zeros((B, N, 4), np.int32)
    for b in range(B):
        st8[b, :, :5] = z["p2_state"][b, 0]                                   # everybody starts from vehicle 0's pose ...
        st8[b, :, 0:2] += rng.uniform(-0.25, 0.25, (N, 2)).astype(np.float32)  # ... scattered within half a metre
        st8[b, :, 2] += rng.uniform(-0.5, 0.5, N).astype(np.float32)
        st8[b, :, 3] = rng.uniform(0.0, 1.0, N).astype(np.float32)
        ids[b, :, 0] = z["p2_path"][b, 0]
    ids[..., 2] = ids[..., 0]
    for e in (dev, ora):
        e.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        e.reset(np.repeat(np.arange(B), N), np.tile(np.arange(N), B), ids.reshape(-1, 4), st8.reshape(-1, 8), 1)
    act = rng.uniform(-0.3, 1.0, (B, N, 2)).astype(np.float32)
    safe_d, u_d, info_d = dev.cbf_qp(act)
    safe_o, u_o, info_o, con, unom = ora.cbf_qp(act, with_data=True)
    assert info_d[:, 1].all() and info_o[:, 1].all(), (info_d, info_o)
    # coupled vehicles per env (from the oracle's active pair rows at its solution): the case must exceed the compact limit somewhere
    n_lane = N * 3 * 2
    worst_coupled = 0
    for b in range(B):
        cpl = set()
        for row in con[b][n_lane:]:
            i, j = int(row[0]), int(row[1])
            g = row[2:6] @ u_o[b].reshape(-1)[[2 * i, 2 * i + 1, 2 * j, 2 * j + 1]] + row[6] + max(row[7], 0.0)
            if g < 0:
                cpl.update((i, j))
        worst_coupled = max(worst_coupled, len(cpl))
    assert worst_coupled > 16, worst_coupled
    assert np.abs(u_d - u_o).max() <= 1e-6, np.abs(u_d - u_o).max()
    check_kkt(ora, u_d, con, unom, "rl", tol=1e-7)
    dev.close()
    ora.close()


def test_cbf_qp_mixed_batch_of_lean_and_deferred_envs():
    """Round 6: up to 32 vehicles the LEAN instantiation solves an env unless its register path cannot take it (more than 8 vehicles in candidate pair rows); such an env
    is flagged and solved by the full-layout launch that follows, one workgroup per span of 16 envs.  A batch over three spans in which every other env is a dense pile
    (deferred) and the rest are spread out (solved by the lean launch): both kinds equal the oracle, and the flags are cleared -- a second call on the same handle with
    every env spread out, and a third with the piles back, are right as well."""
    z, meta = _cbf_fixture()
    B, N = 40, 24
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_obs_noise=False, is_apply_mask=False)
    cfg = make_config(p, mp, B)
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    rng = np.random.default_rng(17)
    ids = np.zeros((B, N, 4), np.int32)

    def states(dense_mask):
        st8 = np.zeros((B, N, 8), np.float32)
        for b in range(B):
            src = b % z["p2_state"].shape[0]
            st8[b, :, :5] = z["p2_state"][src, 0]
            spread = 0.25 if dense_mask[b] else 2.0  # a pile within half a metre, or vehicles metres apart
            st8[b, :, 0:2] += rng.uniform(-spread, spread, (N, 2)).astype(np.float32)
            st8[b, :, 2] += rng.uniform(-0.5, 0.5, N).astype(np.float32)
            st8[b, :, 3] = rng.uniform(0.0, 1.0, N).astype(np.float32)
            ids[b, :, 0] = z["p2_path"][src, 0]
        ids[..., 2] = ids[..., 0]
        return st8

    for e in (dev, ora):
        e.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
    n_iter_dense = []
    for dense in (np.arange(B) % 2 == 0, np.zeros(B, bool), np.arange(B) % 3 == 0):
        st8 = states(dense)
        for e in (dev, ora):
            e.reset(np.repeat(np.arange(B), N), np.tile(np.arange(N), B), ids.reshape(-1, 4), st8.reshape(-1, 8), 1)
        act = rng.uniform(-0.3, 1.0, (B, N, 2)).astype(np.float32)
        safe_d, u_d, info_d = dev.cbf_qp(act)
        safe_o, u_o, info_o = ora.cbf_qp(act)[:3]
        assert np.array_equal(info_d[:, 1], info_o[:, 1]) and info_o[:, 1].all(), (info_d[:, 1], info_o[:, 1])
        assert np.abs(u_d - u_o).max() <= 1e-6 and np.abs(safe_d - safe_o).max() <= 1e-6, (np.abs(u_d - u_o).max(), np.flatnonzero(np.abs(u_d - u_o).max(axis=(1, 2)) > 1e-6))
        n_iter_dense.append(int(info_d[dense, 0].sum()) if dense.any() else 0)
    assert n_iter_dense[0] > 0 and n_iter_dense[2] > 0
    dev.close()
    ora.close()


def test_cbf_qp_is_bitwise_repeatable():
    """The candidate list is compacted in row order and the Newton phase runs on one wavefront (its LDS atomics execute in program / lane
    order): repeated launches on the same state return the same bits."""
    import torch
    from test_cbf_qp import qp_case

    dev, act = qp_case(_hip_env, N=16, B=48)
    a = torch.as_tensor(act).cuda()
    outs = []
    for _ in range(4):
        u = torch.zeros((48, 16, 2), dtype=torch.float64, device="cuda")
        safe = dev.env.cbf_qp(a, None, u, None)
        dev.env.sync()
        outs.append((u.clone(), safe.clone()))
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
    dev.close()


def test_cbf_qp_hip_minimiser_vs_independent_solver_of_the_original_problem():
    """u of the HIP kernel == the interior-point solution of the ORIGINAL 2416-variable problem (tests/qp_original.py) within 1e-5 on the 48
    envs of the set-state fixture (16 agents, 3 circles, lambda penalty)."""
    from test_cbf_qp import compare_with_original_problem, qp_case

    dev, act = qp_case(_hip_env, N=16, B=48, nominal="rl", adaptive_lambda=True)
    ora, _ = qp_case(ob.OracleEnv, N=16, B=48, nominal="rl", adaptive_lambda=True)
    _, u_d, info_d = dev.cbf_qp(act)
    _, _, _, con, unom = ora.cbf_qp(act, with_data=True)
    assert info_d[:, 1].all()
    worst = compare_with_original_problem(ora, u_d, con, unom, "rl", 48)
    dev.close()
    ora.close()
    assert worst <= 1e-5, worst


def test_cbf_qp_full_size_4096_envs():
    """BASELINE config 5 at its size (16 agents x 4096 envs): every env's QP converges, the minimisers of the first and the last 96 envs equal
    the oracle's on the same states, and they pass the KKT check of the original problem."""
    import torch
    from sigmarl_amd.env import SigmaEnv
    from test_cbf_qp import check_kkt

    B, N, S = 4096, 16, 96
    p = Parameters(n_agents=N, scenario_type="cpm_entire", rew_method="cbf", dt=0.05, is_solve_qp=True, is_using_cbf_training=True,
                   is_apply_mask=False, is_obs_noise=False, is_use_mtv_distance=False, adaptive_lambda=True)
    env = SigmaEnv(p, n_envs=B, device="cuda:0")
    env.reset_random(seed=12)
    env.cbf_attach()
    g = torch.Generator(device="cuda").manual_seed(5)
    safe = torch.empty((B, N, 2), device="cuda")
    u = torch.empty((B, N, 2), dtype=torch.float64, device="cuda")
    info = torch.empty((B, 2), dtype=torch.int32, device="cuda")
    for t in range(4):
        act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.4 - 0.2, torch.rand(B, N, generator=g, device="cuda") * 1.0 - 0.5], -1).contiguous()
        env.cbf_qp(act, safe, u, info)
        if t < 3:
            env.step_autoreset(act, seed=3)
    env.sync()
    assert bool(info[:, 1].all()), int((info[:, 1] == 0).sum())
    assert int(info[:, 0].max()) <= 60
    assert bool(torch.isfinite(u).all()) and bool(torch.isfinite(safe).all())
    mp = env.map
    seg_l, seg_r = cbf.load_segment_tables(mp)
    n_changed = 0
    for lo in (0, B - S):
        ora = ob.OracleEnv(make_config(p, mp, S), mp)
        ora.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        st = env.buffer(capi.BUF_STATE)[lo:lo + S].cpu().numpy()
        pa = env.buffer(capi.BUF_PATH)[lo:lo + S].cpu().numpy()
        ora.reset(np.repeat(np.arange(S), N), np.tile(np.arange(N), S), pa.reshape(-1, 4), st.reshape(-1, 8), 1)
        safe_o, u_o, info_o, con, unom = ora.cbf_qp(act[lo:lo + S].cpu().numpy(), with_data=True)
        u_d = u[lo:lo + S].cpu().numpy()
        assert np.abs(u_d - u_o).max() <= 1e-7, (lo, np.abs(u_d - u_o).max())
        assert np.abs(safe[lo:lo + S].cpu().numpy() - safe_o).max() <= 1e-6
        check_kkt(ora, u_d, con, unom, "rl", tol=1e-8)
        n_changed += int((np.abs(u_d - unom).max(axis=(1, 2)) > 1e-6).sum())
        ora.close()
    assert n_changed >= 10
    env.close()


# ---- grouped CBF-QPs (Parameters.is_grouping_agents; sigmarl/cbf_qp.py:193-310, 1562-2281) ---------------------------------------------
@pytest.mark.parametrize("case", [(2, 0.5, "rl"), (3, 0.5, "rl"), (4, 1.0, "rl"), (2, 1.0, "clf"), (5, 0.5, "clf")])
def test_grouped_qp_matches_oracle_and_reference_groups(case):
    """HIP == oracle for the grouped problems: identical groups (the reference's, tests/golden/cbf_grouped.npz), minimiser within 1e-7,
    safe actions and the nominal-action record within 1e-6."""
    import test_cbf_grouped as tg

    m, rng, nominal = case
    z, _ = tg.grouped_fixture()
    k0 = [c[0] for c in tg.fixture_cases() if c[1:] == (m, rng, nominal)][0]
    outs = []
    for make in (ob.OracleEnv, _hip_env):
        env, act, ref = tg.grouped_case(make, m, rng, nominal)
        st = env.get(capi.BUF_SHORT_TERM, copy=False) if make is ob.OracleEnv else None
        if st is not None:
            st[:, :, 2, :] = ref
        else:
            env.env.buffer(capi.BUF_SHORT_TERM)[:, :, 2, :] = __import__("torch").as_tensor(ref).to(env.env.device)
        safe, u, info = env.cbf_qp(act)[:3]
        outs.append((safe, u, info, env.cbf_groups(), env.get(capi.BUF_CBF_NOMINAL).copy()))
        env.close()
    (s0, u0, i0, g0, n0), (s1, u1, i1, g1, n1) = outs
    assert np.array_equal(g0, g1) and np.array_equal(g1, z["grp"][k0:k0 + len(g1)])
    assert i1[:, 1].all()
    assert np.abs(u0 - u1).max() <= 1e-7
    assert np.abs(s0 - s1).max() <= 1e-6 and np.abs(n0 - n1).max() <= 1e-6


def test_grouped_qp_against_the_original_problems_and_regroup():
    """The HIP minimiser of the grouped problems == the interior-point solution of the original problems (every slack / lambda of every group
    problem explicit; tests/qp_original.py) within 1e-5, with the row data taken from the oracle (pinned on the reference's); the groups are kept
    across calls and re-formed after sigmaenv_cbf_regroup; repeated launches return the same bits."""
    import test_cbf_grouped as tg

    m, rng, nominal = 3, 1.0, "rl"
    envo, act, ref = tg.grouped_case(ob.OracleEnv, m, rng, nominal)
    _, _, _, con, unom = envo.cbf_qp(act, with_data=True)
    env, _, _ = tg.grouped_case(_hip_env, m, rng, nominal)
    safe, u, info = env.cbf_qp(act)
    assert info[:, 1].all()
    worst = 0.0
    for b in range(env.B):
        x = tg.solve_grouped_original(envo, con[b], unom[b], nominal, b)
        worst = max(worst, float(np.abs(x - u[b].reshape(-1)).max()))
    assert worst <= 1e-5, worst
    safe2, u2, _ = env.cbf_qp(act)
    assert np.array_equal(u.view(np.uint64), u2.view(np.uint64))
    g0 = env.cbf_groups()
    import torch
    st = env.env.buffer(capi.BUF_STATE)
    st[:, :, 0:2] = torch.flip(st[:, :, 0:2], dims=[1])
    env.cbf_qp(act)
    assert np.array_equal(env.cbf_groups(), g0)
    env.cbf_regroup()
    env.cbf_qp(act)
    assert not np.array_equal(env.cbf_groups(), g0)
    envo.close(); env.close()


def test_grouped_qp_full_size_4096_envs():
    """16 agents x 4096 envs, groups of 4, observation range 0.75 m: every env converges; groups (formed on the device) and minimisers of the
    first and last 96 envs equal the oracle's on the same states; a few steps with the safe actions applied keep the groups."""
    import torch
    from sigmarl_amd.env import SigmaEnv

    B, N, S = 4096, 16, 96
    p = Parameters(n_agents=N, scenario_type="cpm_entire", rew_method="cbf", dt=0.05, is_solve_qp=True, is_using_cbf_training=True, is_apply_mask=False,
                   is_obs_noise=False, is_use_mtv_distance=False, adaptive_lambda=True, is_grouping_agents=True, max_group_size=4, observation_range=0.75)
    env = SigmaEnv(p, n_envs=B, device="cuda:0")
    env.reset_random(seed=13)
    env.cbf_attach()
    g = torch.Generator(device="cuda").manual_seed(6)
    safe = torch.empty((B, N, 2), device="cuda")
    u = torch.empty((B, N, 2), dtype=torch.float64, device="cuda")
    info = torch.empty((B, 2), dtype=torch.int32, device="cuda")
    st0 = env.buffer(capi.BUF_STATE)[:, :, 0:2].cpu().numpy().copy()
    for t in range(3):
        act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.4 - 0.2, torch.rand(B, N, generator=g, device="cuda") * 1.0 - 0.5], -1).contiguous()
        env.cbf_qp(act, safe, u, info)
        if t < 2:
            env.step(safe)  # (no resets: the groups belong to the vehicles of the run)
    env.sync()
    assert bool(info[:, 1].all()), int((info[:, 1] == 0).sum())
    assert bool(torch.isfinite(u).all()) and bool(torch.isfinite(safe).all())
    groups = env.cbf_groups()
    assert all(np.bincount(groups[b], minlength=4).tolist() == [4, 4, 4, 4] for b in range(0, B, 97))
    mp = env.map
    seg_l, seg_r = cbf.load_segment_tables(mp)
    for lo in (0, B - S):
        ora = ob.OracleEnv(make_config(p, mp, S), mp)
        ora.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        pa = env.buffer(capi.BUF_PATH)[lo:lo + S].cpu().numpy()
        st = env.buffer(capi.BUF_STATE)[lo:lo + S].cpu().numpy()
        first = st.copy()
        first[:, :, 0:2] = st0[lo:lo + S]  # the oracle forms its groups on the states of the first call, as the device did
        ora.reset(np.repeat(np.arange(S), N), np.tile(np.arange(N), S), pa.reshape(-1, 4), first.reshape(-1, 8), 1)
        ora.cbf_qp(act[lo:lo + S].cpu().numpy())
        assert np.array_equal(ora.cbf_groups(), groups[lo:lo + S])
        ora.reset(np.repeat(np.arange(S), N), np.tile(np.arange(N), S), pa.reshape(-1, 4), st.reshape(-1, 8), 1)
        safe_o, u_o, info_o = ora.cbf_qp(act[lo:lo + S].cpu().numpy())
        assert np.abs(u[lo:lo + S].cpu().numpy() - u_o).max() <= 1e-7
        assert np.abs(safe[lo:lo + S].cpu().numpy() - safe_o).max() <= 1e-6
        ora.close()
    env.close()


@pytest.mark.parametrize("tag", ["cycle", "crawl0", "crawl1", "crawl2", "crawl3", "crawl4", "noisefloor", "valley", "wall"])
def test_solver_regressions_hip_vs_oracle(tag):
    """The instances of tests/data/qp_regressions.npz (found by tools/fuzz_cbf.py: a 2-cycle of the noise-tolerant acceptance rule, a
    variable creeping towards a bound with a collapsing line search, a stop test below the gradient's noise floor, a zig-zag along a valley
    shared with a variable inside the epsilon band of a bound): the HIP solver converges
    and returns the oracle's minimiser, which tests/test_cbf_qp.py holds against the interior-point solution of the original problem."""
    from test_cbf_qp import regression_case

    outs = []
    for make in (ob.OracleEnv, _hip_env):
        env, act, short, kw = regression_case(make, tag)
        if make is ob.OracleEnv:
            env.get(capi.BUF_SHORT_TERM, copy=False)[:] = short
        else:
            import torch
            env.env.buffer(capi.BUF_SHORT_TERM)[:] = torch.as_tensor(short).to(env.env.device)
        safe, u, info = env.cbf_qp(act)[:3]
        outs.append((safe, u, info))
        env.close()
    (s0, u0, i0), (s1, u1, i1) = outs
    assert i0[0, 1] == 1 and i1[0, 1] == 1, (i0, i1)
    assert np.abs(u0 - u1).max() <= 1e-7 and np.abs(s0 - s1).max() <= 1e-6


def test_fuzz_instance_hip_equals_the_interior_point_solution():
    """The QP solution pinned against GROUND TRUTH instead of against the oracle: on the instance of the randomised differential run where HIP and
    oracle once differed by 1.4e-7 (tests/data/qp_fuzz_instance.npz), the HIP minimiser equals the interior-point solution of the ORIGINAL grouped
    problems (every slack and lambda explicit, tests/qp_original.py) within 1e-6 -- an order of magnitude inside OSQP's own 1e-5."""
    import torch
    from test_cbf_qp import FUZZ_IP_TOL, fuzz_instance, fuzz_instance_ground_truth

    ora, u_ora, x = fuzz_instance_ground_truth()
    dev, act, short, kw = fuzz_instance(_hip_env)
    dev.env.buffer(capi.BUF_SHORT_TERM)[:] = torch.as_tensor(short).to(dev.env.device)
    safe, u, info = dev.cbf_qp(act)[:3]
    assert info[:, 1].all() and np.array_equal(dev.cbf_groups(), ora.cbf_groups())
    err = np.abs(x - u.reshape(len(u), -1)).max(axis=1)
    print("HIP vs interior point per env:", " ".join(f"{e:.1e}" for e in err), "| HIP vs oracle:", f"{np.abs(u - u_ora).max():.1e}")
    assert err.max() <= FUZZ_IP_TOL
    dev.close()
    ora.close()


def test_grouped_qp_unknown_count_not_a_power_of_two():
    """9 vehicles in groups of 6 / 3 (18 unknowns): the Cholesky factor of the compacted system needs a 32 x 32 area in LDS although the
    Hessian is 18 x 18 (the kernel once overran it and hung)."""
    import test_cbf_grouped as tg

    for m in (6, 3):
        outs = []
        for make in (ob.OracleEnv, _hip_env):
            env, act, ref = tg.grouped_case(make, m, 0.5, "rl", N=9)
            safe, u, info = env.cbf_qp(act)[:3]
            outs.append((u, info, env.cbf_groups()))
            env.close()
        assert outs[0][1][:, 1].all() and outs[1][1][:, 1].all()
        assert np.array_equal(outs[0][2], outs[1][2]) and np.abs(outs[0][0] - outs[1][0]).max() <= 1e-7


def test_grouped_qp_beyond_32_vehicles():
    """40 vehicles in groups of at most 6 (grouped CBF-QPs, cbf_qp.py:1562-2281) on sampled starts: the groups and the minimiser of the packed-Hessian variant
    equal the oracle's."""
    N, B = 40, 6
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_obs_noise=False,
                   is_apply_mask=False, adaptive_lambda=True, is_grouping_agents=True, max_group_size=6, observation_range=0.5)
    cfg = make_config(p, mp, B)
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    for e in (dev, ora):
        e.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    dev.auto_reset(5, 0, mp.list_first[0], mp.list_count[0])
    ora.auto_reset(5, 0, mp.list_first[0], mp.list_count[0])
    act = np.random.default_rng(11).uniform(-0.6, 1.2, (B, N, 2)).astype(np.float32)
    safe_d, u_d, info_d = dev.cbf_qp(act)[:3]
    safe_o, u_o, info_o = ora.cbf_qp(act)[:3]
    assert np.array_equal(dev.cbf_groups(), ora.cbf_groups())
    assert info_d[:, 1].all() and info_o[:, 1].all(), (info_d, info_o)
    assert np.abs(u_d - u_o).max() <= 1e-7, np.abs(u_d - u_o).max()
    assert np.abs(safe_d - safe_o).max() <= 1e-6
    dev.close()
    ora.close()


def test_packed_hessian_with_an_odd_vehicle_count():
    """41 vehicles (tests/data/qp_big_odd_instance.npz, found by tools/fuzz_cbf.py --cpm-agents 48): the packed lower triangle of the <BIG> instantiation has
    N (2 N + 1) words -- odd for an odd vehicle count -- and its last word (the last unknown's diagonal entry) was not cleared between the evaluations: the solve
    depended on what the LDS held before (13 ... 100 iterations from run to run, one env not converged).  HIP == oracle, twice, with the oracle's iteration counts."""
    import torch
    from sigmarl_amd.maps import load_map
    from sigmarl_amd.params import make_config

    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "qp_big_odd_instance.npz"), allow_pickle=True)
    kw = eval(str(d["kw"]))
    B, N = d["state"].shape[:2]
    assert N == 41
    mp = load_map(kw["scenario_type"])
    p = Parameters(**kw)
    outs = []
    for make in (ob.OracleEnv, _hip_env, _hip_env):
        env = make(make_config(p, mp, B), mp)
        seg_l, seg_r = cbf.load_segment_tables(mp)
        env.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
        env.reset(np.repeat(np.arange(B), N).astype(np.int32), np.tile(np.arange(N), B).astype(np.int32), d["path"].reshape(-1, 4), d["state"].reshape(-1, 8), 1)
        if make is ob.OracleEnv:
            env.get(capi.BUF_SHORT_TERM, copy=False)[:] = d["short"]
        else:
            env.env.buffer(capi.BUF_SHORT_TERM)[:] = torch.as_tensor(d["short"]).to(env.env.device)
        safe, u, info = env.cbf_qp(d["act"].astype(np.float32))[:3]
        outs.append((u, info))
        env.close()
    (u_o, i_o), (u_1, i_1), (u_2, i_2) = outs
    assert i_o[:, 1].all() and i_1[:, 1].all() and i_2[:, 1].all(), (i_o, i_1, i_2)
    assert np.array_equal(i_1, i_2) and np.array_equal(u_1, u_2)
    assert np.array_equal(i_1[:, 0], i_o[:, 0]), (i_1[:, 0], i_o[:, 0])
    assert np.abs(u_1 - u_o).max() <= 1e-7
