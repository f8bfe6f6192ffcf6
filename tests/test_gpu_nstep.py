"""``sigmaenv_step_autoreset_n`` (the reference's rollout loop over a chunk of steps, helper_training.py:687-788, as ONE launch in which every
wavefront walks its env tile through the steps): bit-identical to n calls of ``sigmaenv_step_autoreset`` -- every buffer, every record row,
every reset draw -- and, through the oracle, to the reference path.  Also the scan's work-list bound for tiles of one and two agent slots."""
import numpy as np
import pytest

import oracle_binding as ob
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config
from test_gpu_parity import FLT_BUFS, INT_BUFS, _compare_all, _hip_env

pytestmark = pytest.mark.gpu


def _actions(rng, T, B, N, wild):
    if wild:
        a = np.stack([rng.uniform(-0.2, 1.3, (T, B, N)), rng.uniform(-0.7, 0.7, (T, B, N))], axis=-1)
        a[2::3] = np.stack([rng.uniform(0.0, 0.3, (T, B, N)), rng.uniform(-0.05, 0.05, (T, B, N))], axis=-1)[2::3]
    else:
        a = np.stack([rng.uniform(0, 1, (T, B, N)), rng.uniform(-0.25, 0.25, (T, B, N))], axis=-1)
    return a.astype(np.float32)


NSTEP_CASES = [
    # scenario, N, B, mtv, rew_method, dt, testing, T, wild actions
    ("cpm_entire", 16, 4096, False, "distance", 0.05, False, 6, False),      # BASELINE config 2
    ("cpm_entire", 32, 8192, False, "distance", 0.05, False, 3, False),      # config 4's shape
    ("cpm_entire", 16, 200, True, "ttc_sparse", 0.1, False, 12, True),
    ("cpm_entire", 5, 33, False, "sparse", 0.05, False, 9, True),            # ragged tile (generic instantiation)
    ("cpm_entire", 8, 24, True, "sparse", 0.1, True, 12, True),              # testing mode: per-agent re-placement inside the loop
    ("cpm_entire", 4, 64, False, "distance", 0.05, False, 10, True),         # 4 x 4 tile
    ("intersection_1", 4, 40, False, "distance", 0.1, False, 16, True),      # non-loop map: entry / exit requests
    ("on_ramp_1", 6, 40, True, "ttc", 0.1, False, 16, True),
]


@pytest.mark.parametrize("scen,N,B,mtv,rew,dt,testing,T,wild", NSTEP_CASES)
def test_nstep_launch_equals_n_single_launches(scen, N, B, mtv, rew, dt, testing, T, wild):
    import torch
    from sigmarl_amd.shard import slab_width

    p = Parameters(n_agents=N, scenario_type=scen, is_use_mtv_distance=mtv, rew_method=rew, dt=dt, is_testing_mode=testing,
                   is_apply_mask=False, is_obs_noise=False, max_steps=9)
    mp = load_map(scen)
    cfg = make_config(p, mp, B)
    pf, pc = mp.list_first[0], mp.list_count[0]
    one, many = _hip_env(cfg, mp), _hip_env(cfg, mp)
    for d in (one, many):
        d.env.buffer(capi.BUF_DONE).fill_(1)
        d.auto_reset(5, 0, pf, pc)
    W = slab_width(N, one.env.D)
    acts = torch.as_tensor(_actions(np.random.default_rng(11), T, B, N, wild)).cuda()
    rec_one = torch.full((T, B, W), float("nan"), device="cuda")
    rec_many = torch.full((T, B, W), float("nan"), device="cuda")
    done_total = 0
    for t in range(T):
        one.env.set_slab(rec_one[t])
        one.env.step_autoreset(acts[t], 5, 100 + t, pf, pc)
        done_total += int(one.env.buffer(capi.BUF_TIMER)[:, 3].sum().item())
    one.env.set_slab(None)
    many.env.step_autoreset_n(acts, rec_many, 5, 100, pf, pc)
    many.env.sync()
    assert torch.equal(rec_one.view(torch.int32), rec_many.view(torch.int32)), "record rows differ"
    for w in INT_BUFS + FLT_BUFS:
        a, b = one.get(w), many.get(w)
        assert a.tobytes() == b.tobytes(), f"buffer {w} differs between {T} launches and the one {T}-step launch"
    assert done_total > 0  # envs finished and were re-placed inside the loop
    # a second chunk continues exactly where the first one stopped (stride 0: the same action block every step; no record)
    for t in range(3):
        one.env.step_autoreset(acts[0], 5, 200 + t, pf, pc)
    many.env.step_autoreset_n_ptr(acts.data_ptr(), 3, 0, 0, 0, 5, 200, pf, pc)
    for w in INT_BUFS + FLT_BUFS:
        assert one.get(w).tobytes() == many.get(w).tobytes(), f"buffer {w} differs after the second chunk"
    one.close()
    many.close()


@pytest.mark.parametrize("scen,N,B,mtv,rew,T", [("cpm_entire", 16, 256, False, "distance", 10), ("intersection_1", 4, 64, True, "ttc", 12)])
def test_nstep_launch_against_the_oracle(scen, N, B, mtv, rew, T):
    """The n-step launch against the CPU oracle stepping and resetting one call at a time: masks / indices bit-exact, fp32 within 1e-5, and
    every record row == the oracle's terminal observation / reward / done of that step."""
    import torch
    from sigmarl_amd.shard import slab_width, unpack_slab

    p = Parameters(n_agents=N, scenario_type=scen, is_use_mtv_distance=mtv, rew_method=rew, dt=0.1, is_apply_mask=False, is_obs_noise=False, max_steps=9)
    mp = load_map(scen)
    cfg = make_config(p, mp, B)
    pf, pc = mp.list_first[0], mp.list_count[0]
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    dev.auto_reset(3, 0, pf, pc)
    ora.auto_reset(3, 0, pf, pc)
    acts = _actions(np.random.default_rng(2), T, B, N, True)
    rec = torch.full((T, B, slab_width(N, dev.env.D)), float("nan"), device="cuda")
    dev.env.step_autoreset_n(torch.as_tensor(acts).cuda(), rec, 3, 50, pf, pc)
    obs, rew_, dn = unpack_slab(rec, N, dev.env.D)
    n_done = 0
    for t in range(T):
        ora.step(acts[t])
        assert np.abs(obs[t].cpu().numpy() - ora.get(capi.BUF_OBS)).max() <= 1e-5
        assert np.abs(rew_[t].cpu().numpy() - ora.get(capi.BUF_REWARD)).max() <= 1e-5
        assert np.array_equal(dn[t].cpu().numpy(), ora.get(capi.BUF_DONE).astype(bool))
        n_done += int(ora.get(capi.BUF_DONE).sum())
        ora.auto_reset(3, 50 + t, pf, pc)
    _compare_all(dev, ora, f"after the {T}-step launch")
    assert n_done > 0
    dev.close()
    ora.close()


@pytest.mark.parametrize("scen,N,B,testing", [("cpm_entire", 16, 96, False), ("intersection_1", 4, 40, False), ("cpm_entire", 8, 24, True)])
def test_observation_noise_on_the_device(scen, N, B, testing):
    """is_obs_noise (the reference's default): the uniform noise of observation_provider_rt.py:613-618 is added ON THE DEVICE -- observation buffer,
    rollout record, what the on-device actor reads.  HIP == oracle (same draws), the T-step launch == T launches bit for bit, and the noise has
    the reference's scaling (range [0, level), mean level / 2)."""
    import torch
    from sigmarl_amd.shard import slab_width, unpack_slab

    T, level = 8, 0.05
    mp = load_map(scen)
    pf, pc = mp.list_first[0], mp.list_count[0]
    kw = dict(n_agents=N, scenario_type=scen, is_use_mtv_distance=False, rew_method="distance", dt=0.1, is_apply_mask=False, max_steps=9, is_testing_mode=testing)
    cfg = make_config(Parameters(is_obs_noise=True, obs_noise_level=level, random_seed=11, **kw), mp, B)
    cfg0 = make_config(Parameters(is_obs_noise=False, **kw), mp, B)
    assert abs(cfg.obs_noise_level - level) < 1e-9 and cfg0.obs_noise_level == 0.0
    one, many, clean, ora = _hip_env(cfg, mp), _hip_env(cfg, mp), _hip_env(cfg0, mp), ob.OracleEnv(cfg, mp)
    for d in (one, many, clean):
        d.env.buffer(capi.BUF_DONE).fill_(1)
        d.auto_reset(5, 0, pf, pc)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    ora.auto_reset(5, 0, pf, pc)
    _compare_all(one, ora, "noisy initial observation")
    acts = _actions(np.random.default_rng(4), T, B, N, True)
    W = slab_width(N, one.env.D)
    rec_one = torch.full((T, B, W), float("nan"), device="cuda")
    rec_many = torch.full((T, B, W), float("nan"), device="cuda")
    rec_clean = torch.full((T, B, W), float("nan"), device="cuda")
    ta = torch.as_tensor(acts).cuda()
    for t in range(T):
        one.env.set_slab(rec_one[t])
        one.env.step_autoreset(ta[t], 5, 100 + t, pf, pc)
        ora.step(acts[t])
        obs_t = unpack_slab(rec_one[t], N, one.env.D)[0].cpu().numpy()
        assert np.abs(obs_t - ora.get(capi.BUF_OBS)).max() <= 1e-5, f"noisy record row, step {t}"
        ora.auto_reset(5, 100 + t, pf, pc)
        _compare_all(one, ora, f"noisy step {t}")
    many.env.step_autoreset_n(ta, rec_many, 5, 100, pf, pc)
    clean.env.step_autoreset_n(ta, rec_clean, 5, 100, pf, pc)
    many.env.sync()
    assert torch.equal(rec_one.view(torch.int32), rec_many.view(torch.int32))
    for w in INT_BUFS + FLT_BUFS:
        assert one.get(w).tobytes() == many.get(w).tobytes(), f"buffer {w}"
    # the noise itself: noisy record - noise-free record (same states: the noise touches nothing but the observation)
    d = (unpack_slab(rec_many, N, one.env.D)[0] - unpack_slab(rec_clean, N, one.env.D)[0]).double().cpu().numpy().ravel()
    assert d.min() >= -1e-6 and d.max() < level + 1e-6 and abs(d.mean() - level / 2) < 1e-3 and abs(d.var() - level ** 2 / 12) < 1e-4
    for w in INT_BUFS + [capi.BUF_STATE, capi.BUF_REWARD]:
        assert many.get(w).tobytes() == clean.get(w).tobytes()
    for e in (one, many, clean, ora):
        e.close()


NSTEP_OBS_VARIANTS = [
    # (scenario, N, B, Parameters switches): every family of the non-default observation rows, in the T-step launch with the rollout record
    ("cpm_entire", 16, 160, dict(is_ego_view=False)),                                                              # bird view
    ("roundabout_2", 6, 64, dict(is_ego_view=False, is_apply_mask=True, is_observe_vertices=False)),               # bird view + the lanelet-relation mask
    ("cpm_entire", 8, 96, dict(is_apply_mask=True, is_obs_steering=True, is_observe_ref_path_other_agents=True)),  # steering + neighbours' reference paths + distance mask
    ("on_ramp_1", 4, 48, dict(is_observe_distance_to_boundaries=False, is_using_opponent_modeling=True, is_testing_mode=True)),  # boundary points (fresh / stepped shifts, per-agent re-placement) + placeholders
    ("cpm_entire", 8, 64, dict(is_observe_vertices=False, is_observe_distance_to_agents=False, is_observe_distance_to_center_line=False, is_obs_noise=True)),
    ("intersection_1", 4, 40, dict(is_ego_view=False, is_partial_observation=False, is_obs_noise=True)),           # full observation (+ sensor noise)
    ("cpm_entire", 16, 64, dict(is_ego_view=False, is_partial_observation=False, is_observe_vertices=False, is_obs_steering=True, is_observe_ref_path_other_agents=True)),
    ("cpm_entire", 8, 48, dict(n_points_short_term=5, is_ego_view=False, is_observe_distance_to_boundaries=False)),  # another build variant (libsigmaenv_ns5.so)
]


@pytest.mark.parametrize("scen,N,B,kw", NSTEP_OBS_VARIANTS)
def test_nstep_launch_with_observation_variants(scen, N, B, kw):
    """Every obs_flags combination is assembled INSIDE the fused step (observe_tile_variant): the T-step launch records the configured row, == T single
    launches bit for bit (every buffer, every record row), == the oracle (pinned on the reference's trajectories) within 1e-5, masks / indices exact."""
    import torch
    from sigmarl_amd.shard import slab_width, unpack_slab

    T = 10
    base = dict(n_agents=N, scenario_type=scen, is_use_mtv_distance=False, rew_method="distance", dt=0.1, is_apply_mask=False, is_obs_noise=False, max_steps=9,
                random_seed=5)
    base.update(kw)
    p = Parameters(**base)
    mp = load_map(scen)
    cfg = make_config(p, mp, B)
    assert cfg.obs_flags != 0
    pf, pc = mp.list_first[0], mp.list_count[0]
    one, many, ora = _hip_env(cfg, mp), _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    D = one.env.D
    assert D == ora.D == capi.obs_dim(cfg.n_nearing, cfg.obs_flags, p.n_points_short_term, N)
    for d in (one, many):
        d.env.buffer(capi.BUF_DONE).fill_(1)
        d.auto_reset(5, 0, pf, pc)
    ora.get(capi.BUF_DONE, copy=False)[:] = 1
    ora.auto_reset(5, 0, pf, pc)
    _compare_all(one, ora, "initial observation")
    W = slab_width(N, D)
    acts = _actions(np.random.default_rng(21), T, B, N, True)
    ta = torch.as_tensor(acts).cuda()
    rec_one = torch.full((T, B, W), float("nan"), device="cuda")
    rec_many = torch.full((T, B, W), float("nan"), device="cuda")
    n_done = n_req = 0
    for t in range(T):
        one.env.set_slab(rec_one[t])
        one.env.step_autoreset(ta[t], 5, 100 + t, pf, pc)
        ora.step(acts[t])
        obs_t, rew_t, dn_t = unpack_slab(rec_one[t], N, D)
        assert np.abs(obs_t.cpu().numpy() - ora.get(capi.BUF_OBS)).max() <= 1e-5, f"record row, step {t}"
        assert np.abs(rew_t.cpu().numpy() - ora.get(capi.BUF_REWARD)).max() <= 1e-5
        assert np.array_equal(dn_t.cpu().numpy(), ora.get(capi.BUF_DONE).astype(bool))
        n_done += int(ora.get(capi.BUF_DONE).sum())
        n_req += int(ora.get(capi.BUF_COL_FLAGS)[..., 3].sum())
        ora.auto_reset(5, 100 + t, pf, pc)
        _compare_all(one, ora, f"step {t}")
    one.env.set_slab(None)
    many.env.step_autoreset_n(ta, rec_many, 5, 100, pf, pc)
    many.env.sync()
    assert torch.equal(rec_one.view(torch.int32), rec_many.view(torch.int32)), "record rows differ"
    for w in INT_BUFS + FLT_BUFS:
        assert one.get(w).tobytes() == many.get(w).tobytes(), f"buffer {w} differs between {T} launches and the one {T}-step launch"
    assert n_done > 0
    if scen == "on_ramp_1":
        assert n_req > 0  # per-agent re-placements inside the loop: re-placed agents' boundary points use the 'fresh' index shift, the others' the stepped one
    one.env.observe()
    ora.observe()
    _compare_all(one, ora, "sigmaenv_observe")
    for e in (one, many, ora):
        e.close()


def test_nstep_rejects_what_it_cannot_do():
    import torch

    mp = load_map("cpm_entire")
    p = Parameters(n_agents=4, scenario_type="cpm_entire", rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True, is_apply_mask=False, is_obs_noise=False)
    dev = _hip_env(make_config(p, mp, 8), mp)
    acts = torch.zeros((2, 8, 4, 2), device="cuda")
    with pytest.raises(RuntimeError, match="cbf_attach"):  # a CBF-informed chunk needs the segment tables
        dev.env.step_autoreset_n(acts)
    dev.env.step_autoreset_n(acts[:1])  # one step is the plain fused launch
    with pytest.raises(ValueError):
        dev.env.step_autoreset_n(acts[0])
    dev.close()


@pytest.mark.parametrize("qp,apply_action", [(False, False), (True, False), (True, True)])
def test_nstep_chunk_with_cbf_launches_equals_stepwise_calls(qp, apply_action):
    """rew_method "cbf" (QP-free margins, the centralized QP, the QP with its safe action applied): a chunk of T steps through ``sigmaenv_step_autoreset_n`` runs
    the CBF launch on step t's actions before step t (helper_training.py:1616-1627) and is bit-identical to the per-step calls -- every buffer, every record row."""
    import torch
    from sigmarl_amd.shard import slab_width

    N, B, T = 6, 40, 5
    p = Parameters(n_agents=N, scenario_type="cpm_entire", rew_method="cbf", dt=0.05, is_solve_qp=qp, is_using_cbf_training=True, is_apply_cbf_action=apply_action,
                   is_apply_mask=False, is_obs_noise=False, max_steps=9)
    from sigmarl_amd.env import NumpyAdapter, SigmaEnv

    one, many = NumpyAdapter(SigmaEnv(p, n_envs=B, device="cuda:0")), NumpyAdapter(SigmaEnv(p, n_envs=B, device="cuda:0"))
    mp = one.env.map
    pf, pc = mp.list_first[0], mp.list_count[0]
    for d in (one, many):
        d.env.cbf_attach()
        d.env.buffer(capi.BUF_DONE).fill_(1)
        d.auto_reset(5, 0, pf, pc)
    W = slab_width(N, one.env.D)
    acts = torch.as_tensor(_actions(np.random.default_rng(3), T, B, N, False)).cuda()
    rec_one = torch.full((T, B, W), float("nan"), device="cuda")
    rec_many = torch.full((T, B, W), float("nan"), device="cuda")
    safe = torch.zeros((B, N, 2), device="cuda")
    for t in range(T):
        one.env.set_slab(rec_one[t])
        if qp:
            one.env.cbf_qp(acts[t], safe)
            one.env.step_autoreset(safe if apply_action else acts[t], 5, 100 + t, pf, pc)
        else:
            one.env.cbf_rewards(acts[t])
            one.env.step_autoreset(acts[t], 5, 100 + t, pf, pc)
    one.env.set_slab(None)
    many.env.step_autoreset_n(acts, rec_many, 5, 100, pf, pc)
    many.env.sync()
    assert torch.equal(rec_one.view(torch.int32), rec_many.view(torch.int32)), "record rows differ"
    for w in INT_BUFS + FLT_BUFS + [capi.BUF_REWARD_INFO]:
        assert one.get(w).tobytes() == many.get(w).tobytes(), f"buffer {w} differs between the per-step calls and the {T}-step chunk"
    assert np.abs(one.get(capi.BUF_REWARD_INFO)).sum() > 0
    one.close()
    many.close()


@pytest.mark.parametrize("N", [1, 2])
def test_one_and_two_agent_tiles_far_from_their_path(N):
    """Tiles of one or two agent slots (the balanced scan's work list is at its minimum size there) with agents far from their own path: every
    candidate chunk of a polyline then passes the box test, up to 64 per task.  The list holds at least one whole task, so the scan ends -- and
    gives the oracle's (full-scan) result."""
    B = 12
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, rew_method="distance", is_apply_mask=False, is_obs_noise=False,
                   n_nearing_agents_observed=N - 1)
    cfg = make_config(p, mp, B)
    dev, ora = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp)
    rng = np.random.default_rng(N)
    st = np.zeros((B, N, 8), np.float32)
    ids = np.zeros((B, N, 4), np.int32)
    for b in range(B):
        for i in range(N):
            gp = mp.list_first[0] + int(rng.integers(mp.list_count[0]))
            ids[b, i] = (gp, 0, gp - mp.list_first[0], 5)
            # anywhere in (and beyond) the world, unrelated to the path: the closest-segment hint (point 5) is stale by metres
            st[b, i, 0:2] = rng.uniform(-1.0, 5.5, 2)
            st[b, i, 2] = rng.uniform(-3, 3)
    ei, ai = np.repeat(np.arange(B, dtype=np.int32), N), np.tile(np.arange(N, dtype=np.int32), B)
    for e in (dev, ora):
        e.reset(ei, ai, ids.reshape(-1, 4), st.reshape(-1, 8), 1)
        e.observe()
    _compare_all(dev, ora, "far-from-path start")
    for t in range(3):
        act = np.stack([rng.uniform(0, 1, (B, N)), rng.uniform(-0.3, 0.3, (B, N))], axis=-1).astype(np.float32)
        dev.step(act)
        ora.step(act)
        _compare_all(dev, ora, f"far-from-path step {t}")
    dev.close()
    ora.close()
