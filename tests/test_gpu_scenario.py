"""The VMAS-surface mirror (sigmarl_amd.scenario) driven in the VMAS call order on an MI355X."""
import numpy as np
import pytest

import traj_replay as tr
from sigmarl_amd import capi
from sigmarl_amd.params import Parameters

pytestmark = pytest.mark.gpu

INFO_KEYS = [
    "pos", "pos_nom", "rot", "rot_nom", "vel", "vel_nom", "act_vel", "act_vel_nom", "act_steer", "act_steer_nom", "ref", "ref_nom",
    "distance_ref", "distance_ref_nom", "distance_left_b", "distance_left_b_nom", "distance_right_b", "distance_right_b_nom",
    "is_collision_with_agents", "is_collision_with_lanelets", "is_reach_goal", "ref_lanelet_ids", "path_id", "applied_action_vel",
    "applied_action_steer", "nominal_action_vel", "nominal_action_steer",
] + list(capi.REWARD_INFO_FIELDS)


def _vmas_step(sc, actions):
    """What vmas.Environment.step does with a scenario (>= 1.4 order)."""
    import torch

    world = sc.world
    for i, a in enumerate(world.agents):
        u = actions[:, i].clone()
        rng = torch.tensor([float(a.u_range[0]), float(a.u_range[1])], device=u.device)
        a.action.u = u.clamp(-rng, rng)
    sc.pre_step()
    world.step()
    sc.post_step()
    rew = [sc.reward(a).clone() for a in world.agents]
    obs = [sc.observation(a).clone() for a in world.agents]
    info = [{k: v.clone() for k, v in sc.info(a).items()} for a in world.agents]
    done = sc.done().clone()
    return obs, rew, done, info


def test_surface_shapes_and_info_keys():
    import torch
    from sigmarl_amd.scenario import make_scenario

    B, N = 32, 16
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, num_vmas_envs=B)
    sc = make_scenario(p)
    world = sc.env_make_world(B, "cuda:0", n_agents=N)
    assert world.batch_dim == B and len(world.agents) == N and world.parameters is p
    sc.env_reset_world_at(None)
    a0 = world.agents[0]
    assert a0.action.u is None
    obs0 = sc.observation(a0)
    assert obs0.shape == (B, 32) and obs0.dtype == torch.float32
    info0 = sc.info(a0)  # callable before the first step (road_traffic.py:1505)
    assert list(info0.keys()) == INFO_KEYS and len(INFO_KEYS) == 39
    assert a0.state.pos.shape == (B, 2) and a0.state.rot.shape == (B, 1) and a0.state.speed.shape == (B, 1)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(5):
        act = torch.rand((B, N, 2), generator=gen, device="cuda") * torch.tensor([1.3, 1.4], device="cuda") - torch.tensor([0.1, 0.7], device="cuda")
        obs, rew, done, info = _vmas_step(sc, act)
        assert done.dtype == torch.bool and done.shape == (B,)
        assert rew[3].shape == (B,) and obs[3].shape == (B, 32)
        assert info[3]["ref_lanelet_ids"].shape == (B, 104) and info[3]["ref"].shape == (B, 6)
        # WorldCustom.step clamps the action in place (helper_training.py:807-818)
        assert float(a0.action.u[:, 0].abs().max()) <= 1.0 + 1e-6
        assert torch.equal(info[3]["act_vel"], world.agents[3].action.u[:, 0])
        for e in torch.nonzero(done).flatten().tolist():  # what TorchRL's step_and_maybe_reset does
            sc.env_reset_world_at(e)
    st = sc.env.state
    assert torch.isfinite(st).all()
    # the views really alias the device buffers
    assert world.agents[5].state.pos.data_ptr() == st[:, 5, 0:2].data_ptr()
    sc.env.close()


def test_scenario_matches_plain_env_on_golden_prefix():
    """Golden initial state injected through the scenario's env; stepping through the VMAS surface reproduces the reference."""
    import torch
    from sigmarl_amd.scenario import make_scenario

    z, meta = tr.load_fixture("cpm16_c2c_noreset")
    p = tr.params_from_meta(meta)
    sc = make_scenario(p)
    sc.env_make_world(meta["B"], "cuda:0", n_agents=meta["n_agents"])
    from sigmarl_amd.env import NumpyAdapter

    tr.apply_initial_reset(NumpyAdapter(sc.env), z, sc.map)
    for t in range(12):
        act = torch.as_tensor(z["act"][t]).cuda()
        obs, rew, done, info = _vmas_step(sc, act)
        assert np.abs(torch.stack(rew, 1).cpu().numpy() - z["post_reward"][t]).max() <= 1e-5
        assert np.abs(torch.stack(obs, 1).cpu().numpy() - z["post_obs"][t]).max() <= 1e-5
        assert np.array_equal(done.cpu().numpy(), z["done"][t])
        rot = torch.stack([i["rot"] for i in info], 1).squeeze(-1).cpu().numpy()
        assert np.abs(rot - z["post_info_rot"][t]).max() <= 1e-5
        dl = torch.stack([i["distance_left_b"] for i in info], 1).cpu().numpy()
        assert np.abs(dl - z["post_info_distance_left_b"][t]).max() <= 1e-5
        tot = torch.stack([i["rew_total"] for i in info], 1).cpu().numpy()
        assert np.abs(tot - z["post_info_rew_total"][t]).max() <= 1e-5  # RewardInfo.reset quirk: only the last agent's entry survives
    sc.env.close()


def test_cbf_margin_reward_through_the_surface():
    """rew_method "cbf_sparse" with the QP-free margin reward: CBFQP.update_qp (one launch for the batch) before every VMAS step
    reproduces the reference's rewards (golden traj_intersection4_cbf: is_solve_qp=False, mtv distances)."""
    import torch
    from sigmarl_amd.cbf import CBFQP, cbf_constrained_centralized_policy
    from sigmarl_amd.env import NumpyAdapter
    from sigmarl_amd.scenario import make_scenario

    z, meta = tr.load_fixture("intersection4_cbf")
    p = tr.params_from_meta(meta)
    assert p.rew_method == "cbf_sparse" and not p.is_solve_qp
    sc = make_scenario(p)
    sc.env_make_world(meta["B"], "cuda:0", n_agents=meta["n_agents"])
    ad = NumpyAdapter(sc.env)
    tr.apply_initial_reset(ad, z, sc.map)
    controllers = [CBFQP(env=sc, env_idx=e) for e in range(meta["B"])]  # the reference's per-env list (mappo_cavs.py:583)
    n_cbf = 0
    for t in range(meta["T"]):
        act = torch.as_tensor(z["act"][t]).cuda()
        td = {}
        cbf_constrained_centralized_policy(td, lambda d: d.__setitem__(("agents", "action"), act), controllers)
        ch = torch.stack([sc.reward_info.rew_near_left_lane, sc.reward_info.rew_near_right_lane, sc.reward_info.rew_near_other_agents]).cpu().numpy()
        assert np.abs(ch - z["cbf_rew"][t]).max() <= 1e-5
        n_cbf += int((ch != 0).sum())
        obs, rew, done, info = _vmas_step(sc, act)
        assert np.abs(torch.stack(rew, 1).cpu().numpy() - z["post_reward"][t]).max() <= 5e-5  # mtv tolerance (see test_gpu_parity)
        assert np.array_equal(done.cpu().numpy(), z["done"][t])
        if tr.apply_events(ad, z, sc.map, t):
            ad.observe()
    assert n_cbf > 100
    sc.env.close()


@pytest.mark.parametrize("apply", [False, True])
def test_cbf_qp_through_the_surface(apply):
    """is_solve_qp=True: CBFQP.update_qp solves the centralized QP of every env, leaves world_state.nominal_action_* behind (and overwrites the
    action tensor when is_apply_cbf_action); the step's reward carries the deviation penalty (road_traffic.py:1112-1139)."""
    import torch
    from sigmarl_amd.cbf import CBFQP
    from sigmarl_amd.scenario import make_scenario

    B, N = 12, 6
    p = Parameters(n_agents=N, scenario_type="cpm_entire", rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_apply_cbf_action=apply,
                   is_apply_mask=False, is_obs_noise=False, is_use_mtv_distance=False, num_vmas_envs=B)
    sc = make_scenario(p)
    sc.env_make_world(B, "cuda:0", n_agents=N)
    sc.env_reset_world_at(None)
    ctl = CBFQP(env=sc)
    gen = torch.Generator(device="cuda").manual_seed(1)
    seen = 0
    for t in range(4):
        act = (torch.rand((B, N, 2), generator=gen, device="cuda") * torch.tensor([1.6, 1.2], device="cuda") - torch.tensor([0.5, 0.6], device="cuda")).contiguous()
        rl = act.clone()
        td = {("agents", "action"): act}
        ctl.update_qp(td)
        sc.env.sync()
        nom = torch.stack([sc.world_state.nominal_action_vel, sc.world_state.nominal_action_steer], -1).clone()
        rl_c = torch.stack([rl[..., 0].clamp(-0.5, 1.0), rl[..., 1].clamp(-float(sc.max_steering), float(sc.max_steering))], -1)
        if apply:
            assert torch.allclose(nom, rl_c, atol=1e-7)   # the nominal slot keeps the (clamped) policy action ...
            safe = act                                      # ... and the action tensor was overwritten with the safe action
        else:
            assert torch.equal(act, rl)
            safe = nom
        seen += int(((safe - rl_c).abs().amax(dim=(1, 2)) > 1e-5).sum())
        obs, rew, done, info = _vmas_step(sc, act)
        # the deviation penalty: -0.05 * |applied - nominal| / max, per channel, inside the clamped reward
        applied = torch.stack([a.action.u for a in sc.world.agents], 1)
        pen = -0.05 * ((applied[..., 0] - nom[..., 0]).abs() / 1.0) + -0.05 * ((applied[..., 1] - nom[..., 1]).abs() / float(sc.max_steering))
        r = torch.stack(rew, 1)
        base = torch.stack([sc.reward_info.rew_total for _ in range(1)], 0)  # (only the last agent's entry survives; not used further)
        assert torch.isfinite(r).all() and base.shape[-1] == N
        assert float((r - pen).abs().max()) < 1.2  # progress / goal terms on top of the penalty; sanity only
        for e in torch.nonzero(done).flatten().tolist():
            sc.env_reset_world_at(e)
    assert seen > 0
    sc.env.close()


@pytest.mark.parametrize("scen,N,testing", [("intersection_1", 4, False), ("on_ramp_1", 4, False), ("cpm_entire", 4, True)])
def test_host_driven_agent_resets(scen, N, testing):
    """Non-loop maps / testing mode: done() performs the per-agent resets (torch RNG) the reference performs there.
    Agents are injected just before the end of their path (non-loop maps) or straight at a lane boundary (testing mode) so that
    the reset requests fire within a few steps."""
    import torch
    from sigmarl_amd.scenario import make_scenario

    torch.manual_seed(0)
    B = 4
    p = Parameters(n_agents=N, scenario_type=scen, is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, is_testing_mode=testing,
                   dt=0.1, max_steps=1000)
    sc = make_scenario(p)
    world = sc.env_make_world(B, "cuda:0", n_agents=N)
    sc.env_reset_world_at(None)
    mp = sc.map
    ids, st = [], []
    for b in range(B):
        for i in range(N):
            gp = mp.list_first[0] + (i % mp.list_count[0])
            n = int(mp.n_center[gp])
            k = (n - 4) if not testing else 10 + 12 * i
            x, y = mp.center[gp, k]
            yaw = float(mp.yaw[gp, min(k, int(mp.n_yaw[gp]) - 1)]) + (0.9 if testing else 0.0)
            ids.append((gp, 0, gp - mp.list_first[0], k))
            st.append((x, y, yaw, 0.8, 0.0, 0.8 * np.cos(yaw), 0.8 * np.sin(yaw), 0.0))
    sc.env.reset(np.repeat(np.arange(B), N), np.tile(np.arange(N), B), np.asarray(ids, np.int32), np.asarray(st, np.float32), True)
    sc._obs_dirty = True
    calls = []
    orig = sc.reset_world_at

    def counting(env_index=None, agent_index=None):
        if agent_index is not None:
            calls.append((int(env_index), int(agent_index)))
        return orig(env_index=env_index, agent_index=agent_index)

    sc.reset_world_at = counting
    obs = [sc.observation(a) for a in world.agents]
    act = torch.zeros((B, N, 2), device="cuda")
    act[..., 0] = 1.0
    for t in range(12):
        obs, rew, done, info = _vmas_step(sc, act)
        for e in torch.nonzero(done).flatten().tolist():
            sc.env_reset_world_at(e)
        obs = [sc.observation(a) for a in world.agents]
        assert torch.isfinite(torch.stack(obs, 1)).all()
    assert len(calls) > 0, "no per-agent reset was exercised"
    assert torch.isfinite(sc.env.state).all()
    # every reset (and every step) leaves prev_pos == pos (state_buffer semantics, road_traffic.py:902-923,1226-1240)
    assert torch.equal(sc.env.buffer(capi.BUF_PREV_POS), sc.env.state[..., 0:2])
    sc.env.close()


def test_device_side_resets_through_the_surface():
    """device_side_resets=True: done() serves finished envs and per-agent requests on the GPU; TorchRL's later reset_world_at(e)
    for the finished envs is then a no-op for that step."""
    import torch
    from sigmarl_amd.scenario import make_scenario

    B, N = 64, 16
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False)
    sc = make_scenario(p)
    sc.device_side_resets = True
    world = sc.env_make_world(B, "cuda:0", n_agents=N)
    sc.env_reset_world_at(None)
    gen = torch.Generator(device="cuda").manual_seed(3)
    total_done = 0
    for _ in range(8):
        act = torch.rand((B, N, 2), generator=gen, device="cuda") - torch.tensor([0.0, 0.5], device="cuda")
        obs, rew, done, info = _vmas_step(sc, act)
        resets_before = int(sc.env.buffer(capi.BUF_TIMER)[:, 3].sum())
        for e in torch.nonzero(done).flatten().tolist():
            sc.env_reset_world_at(e)  # no second reset
        assert int(sc.env.buffer(capi.BUF_TIMER)[:, 3].sum()) == resets_before
        total_done += int(done.sum())
        assert not sc.env.done.any() and (sc.timer.step[done] == 0).all()
        new_obs = torch.stack([sc.observation(a) for a in world.agents], 1)
        assert torch.isfinite(new_obs).all()
    assert total_done > 0
    sc.env.close()


def test_prioritized_marl_info_and_opponent_placeholders_through_the_surface():
    """is_using_prioritized_marl: info() carries base_observation (the observation padded with the 2 K placeholder columns) and priority_observation
    (road_traffic.py:1513-1520, :1616-1625).  is_using_opponent_modeling: observation() itself ends with the placeholders (observation_provider_rt.py:606-611)
    and env.opponent_fill is the gather opponent_modeling performs between its two policy calls (helper_training.py:1117-1137).  is_using_pseudo_distance
    is accepted and changes nothing (it is read nowhere in the reference)."""
    import torch
    from sigmarl_amd.scenario import make_scenario

    B, N, K = 16, 8, 2
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, num_vmas_envs=B,
                   is_using_prioritized_marl=True, is_using_pseudo_distance=True)
    sc = make_scenario(p)
    world = sc.env_make_world(B, "cuda:0", n_agents=N)
    sc.env_reset_world_at(None)
    gen = torch.Generator(device="cuda").manual_seed(1)
    act = torch.rand((B, N, 2), generator=gen, device="cuda") - torch.tensor([0.0, 0.5], device="cuda")
    obs, rew, done, info = _vmas_step(sc, act)
    assert list(info[2].keys())[:29] == INFO_KEYS[:27] + ["base_observation", "priority_observation"] and len(info[2]) == 41
    assert info[2]["base_observation"].shape == (B, 32 + 2 * K) and torch.equal(info[2]["base_observation"][:, :32], obs[2])
    assert (info[2]["base_observation"][:, 32:] == 0).all() and torch.equal(info[2]["priority_observation"], obs[2])
    sc.env.close()

    p2 = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, num_vmas_envs=B,
                    is_using_opponent_modeling=True)
    sc2 = make_scenario(p2)
    world2 = sc2.env_make_world(B, "cuda:0", n_agents=N)
    sc2.env_reset_world_at(None)
    obs, rew, done, info = _vmas_step(sc2, act)
    o = torch.stack(obs, 1)
    assert o.shape == (B, N, 32 + 2 * K) and (o[..., 32:] == 0).all()
    tentative = torch.randn((B, N, 2), device="cuda")
    sc2.env.opponent_fill(tentative)
    filled = torch.stack([sc2.observation(a) for a in world2.agents], 1)
    near = sc2.env.buffer(capi.BUF_NEARING).long()
    want = torch.gather(tentative[:, None].expand(B, N, N, 2), 2, near[..., None].expand(B, N, K, 2)).reshape(B, N, 2 * K)
    assert torch.equal(filled[..., 32:], want) and torch.equal(filled[..., :32], o[..., :32])
    sc2.env.close()


def test_mixed_map_device_side_resets_through_the_surface():
    """cpm_mixed with device_side_resets=True (raised NotImplementedError until round 3): the initial reset and done() draw every env's sub-scenario
    from cpm_scenario_probabilities on the GPU; the agents of an env share it and their paths belong to its list."""
    import torch
    from sigmarl_amd.scenario import make_scenario

    B, N = 128, 2
    p = Parameters(n_agents=N, scenario_type="cpm_mixed", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, dt=0.1, max_steps=6,
                   cpm_scenario_probabilities=[0.4, 0.3, 0.3])
    sc = make_scenario(p)
    sc.device_side_resets = True
    world = sc.env_make_world(B, "cuda:0", n_agents=N)
    sc.env_reset_world_at(None)
    mp = sc.map
    seen = set()
    gen = torch.Generator(device="cuda").manual_seed(5)
    for _ in range(10):
        act = torch.rand((B, N, 2), generator=gen, device="cuda") - torch.tensor([0.0, 0.5], device="cuda")
        obs, rew, done, info = _vmas_step(sc, act)
        pa = sc.env.buffer(capi.BUF_PATH).cpu().numpy()
        sid = pa[..., 1]
        assert (sid == sid[:, :1]).all() and sid.min() >= 1 and sid.max() <= 3
        first = np.asarray([mp.list_first[k] for k in range(4)])[sid]
        count = np.asarray([mp.list_count[k] for k in range(4)])[sid]
        assert ((pa[..., 0] >= first) & (pa[..., 0] < first + count)).all()
        seen |= set(np.unique(sid).tolist())
        assert not sc.env.done.any()
    assert seen == {1, 2, 3}
    sc.env.close()


@pytest.mark.parametrize("name,kw", [("cpm16", dict(n_agents=16, scenario_type="cpm_entire")), ("intersection4", dict(n_agents=4, scenario_type="intersection_1")),
                                     ("cpm8_testing", dict(n_agents=8, scenario_type="cpm_entire", is_testing_mode=True))])
def test_seeded_initial_reset_reproduces_the_reference(name, kw):
    """`torch.manual_seed(s)` right before `env_reset_world_at(None)` gives the reference's initial states through the mirrored surface
    (road_traffic.py:832-834 loops the envs; world_state_rt_sim.py:215-311 draws path / point / speed from torch's global generator in this
    order): fixture tests/golden/initial_reset.npz was recorded from the reference exactly that way.  Only `device_side_resets=True`
    switches the initial reset to the device sampler."""
    import os

    import torch
    from sigmarl_amd.scenario import make_scenario

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "initial_reset.npz"))
    B, N, seed = [int(v) for v in z[name + "_meta"]]
    p = Parameters(is_apply_mask=False, is_obs_noise=False, num_vmas_envs=B, max_steps=128, **kw)
    sc = make_scenario(p)
    sc.env_make_world(B, "cuda:0", n_agents=N)
    torch.manual_seed(seed)
    sc.env_reset_world_at(None)
    env = sc.env
    env.sync()
    path = env.buffer(capi.BUF_PATH).cpu().numpy()
    assert np.array_equal(path[:, :, 2], z[name + "_path_id"])
    assert np.array_equal(path[:, :, 3], z[name + "_point_id"])
    st = env.state.cpu().numpy()
    assert np.abs(st[:, :, 0:2] - z[name + "_pos"]).max() <= 1e-6
    assert np.abs(st[:, :, 2] - z[name + "_rot"]).max() <= 1e-6
    assert np.abs(st[:, :, 3] - z[name + "_speed"]).max() <= 1e-6
    env.close()


@pytest.mark.parametrize("vmas_bases", [False, True])
def test_mirror_under_an_environment_shaped_driver(vmas_bases):
    """(vmas_bases: the same run with the mirror's classes DERIVED from vmas' World / Agent / AgentState / BaseScenario -- tests/fake_vmas.py, the published class
    skeleton whose constructors and tensor-touching methods raise when reached -- as they are wherever vmas is installed: isinstance holds and nothing changes.)
    The plugin surface driven the way vmas' Environment drives a scenario (tests/vmas_env_shim.py: __init__ -> env_make_world + reset, step ->
    _set_action / env_process_action / pre_step / world.step / post_step / get_from_scenario, reset_at for finished envs as TorchRL's VmasEnv
    issues it): every attribute that driver touches exists on WorldCustom / Vehicle / Action, out-of-range actions trip its assertion unless
    clamp_actions is set, and the results equal the same episode driven through the callbacks by hand."""
    import contextlib
    import importlib
    import sys

    import torch

    import fake_vmas
    import sigmarl_amd.scenario as scmod
    from vmas_env_shim import EnvironmentShim

    with (fake_vmas.installed() if vmas_bases else contextlib.nullcontext()):
        try:
            scmod = importlib.reload(scmod)
            _environment_shaped_driver(scmod, EnvironmentShim, torch)
            if vmas_bases:
                core = sys.modules["vmas.simulator.core"]
                assert issubclass(scmod.WorldCustom, core.World) and issubclass(scmod.Vehicle, core.Agent) and issubclass(scmod.VehicleState, core.AgentState)
        finally:
            if vmas_bases:
                for n in [k for k in sys.modules if k == "vmas" or k.startswith("vmas.")]:
                    sys.modules.pop(n)
            importlib.reload(scmod)


def _environment_shaped_driver(scmod, EnvironmentShim, torch):
    make_scenario = scmod.make_scenario
    B, N, T = 24, 4, 14  # (4 agents: the reference's rejection sampler, restated on the host, does not terminate for more on this small map)
    kw = dict(n_agents=N, scenario_type="intersection_1", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, num_vmas_envs=B, max_steps=9, dt=0.1)
    sc_a, sc_b = make_scenario(Parameters(**kw)), make_scenario(Parameters(**kw))
    env = EnvironmentShim(sc_a, num_envs=B, device="cuda:0", continuous_actions=True, max_steps=None, seed=5, clamp_actions=True, n_agents=N)
    torch.manual_seed(5)
    world_b = sc_b.env_make_world(B, "cuda:0", n_agents=N)
    sc_b.env_reset_world_at(None)
    assert env.n_agents == N and env.world.dim_c == 0 and all(a.action_size == 2 and a.silent for a in env.agents)
    assert torch.allclose(env.agents[0].action.u_range_tensor.cpu(), torch.tensor([1.0, 31 * np.pi / 180], dtype=torch.float32))
    obs0 = env.reset(seed=5)
    torch.manual_seed(5)
    sc_b.env_reset_world_at(None)
    assert len(obs0) == N and obs0[0].shape == (B, 32)
    gen = torch.Generator(device="cuda").manual_seed(1)
    seen_done = 0
    for t in range(T):
        act = torch.rand((B, N, 2), generator=gen, device="cuda") * torch.tensor([1.3, 1.4], device="cuda") - torch.tensor([0.1, 0.7], device="cuda")
        st0 = torch.get_rng_state()  # done() re-places agents that left through an exit with draws from torch's generator: same draws on both sides
        obs, rew, done, info = env.step([act[:, i] for i in range(N)])
        torch.set_rng_state(st0)
        obs_b, rew_b, done_b, info_b = _vmas_step(sc_b, act)  # by hand: the same clamp, then the callback order
        assert all(torch.equal(a, b) for a, b in zip(obs, obs_b)) and all(torch.equal(a, b) for a, b in zip(rew, rew_b)) and torch.equal(done, done_b)
        assert list(info[0].keys()) == INFO_KEYS and all(torch.equal(info[2][k], info_b[2][k]) for k in INFO_KEYS)
        idx = torch.nonzero(done).flatten().tolist()
        seen_done += len(idx)
        st = torch.get_rng_state()
        for e in idx:  # TorchRL's VmasEnv._reset: reset_at for every finished env
            env.reset_at(e)
        torch.set_rng_state(st)
        for e in idx:
            sc_b.env_reset_world_at(e)
        assert torch.equal(sc_a.env.state, sc_b.env.state)
    assert seen_done > 0
    # without clamp_actions an out-of-range action trips the driver's own assertion (as in vmas)
    env2 = EnvironmentShim(make_scenario(Parameters(**kw)), num_envs=4, device="cuda:0", n_agents=N)
    with pytest.raises(AssertionError, match="out of its range"):
        env2.step([torch.full((4, 2), 2.0, device="cuda") for _ in range(N)])
    env2.to("cuda:0")
    with pytest.raises(RuntimeError):
        env2.world.to("cpu")
    for s_ in (sc_a, sc_b, env2.scenario):
        s_.env.close()
