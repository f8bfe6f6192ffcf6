"""Sensor noise of an observation that is TAKEN AGAIN (observation_provider_rt.py:613-618: ``obs + noise_level * rand_like(obs)`` on every ``observation()`` call).
The product's draws are a function of the env's own counters -- plus, for a stand-alone re-observation (``sigmaenv_observe``), the number of that call on its handle
(``obs_noise``, shared specification with the oracle): two observations of the same state carry INDEPENDENT uniform noise, as the reference's two calls do."""
import numpy as np
import pytest

import oracle_binding as ob
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config

LEVEL = 0.07


def _cfg(B=512, N=8, **kw):
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, rew_method="distance", dt=0.05, is_apply_mask=False, is_obs_noise=True,
                   obs_noise_level=LEVEL, max_steps=50, random_seed=11, **kw)
    mp = load_map("cpm_entire")
    return p, mp, make_config(p, mp, B)


def _start(env, mp):
    env.get(capi.BUF_DONE, copy=False)[:] = 1 if isinstance(env, ob.OracleEnv) else 0
    if not isinstance(env, ob.OracleEnv):
        env.env.buffer(capi.BUF_DONE).fill_(1)
    env.auto_reset(4, 0, mp.list_first[0], mp.list_count[0])


def _check_retaken(env, clean):
    """three re-observations of one state: each is the noise-free row + U[0, level), the three noise fields are pairwise uncorrelated and differ from the reset's own"""
    base = clean.get(capi.BUF_OBS).astype(np.float64)
    n0 = env.get(capi.BUF_OBS).astype(np.float64) - base  # the observation the reset produced
    fields = [n0]
    for _ in range(3):
        env.observe()
        fields.append(env.get(capi.BUF_OBS).astype(np.float64) - base)
    for n in fields:
        assert n.min() >= -1e-6 and n.max() < LEVEL + 1e-6
        assert abs(n.mean() - LEVEL / 2) < 2e-4 and abs(n.var() - LEVEL ** 2 / 12) < 2e-5
    for a in range(len(fields)):
        for b in range(a + 1, len(fields)):
            d = fields[a] - fields[b]
            assert np.abs(d).max() > LEVEL / 4, "a re-taken observation repeats the noise"
            assert abs(d.mean()) < 3e-4 and abs(d.var() - 2 * LEVEL ** 2 / 12) < 4e-5       # difference of two independent uniforms
            r = np.corrcoef(fields[a].ravel(), fields[b].ravel())[0, 1]
            assert abs(r) < 0.01, f"noise fields {a} and {b} are correlated ({r:.3f})"
    return fields


def test_oracle_retaken_observation_draws_new_noise():
    p, mp, cfg = _cfg()
    pc, mpc, cfgc = _cfg(B=512)
    cfgc.obs_noise_level = 0.0
    noisy, clean = ob.OracleEnv(cfg, mp), ob.OracleEnv(cfgc, mp)
    for e in (noisy, clean):
        _start(e, mp)
    _check_retaken(noisy, clean)
    # a step's observation is keyed on the counters alone (salt 0): the same whatever was re-observed before -- T-step launches, shards and the record agree on it
    other = ob.OracleEnv(cfg, mp)
    _start(other, mp)
    act = np.tile(np.array([0.5, 0.0], np.float32), (noisy.B, noisy.N, 1))
    noisy.step(act)
    other.step(act)
    assert np.array_equal(noisy.get(capi.BUF_OBS), other.get(capi.BUF_OBS))
    for e in (noisy, clean, other):
        e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [{}, {"is_ego_view": False}, {"is_ego_view": False, "is_partial_observation": False}])
def test_hip_retaken_observation_equals_the_oracle_and_draws_new_noise(extra):
    """the HIP path through the C-ABI: the same call sequence as the oracle gives the same rows (1e-5), default row / variant rows / full observation"""
    from test_gpu_parity import _hip_env

    p, mp, cfg = _cfg(**extra)
    pc, mpc, cfgc = _cfg(**extra)
    cfgc.obs_noise_level = 0.0
    dev, ora, clean = _hip_env(cfg, mp), ob.OracleEnv(cfg, mp), _hip_env(cfgc, mp)
    for e in (dev, ora, clean):
        _start(e, mp)
    assert np.abs(dev.get(capi.BUF_OBS) - ora.get(capi.BUF_OBS)).max() <= 1e-5
    fields = _check_retaken(dev, clean)
    for _ in range(3):
        ora.observe()
    assert np.abs(dev.get(capi.BUF_OBS) - ora.get(capi.BUF_OBS)).max() <= 1e-5  # (the third re-observation on both sides)
    assert len(fields) == 4
    for e in (dev, ora, clean):
        e.close()


@pytest.mark.gpu
def test_surface_second_observation_call_returns_fresh_noise():
    """ScenarioRoadTraffic.observation(agent) called again without a step in between (the reference draws rand_like per call)"""
    import torch
    from sigmarl_amd.scenario import make_scenario

    p = Parameters(n_agents=4, scenario_type="cpm_entire", is_obs_noise=True, obs_noise_level=LEVEL, is_apply_mask=False, random_seed=3)
    sc = make_scenario(p)
    sc.device_side_resets = True
    world = sc.env_make_world(256, "cuda:0", n_agents=4)
    sc.env_reset_world_at(None)
    first = [sc.observation(a).clone() for a in world.agents]
    again = [sc.observation(a).clone() for a in world.agents]
    for a, b in zip(first, again):
        d = (a - b).double()
        assert float(d.abs().max()) > LEVEL / 4 and float(d.abs().max()) < LEVEL + 1e-6   # same state, new draws
    for a in world.agents:
        a.action.u = torch.zeros((256, 2), device="cuda:0")
    world.step()
    stepped = [sc.observation(a).clone() for a in world.agents]  # the step's own observation: one call each, no re-observation
    assert all(float((x - y).abs().max()) > 0 for x, y in zip(stepped, again))
    sc.env.close()
