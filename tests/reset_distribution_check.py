"""The device-side reset sampler against the REFERENCE's sampler, statistically (test infrastructure shared by the CPU and GPU tests).

tests/golden/reset_distribution.npz holds histograms of (path, point, speed) drawn by the reference's rejection sampler
(world_state_rt_sim.py:215-311; generator tests/golden/gen/gen_reset_distribution.py).  ``sample_histograms`` draws the same quantities from an
env twin (oracle or HIP: same counter-based specification) and ``compare`` runs two-sample chi-square tests on every marginal."""
import os

import numpy as np
from scipy.stats import chi2

from sigmarl_amd import capi

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reset_distribution.npz")


def sample_histograms(env, mp, rounds, seed=77):
    """Full-env resets of every env of ``env`` (an OracleEnv / NumpyAdapter on the CPM map), ``rounds`` times."""
    B, N = env.B, env.N
    pf, pc = mp.list_first[0], mp.list_count[0]
    path_c = np.zeros(pc, np.int64)
    frac_c, speed_c, first_pt, last_pt = (np.zeros(20, np.int64) for _ in range(4))
    point_raw = np.zeros(64, np.int64)
    min_sp = np.inf
    for r in range(rounds):
        if hasattr(env, "env"):
            env.env.buffer(capi.BUF_DONE).fill_(1)
        else:
            env.get(capi.BUF_DONE, copy=False)[:] = 1
        env.auto_reset(seed, r, pf, pc)
        st, pa = env.get(capi.BUF_STATE), env.get(capi.BUF_PATH)
        pid, pt = pa[..., 2].astype(np.int64), pa[..., 3].astype(np.int64)
        half = mp.n_center[pa[..., 0]] // 2
        frac = (pt - 3) / np.maximum(1, half - 3)
        np.add.at(path_c, pid.ravel(), 1)
        np.add.at(frac_c, np.minimum(19, (frac.ravel() * 20).astype(np.int64)), 1)
        np.add.at(speed_c, np.minimum(19, (st[..., 3].ravel() * 20).astype(np.int64)), 1)
        np.add.at(first_pt, np.minimum(19, (frac[:, 0] * 20).astype(np.int64)), 1)
        np.add.at(last_pt, np.minimum(19, (frac[:, -1] * 20).astype(np.int64)), 1)
        np.add.at(point_raw, np.minimum(63, pt.ravel()), 1)
        pos = st[..., 0:2].astype(np.float64)
        d = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1)) + np.eye(N)[None] * 1e9
        min_sp = min(min_sp, float(d.min()))
    return dict(path_counts=path_c, point_frac_counts=frac_c, speed_counts=speed_c, first_agent_point=first_pt, last_agent_point=last_pt,
                point_raw_counts=point_raw, min_spacing=min_sp)


def chi2_two_sample(a, b):
    """p-value of the two-sample chi-square test on two histograms of the same bins (bins empty in both are dropped)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    m = (a + b) > 0
    a, b = a[m], b[m]
    k1, k2 = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    stat = float((((k1 * a - k2 * b) ** 2) / (a + b)).sum())
    return float(chi2.sf(stat, max(1, m.sum() - 1))), stat


def compare(tag, got, p_min=1e-4):
    z = np.load(FIXTURE)
    res = {}
    for key in ("path_counts", "point_frac_counts", "speed_counts", "first_agent_point", "last_agent_point", "point_raw_counts"):
        p, stat = chi2_two_sample(z[f"{tag}_{key}"], got[key])
        res[key] = p
        assert p >= p_min, (tag, key, p, stat, z[f"{tag}_{key}"], got[key])
    # both samplers reject starts closer than reset_agent_min_distance (the bounded device sampler falls back after 64 tries: never on this map)
    assert got["min_spacing"] >= float(z[f"{tag}_min_distance"]) - 1e-6, got["min_spacing"]
    return res


# ---- cpm_mixed: sub-scenario lists (world_state_rt_sim.py:313-358) -----------------------------------------------------------------------
def sample_mixed(env, mp, rounds, probabilities, seed=91):
    """Full-env resets of every env of a cpm_mixed twin through the scenario lists: histograms of the drawn sub-scenario, of the
    (sub-scenario, list-local path) pairs, of the point position and of the speed."""
    env.set_scenario_lists(list(probabilities))
    counts = [mp.list_count[k] for k in (1, 2, 3)]
    offs = np.concatenate([[0], np.cumsum(counts)])
    scen_c, pair_c = np.zeros(3, np.int64), np.zeros(int(offs[-1]), np.int64)
    frac_c, speed_c = np.zeros(20, np.int64), np.zeros(20, np.int64)
    min_distance = float(np.load(FIXTURE)["mixed_min_distance"])
    n_fallback = 0
    for r in range(rounds):
        if hasattr(env, "env"):
            env.env.buffer(capi.BUF_DONE).fill_(1)
        else:
            env.get(capi.BUF_DONE, copy=False)[:] = 1
        env.auto_reset(seed, r, 0, capi.SCENARIO_LISTS)
        st, pa = env.get(capi.BUF_STATE), env.get(capi.BUF_PATH)
        gp, sid, pid, pt = (pa[..., k].astype(np.int64) for k in range(4))
        assert (sid == sid[:, :1]).all() and sid.min() >= 1 and sid.max() <= 3           # one sub-scenario per env (:343)
        first = np.asarray([mp.list_first[k] for k in range(4)])[sid]
        assert np.array_equal(gp, first + pid) and (pid >= 0).all() and (pid < np.asarray([0] + counts)[sid]).all()  # the path belongs to that list
        half = mp.n_center[gp] // 2
        frac = (pt - 3) / np.maximum(1, half - 3)
        # envs the bounded sampler could not place (64 tries, then the last try stands): the reference's unbounded loop never returns on those -- its
        # generator abandons them (gen_reset_distribution.py) -- so both sides count the feasible resets only
        pos = st[..., 0:2].astype(np.float64)
        d = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1)) + np.eye(env.N)[None] * 1e9
        ok = d.min(axis=(1, 2)) >= min_distance - 1e-6
        n_fallback += int((~ok).sum())
        np.add.at(scen_c, sid[ok, 0] - 1, 1)
        np.add.at(pair_c, (offs[sid[ok] - 1] + pid[ok]).ravel(), 1)
        np.add.at(frac_c, np.minimum(19, (frac[ok].ravel() * 20).astype(np.int64)), 1)
        np.add.at(speed_c, np.minimum(19, (st[ok][..., 3].ravel() * 20).astype(np.int64)), 1)
    return dict(scenario_counts=scen_c, pair_counts=pair_c, point_frac_counts=frac_c, speed_counts=speed_c, n_fallback=n_fallback, n_total=rounds * env.B)


def compare_mixed(got, p_min=1e-4):
    z = np.load(FIXTURE)
    assert "mixed_scenario_counts" in z.files, "tests/golden/reset_distribution.npz has no cpm_mixed histograms (gen_reset_distribution.py --mixed-only)"
    res = {}
    for key in ("scenario_counts", "pair_counts", "point_frac_counts", "speed_counts"):
        p, stat = chi2_two_sample(z[f"mixed_{key}"], got[key])
        res[key] = p
        assert p >= p_min, (key, p, stat, z[f"mixed_{key}"], got[key])
    # the share of infeasible placements agrees as well (reference: resets its unbounded loop never finished; here: fallbacks of the bounded one)
    ref_stuck = float(z["mixed_n_stuck"]) / float(z["mixed_n_stuck"] + z["mixed_n_env_resets"])
    assert abs(got["n_fallback"] / got["n_total"] - ref_stuck) < 0.01, (got["n_fallback"], got["n_total"], ref_stuck)
    return res
