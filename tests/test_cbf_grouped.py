"""CPU tests of the GROUPED CBF-QPs of the oracle (oracle/sigmaenv_cbf_oracle.inc; sigmarl/cbf_qp.py:193-310, :1562-2281, :2491-2532).

Pinned on tests/golden/cbf_grouped.npz, which the reference's own CBFQP(is_grouping_agents=True).update_qp produced under a storing
stand-in for cvxpy (tests/golden/gen/gen_cbf_grouped.py): groups, neighbour lists, every intra-group pair row and cross-group row, U_nom
and the nominal-action record.  The SOLUTION cannot come from the reference (no cvxpy / OSQP here): it is compared with the independent
interior-point solver of the original problem (tests/qp_original.py) with every slack / lambda of every group problem explicit.
"""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
from sigmarl_amd import cbf
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config
from test_oracle_golden import _cbf_fixture

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cbf_grouped.npz")
ENVS = [0, 1, 2, 3, 5, 8, 13, 21, 30, 47]  # the envs of cbf_functions.npz the fixture was recorded on (gen_cbf_grouped.py)


def grouped_fixture():
    z = np.load(GOLDEN)
    return z, json.loads(str(z["meta_json"]))


def fixture_cases():
    """[(first sample, max_group_size, observation_range, nominal)]: the fixture holds len(ENVS) consecutive samples per case."""
    _, meta = grouped_fixture()
    out = []
    for k in range(0, len(meta), len(ENVS)):
        assert [m["env"] for m in meta[k:k + len(ENVS)]] == ENVS
        out.append((k, meta[k]["max_group_size"], meta[k]["observation_range"], meta[k]["nominal"]))
    return out


def grouped_case(make_env, m, rng, nominal, envs=ENVS, N=16):
    """Env on the set states of cbf_functions.npz (rows `envs`) with the grouped-QP constants attached."""
    z, meta = _cbf_fixture()
    B = len(envs)
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=meta["dt"], rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, h_nom=meta["h_nom"],
                   is_obs_noise=False, is_apply_mask=False, nom_controller_type=nominal, adaptive_lambda=True, is_grouping_agents=True, max_group_size=m,
                   observation_range=rng)
    env = make_env(make_config(p, mp, B), mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    env.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
    st8 = np.zeros((B, N, 8), np.float32)
    st8[..., :5] = z["p2_state"][envs, :N]
    ids = np.zeros((B, N, 4), np.int32)
    ids[..., 0] = z["p2_path"][envs, :N]
    ids[..., 2] = z["p2_path"][envs, :N]
    env.reset(np.repeat(np.arange(B), N), np.tile(np.arange(N), B), ids.reshape(-1, 4), st8.reshape(-1, 8), 1)
    return env, z["p2_act"][envs, :N].copy(), z["g10_ref"][envs, :N].copy()


def rows_by_key(con_b, Cc):
    """{(i, j): [Cc*Cc rows]} of the intra-group pair rows and {(i, external j): rows} of the cross rows of one env (row order (ci, cj))."""
    pair, cross = {}, {}
    for row in con_b:
        i, j = int(row[0]), int(row[1])
        if i < 0 or j == -1:
            continue
        (pair if j >= 0 else cross).setdefault((i, j if j >= 0 else -2 - j), []).append(row)
    return pair, cross


def check_grouped_data(env, con, unom, groups, z, k0, Cc=3):
    """Groups, neighbour lists and every row against the reference's data (samples k0 ... of the fixture); returns error statistics."""
    cc = env.cbf_cfg
    N = env.N
    n_exact = n_all = 0
    worst = 0.0
    for b in range(env.B):
        k = k0 + b
        assert np.array_equal(groups[b], z["grp"][k]), (k, groups[b], z["grp"][k])
        pair, cross = rows_by_key(con[b], Cc)
        want_pairs = {(i, j) for i in range(N) for j in range(i + 1, N) if np.isfinite(z["pair6"][k, i, j, 0, 0, 0])}
        assert set(pair) == want_pairs
        want_cross = {(i, j) for i in range(N) for j in range(N) if z["nbr"][k, i, j]}
        assert set(cross) == want_cross, (k, sorted(set(cross) ^ want_cross))
        for (i, j), rows in pair.items():
            got = np.asarray(rows)[:, 2:8].reshape(Cc, Cc, 6)
            want = z["pair6"][k, i, j]
            err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
            worst = max(worst, float(err.max())); n_exact += int((err <= 1e-9).sum()); n_all += err.size
        for (i, j), rows in cross.items():
            got = np.asarray(rows).reshape(Cc, Cc, 8)  # emitted in the order (circle of the lower-numbered vehicle, circle of the higher one)
            if i > j:
                got = got.transpose(1, 0, 2)           # the fixture: [circle of the local vehicle, circle of the external one]
            want = z["cross4"][k, i, j]  # A_i (2), b0, h as the reference hands them to cvxpy; the constraint uses b0 / 2 and rs * h (:1747-1758)
            assert np.all(got[..., 4:6] == 0.0)
            ref = np.stack([want[..., 0], want[..., 1], 0.5 * want[..., 2], cc.rs * want[..., 3]], -1)
            err = np.abs(got[..., [2, 3, 6, 7]] - ref) / np.maximum(1.0, np.abs(ref))
            worst = max(worst, float(err.max())); n_exact += int((err <= 1e-9).sum()); n_all += err.size
        assert np.abs(unom[b] - z["unom"][k]).max() <= 1e-5
    # as for G10 (test_cbf_qp.check_g10): the circle centres are float32 cos / sin of the yaw -- torch's value and the correctly rounded one
    # differ by an ulp now and then, which moves a row by ~1e-7 relative; everything else agrees to rounding
    assert worst <= 2e-6, worst
    assert n_exact / n_all >= 0.80, n_exact / n_all
    return dict(worst=worst, exact=n_exact / n_all)


@pytest.mark.parametrize("case", fixture_cases())
def test_groups_neighbours_and_rows_match_the_reference(case):
    k0, m, rng, nominal = case
    z, _ = grouped_fixture()
    env, act, ref = grouped_case(ob.OracleEnv, m, rng, nominal)
    env.get(4, copy=False)[:, :, 2, :] = ref  # the "clf" controller tracks the third short-term reference point (info "ref"[4:6], :1936)
    safe, u, info, con, unom = env.cbf_qp(act, with_data=True)
    assert info[:, 1].all()
    check_grouped_data(env, con, unom, env.cbf_groups(), z, k0)
    # world_state.nominal_action_{vel,steer} after the update (:2254-2268): the clamped policy action ("rl"), U_nom itself ("clf")
    assert np.abs(env.get(20) - z["nom"][k0:k0 + env.B]).max() <= 1e-6
    env.close()


def test_groups_are_kept_until_regroup():
    """use_fixed_groups (:1897-1909): the partition of the first call stays although the vehicles move; sigmaenv_cbf_regroup forms it again."""
    env, act, _ = grouped_case(ob.OracleEnv, 3, 0.5, "rl")
    env.cbf_qp(act)
    g0 = env.cbf_groups()
    st = env.get(0, copy=False)
    st[:, :, 0] = st[:, ::-1, 0].copy()  # shuffle the positions
    st[:, :, 1] = st[:, ::-1, 1].copy()
    env.cbf_qp(act)
    assert np.array_equal(env.cbf_groups(), g0)
    env.cbf_regroup()
    env.cbf_qp(act)
    g1 = env.cbf_groups()
    assert not np.array_equal(g1, g0)
    for b in range(env.B):  # K = ceil(16 / 3) = 6 groups of at most 3, ordered by their first member
        sizes = np.bincount(g1[b], minlength=6)
        assert sizes.max() <= 3 and sizes.min() >= 1 and len(sizes) == 6
        firsts = [int(np.flatnonzero(g1[b] == g)[0]) for g in range(6)]
        assert firsts == sorted(firsts)
    env.close()


def solve_grouped_original(env, con_b, unom_b, nominal, b, Cc=3):
    """u* of ALL group problems of one env (they share no variable: one block-separable problem) by the interior-point solver of the
    original form: every slack and lambda explicit, per-row weights (lane / pair / cross)."""
    from qp_original import build_original_qp, solve_original

    N = env.N
    cfg, cc = env.cfg, env.cbf_cfg
    rows = con_b[con_b[:, 0] >= 0]
    j = rows[:, 1].astype(int)
    ws = np.where(j == -1, cc.qp_w_lane, np.where(j >= 0, cc.qp_w_pair, cc.qp_w_cross))
    wl = np.where(j <= -2, cc.qp_w_lambda_cross, cc.qp_w_lambda)
    rows = rows.copy()
    rows[j <= -2, 1] = -1  # (the builder's "one vehicle" marker)
    lo = np.tile([cfg.min_acc, -cc.steering_rate_max], N).astype(np.float64)
    hi = np.tile([cfg.max_acc, cc.steering_rate_max], N).astype(np.float64)
    w = np.tile([cc.qp_w_acc, cc.qp_w_steer], N)
    clf_e, clf_v = np.zeros(2 * N), np.zeros(2 * N)
    if nominal == "clf":
        st, short = env.get(0), env.get(4)
        desired = np.arctan2(short[b, :, 2, 1].astype(np.float64) - st[b, :, 1], short[b, :, 2, 0].astype(np.float64) - st[b, :, 0])
        e_h = (desired - st[b, :, 2].astype(np.float64) + np.pi) % (2 * np.pi) - np.pi
        e_v = cc.ref_speed - st[b, :, 3].astype(np.float64)
        clf_e = np.stack([e_v, e_h], -1).reshape(-1)
        clf_v = cc.lam_clf * 0.5 * clf_e ** 2
    P, q, A, l, uu, n = build_original_qp(rows, unom_b.reshape(-1), lo, hi, w, 0.0, 0.0, 0.0, 0, clf_e, clf_v, cc.qp_w_clf, ws_rows=ws, wl_rows=wl)
    x, inf = solve_original(P, q, A, l, uu)
    assert inf["primal"] <= 1e-8 and inf["dual"] <= 1e-6 and inf["gap"] <= 1e-12, inf
    return x[:n]


@pytest.mark.parametrize("case", [(2, 1.0, "rl"), (4, 0.5, "clf"), (3, 1.5, "rl")])
def test_grouped_solution_matches_an_independent_solver_of_the_original_problems(case):
    m, rng, nominal = case
    env, act, ref = grouped_case(ob.OracleEnv, m, rng, nominal)
    env.get(4, copy=False)[:, :, 2, :] = ref
    safe, u, info, con, unom = env.cbf_qp(act, with_data=True)
    assert info[:, 1].all()
    worst, n_cross_active = 0.0, 0
    for b in range(env.B):
        x = solve_grouped_original(env, con[b], unom[b], nominal, b)
        worst = max(worst, float(np.abs(x - u[b].reshape(-1)).max()))
    assert worst <= 1e-5, worst  # OSQP's tolerance in the reference; measured ~1e-8
    assert (np.abs(u - unom).max(axis=(1, 2)) > 1e-6).sum() >= 3  # the filter acts
    env.close()
