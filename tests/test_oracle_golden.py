"""Pins the CPU oracle against golden vectors produced by RUNNING the reference (tests/golden/gen/gen_golden.py).

Bar: bit-exact for masks and indices, <= 1e-5 abs for fp32 states / rewards / observations -- the mtv distance included (observed
worst case 4.3e-6): the reference's formulation projects world coordinates (~4.5 m) onto axes normalised from
0.107 m edges, which amplifies the 1-ulp difference between torch's and the correctly rounded trig ~40x ("Tolerances").
"""
import ctypes as C
import os

import json

import numpy as np
import pytest

import oracle_binding as ob
import traj_replay as tr
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config

FTOL = 1e-5
MTV_TOL = 1e-5


@pytest.fixture(scope="module")
def fn():
    return np.load(os.path.join(tr.GOLDEN_DIR, "functions.npz"))


def _cfg(dt=0.05):
    return make_config(Parameters(n_agents=2, dt=dt, scenario_type="cpm_entire", is_apply_mask=False, is_use_mtv_distance=False),
                       load_map("cpm_entire"), 1)


@pytest.mark.parametrize("tag,dt", [("dt05", 0.05), ("dt10", 0.1)])
def test_g1_bicycle_step(oracle_lib, fn, tag, dt):
    """WorldCustom.step + KinematicBicycleModel.step (helper_training.py:797-861, dynamics.py:62-192)."""
    x = np.ascontiguousarray(fn[f"g1_{tag}_in"], np.float32)
    want = fn[f"g1_{tag}_out"]  # pos2, rot, speed, steering, vel2, sideslip, clamped action 2
    out = np.zeros((len(x), 10), np.float32)
    cfg = _cfg(dt)
    oracle_lib.fn_bicycle(C.byref(cfg), len(x), ob.ptr(x), ob.ptr(out))
    assert np.abs(out - want).max() <= FTOL
    # clamps and the [-pi, pi) steering wrap were exercised
    assert (np.abs(x[:, 5]) > 1.0).any() and (np.abs(x[:, 6]) > capi.AGENTS["max_steering"]).any()
    assert np.array_equal(out[:, 8:10], want[:, 8:10])  # clamped actions are exact


def test_g2_vertices(oracle_lib, fn):
    x = np.ascontiguousarray(fn["g2_in"], np.float32)
    out = np.zeros((len(x), 5, 2), np.float32)
    cfg = _cfg()
    oracle_lib.fn_vertices(C.byref(cfg), len(x), ob.ptr(x), ob.ptr(out))
    assert np.abs(out - fn["g2_out"]).max() <= FTOL


def test_g5_g6_mtv_c2c_interx_rect(oracle_lib, fn):
    v = np.ascontiguousarray(fn["g5_vertices"], np.float32)
    n = len(v)
    d = np.zeros(n, np.float32)
    oracle_lib.fn_mtv(n, ob.ptr(v), ob.ptr(d))
    want = fn["g5_mtv"][:, 0, 1]
    assert np.abs(d - want).max() <= MTV_TOL
    # sign structure (overlap <=> negative) is exact on this grid
    assert np.array_equal(d < 0, want < 0)
    assert (want < 0).sum() > 1000
    pos = fn["g5_pos"]
    c2c = np.sqrt(((pos[:, 0] - pos[:, 1]) ** 2).sum(-1, dtype=np.float32)).astype(np.float32)
    # torch.sqrt on large contiguous tensors goes through MKL VML (<= 1 ulp, not correctly rounded; measured in the build
    # container), so the c2c distance is pinned to 1 ulp, not bit-exact.  No mask or index depends on it.
    assert np.abs(c2c - fn["g5_c2c"][:, 0, 1]).max() <= 1.2e-7
    hit = np.zeros(n, np.uint8)
    v0 = np.ascontiguousarray(v[:, 0])
    v1 = np.ascontiguousarray(v[:, 1])
    oracle_lib.fn_interx(n, ob.ptr(v0), 5, 10, ob.ptr(v1), 5, 10, ob.ptr(hit))
    assert np.array_equal(hit.astype(bool), fn["g6_hit"])  # same fp32 inputs -> bit-exact predicate
    assert fn["g6_hit"].sum() > 1000


def test_g3_g4_g6_polylines(oracle_lib, fn):
    cfg = _cfg()
    for pid in (0, 5, 17, 39):
        sel = fn["g3_path"] == pid
        pts = np.ascontiguousarray(fn["g3_pts"][sel], np.float32)
        n_lt, n_l, n_r, is_loop = [int(x) for x in fn[f"g3_path{pid}_n"]]
        m = len(pts)
        for key, n_pts, dk, ik in (("long_term", n_lt, "g3_d_ref", "g3_i_ref"), ("left", n_l, "g3_d_left", "g3_i_left"),
                                   ("right", n_r, "g3_d_right", "g3_i_right")):
            poly = np.ascontiguousarray(fn[f"g3_path{pid}_{key}"], np.float32)
            d = np.zeros(m, np.float32)
            idx = np.zeros(m, np.int32)
            oracle_lib.fn_point_polyline(m, ob.ptr(pts), ob.ptr(poly), n_pts, ob.ptr(d), ob.ptr(idx))
            assert np.array_equal(idx, fn[ik][sel]), (pid, key)   # indices bit-exact (same fp32 inputs, same arithmetic)
            assert np.array_equal(d, fn[dk][sel]), (pid, key)     # and so are the distances
        lt = np.ascontiguousarray(fn[f"g3_path{pid}_long_term"], np.float32)
        cp = np.ascontiguousarray(fn["g3_i_ref"][sel], np.int32)
        st = np.zeros((m, 3, 2), np.float32)
        oracle_lib.fn_short_term(m, ob.ptr(lt), n_lt, is_loop, ob.ptr(cp), ob.ptr(st))
        assert np.array_equal(st, fn["g4_short_term"][sel])
        # rectangle vs lane boundary: vertices from the oracle (trig differs by <= 1 ulp), predicate compared where the reference
        # rectangle is not within 1e-5 of touching; with the reference's own vertices it must be exact.
        rot = fn["g3_rot"][sel]
        from_ref = np.zeros((m, 5, 2), np.float32)
        x3 = np.ascontiguousarray(np.concatenate([pts, rot], axis=1), np.float32)
        oracle_lib.fn_vertices(C.byref(cfg), m, ob.ptr(x3), ob.ptr(from_ref))
        for key, n_pts, hk in (("left", n_l, "g6_hit_left"), ("right", n_r, "g6_hit_right")):
            poly = np.ascontiguousarray(fn[f"g3_path{pid}_{key}"], np.float32)
            hit = np.zeros(m, np.uint8)
            oracle_lib.fn_interx(m, ob.ptr(from_ref), 5, 10, ob.ptr(poly), len(poly), 0, ob.ptr(hit))
            assert (hit.astype(bool) != fn[hk][sel]).sum() <= 1, (pid, key)


def test_g8_ego_transform_and_wrap(oracle_lib, fn):
    pi_, pj, ri = [np.ascontiguousarray(fn[k], np.float32) for k in ("g8_pos_i", "g8_pos_j", "g8_rot_i")]
    out = np.zeros_like(pj)
    oracle_lib.fn_ego(len(pi_), ob.ptr(pi_), ob.ptr(ri), pj.shape[1], ob.ptr(pj), ob.ptr(out))
    assert np.abs(out - fn["g8_rel"]).max() <= FTOL
    a = np.ascontiguousarray(fn["g8_angle_in"], np.float32)
    w = np.zeros_like(a)
    oracle_lib.fn_wrap(len(a), ob.ptr(a), ob.ptr(w))
    assert np.array_equal(w, fn["g8_angle_out"])  # fmod-based wrap is exact


def test_path_table_padding_matches_reference(fn):
    """Padded per-path polylines == the reference's per-(env, agent) copies (world_state_rt.py:279-420)."""
    cfg = _cfg()
    env = ob.OracleEnv(cfg, load_map("cpm_entire"))
    center, left, right = env.path_table()
    for pid in (0, 5, 17, 39):
        assert np.array_equal(center[pid], fn[f"g3_path{pid}_long_term"])
        assert np.array_equal(left[pid], fn[f"g3_path{pid}_left"])
        assert np.array_equal(right[pid], fn[f"g3_path{pid}_right"])
    env.close()


@pytest.mark.parametrize("name", tr.TRAJ_NAMES)
def test_trajectory(name):
    """End-to-end: every hot-path tensor after every step of a reference rollout, incl. replayed resets."""
    z, meta = tr.load_fixture(name)
    cfg, mp = tr.config_from_meta(meta)
    env = ob.OracleEnv(cfg, mp)
    rep = tr.replay(env, z, meta, mp)
    env.close()
    assert rep.total_mismatch() == 0, str(rep)  # masks, indices, counters, done: bit-exact
    if rep.cbf_count:
        print(f"{name}: {rep}")  # observed worst and outlier count per CBF quantity (pytest -s / on failure)
    assert rep.cbf_ok(name), str(rep)
    for key, err in rep.max_abs.items():
        tol = MTV_TOL if (meta["is_use_mtv_distance"] and key in ("dist_agents", "obs", "reward", "rew_total", "rew_near_other_agents")) else FTOL
        assert err <= tol, (key, err, str(rep))


def _cbf_fixture():
    z = np.load(os.path.join(tr.GOLDEN_DIR, "cbf_functions.npz"))
    return z, json.loads(str(z["meta_json"]))


def test_pseudo_distance_f16_bit_exact():
    """PseudoDistance.get_distance (sigmarl/pseudo_distance.py:204-242): fp16 bit patterns of 9000 points x 2 boundaries."""
    from sigmarl_amd import cbf

    z, _ = _cbf_fixture()
    mp = load_map("cpm_entire")
    seg_l, seg_r = cbf.load_segment_tables(mp)
    lib = ob.load_oracle()
    for pid in np.unique(z["p1_path"]):
        m = z["p1_path"] == pid
        pts = np.ascontiguousarray(z["p1_pts"][m])
        for poly, n, seg, want in ((mp.left, mp.n_left, seg_l, z["p1_left_f16"][m]), (mp.right, mp.n_right, seg_r, z["p1_right_f16"][m])):
            out = np.zeros(len(pts), np.uint16)
            lib.fn_pseudo_distance(len(pts), ob.ptr(pts), ob.ptr(np.ascontiguousarray(poly[pid])), ob.ptr(np.ascontiguousarray(seg[pid])),
                                   int(n[pid]), ob.ptr(out))
            assert np.array_equal(out, want), (int(pid), int((out != want).sum()))


def cbf_case_env(make_env, z, meta):
    """Env with the directly set states of the ``p2`` case of cbf_functions.npz (shared with the GPU test)."""
    from sigmarl_amd import cbf
    from sigmarl_amd.params import Parameters, make_config

    B, N = meta["B"], meta["N"]
    mp = load_map("cpm_entire")
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=meta["dt"], rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True,
                   h_nom=meta["h_nom"], is_obs_noise=False, is_apply_mask=False)
    env = make_env(make_config(p, mp, B), mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    env.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
    st8 = np.zeros((B, N, 8), np.float32)
    st8[..., :5] = z["p2_state"]
    ids = np.zeros((B, N, 4), np.int32)
    ids[..., 0] = z["p2_path"]
    ids[..., 2] = z["p2_path"]
    env.reset(np.repeat(np.arange(B), N), np.tile(np.arange(N), B), ids.reshape(-1, 4), st8.reshape(-1, 8), 1)
    return env


def test_cbf_margins_on_set_states():
    """compute_nominal_cbf_constraint_margins + compute_cbf_violation_rewards_from_margins (sigmarl/cbf_qp.py:2562-2804) on 48 x 16
    directly set vehicle states, incl. close pairs and agents far off their lane."""
    z, meta = _cbf_fixture()
    env = cbf_case_env(ob.OracleEnv, z, meta)
    lane_l, lane_r, pair = env.cbf_rewards(z["p2_act"])
    rep = tr.Report()
    rep.cbf("lane_left", lane_l, z["p2_lane_left"])
    rep.cbf("lane_right", lane_r, z["p2_lane_right"])
    rep.cbf("pair", pair, z["p2_pair"])
    ri = env.get(capi.BUF_REWARD_INFO)
    rep.cbf("rew", np.stack([ri[5], ri[6], ri[4]]), z["p2_rew"])
    # with the reference's own float32 circle centres injected, no exception is needed: every margin and channel within CBF_TOL
    env.cbf_inject_centers(z["p2_centers"])
    lane_l, lane_r, pair = env.cbf_rewards(z["p2_act"])
    rep.cbf("inj_lane_left", lane_l, z["p2_lane_left"])
    rep.cbf("inj_lane_right", lane_r, z["p2_lane_right"])
    rep.cbf("inj_pair", pair, z["p2_pair"])
    ri = env.get(capi.BUF_REWARD_INFO)
    rep.cbf("inj_rew", np.stack([ri[5], ri[6], ri[4]]), z["p2_rew"])
    env.close()
    print(f"cbf_functions: {rep}")
    assert rep.cbf_ok("cbf_functions"), str(rep)
    assert all(rep.cbf_bad[k] == 0 for k in rep.cbf_bad if k.startswith("inj_")), str(rep)
    assert (z["p2_pair"] < 0).sum() > 100 and (z["p2_lane_left"] < 0).sum() > 100  # the case exercises violations


def test_own_segment_tables_close_to_reference_tables():
    """sigmarl_amd.cbf.segment_tables (numpy) against the shipped tables (the reference's torch ops): identical lengths, angles within an ulp."""
    from sigmarl_amd import cbf

    mp = load_map("intersection_1")
    seg_l, _ = cbf.load_segment_tables(mp)
    for p in range(mp.n_paths):
        n = int(mp.n_left[p])
        own = cbf.segment_tables(mp.left[p, :n])
        ref = seg_l[p, : n - 1]
        assert np.array_equal(own[:, 4], ref[:, 4])
        assert np.abs(own[:, :2] - ref[:, :2]).max() <= 1.2e-7
        assert np.abs(own[:, 2:4] - ref[:, 2:4]).max() <= 1e-5


def test_trajectory_fixtures_cover_events():
    """The goldens exercise what they claim: collisions, resets, non-loop exits, clamps."""
    tot = dict(n_done=0, n_col_agents=0, n_col_lane=0, n_exit=0, n_events=0)
    for name in tr.TRAJ_NAMES:
        _, meta = tr.load_fixture(name)
        for k in tot:
            tot[k] += meta[k]
    assert tot["n_done"] > 50 and tot["n_col_lane"] > 50 and tot["n_col_agents"] > 10 and tot["n_exit"] >= 1


def test_adversarial_collinear_configurations_match_the_reference():
    """tests/golden/adversarial.npz: rectangle edges exactly collinear with long straight boundary stretches (far boundary segments collinear with
    an edge), offsets of a few ulps around touching, vehicles in line with collinear side edges at centre distances around and beyond the
    circumcircle sum -- the configurations in which a PRUNED far segment / far rectangle could change the strict-sign interX predicate
    (helper_scenario.py:1148-1229).  The oracle scans everything: with the reference's own vertices its flags must equal the reference's."""
    z = np.load(os.path.join(tr.GOLDEN_DIR, "adversarial.npz"))
    lib = ob.load_oracle()
    mp = load_map("cpm_entire")
    env = ob.OracleEnv(_cfg(), mp)
    _, left, right = env.path_table()
    env.close()
    first = mp.list_first[0]
    for pid in np.unique(z["b_path"]):
        sel = z["b_path"] == pid
        v = np.ascontiguousarray(z["b_vertices"][sel], np.float32)
        for poly, want in ((left[first + pid], z["b_hit_left"][sel]), (right[first + pid], z["b_hit_right"][sel])):
            hit = np.zeros(len(v), np.uint8)
            poly = np.ascontiguousarray(poly, np.float32)
            lib.fn_interx(len(v), ob.ptr(v), 5, 10, ob.ptr(poly), len(poly), 0, ob.ptr(hit))
            assert np.array_equal(hit.astype(bool), want), (int(pid), int((hit.astype(bool) != want).sum()))
    va, vb = np.ascontiguousarray(z["r_vertices_a"], np.float32), np.ascontiguousarray(z["r_vertices_b"], np.float32)
    hit = np.zeros(len(va), np.uint8)
    lib.fn_interx(len(va), ob.ptr(va), 5, 10, ob.ptr(vb), 5, 10, ob.ptr(hit))
    assert np.array_equal(hit.astype(bool), z["r_hit"])
    assert z["b_hit_left"].sum() > 500 and (~z["b_hit_left"]).sum() > 500 and z["r_hit"].sum() > 50


def test_fixtures_match_the_committed_manifest():
    """tests/golden/MANIFEST.json (written by the ONE generation command, `python tests/golden/gen/gen_golden.py`, which runs every fixture in
    its own interpreter) holds a content hash of every fixture: the committed bytes are the recipe's output, not an earlier revision's."""
    import hashlib

    man = json.load(open(os.path.join(tr.GOLDEN_DIR, "MANIFEST.json")))
    names = sorted(f for f in os.listdir(tr.GOLDEN_DIR) if f.endswith(".npz"))
    assert sorted(man) == names
    for f in names:
        z = np.load(os.path.join(tr.GOLDEN_DIR, f))
        h = hashlib.sha256()
        for k in sorted(z.files):
            a = np.ascontiguousarray(z[k])
            h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
        assert h.hexdigest() == man[f], f


def test_opponent_fill_restates_the_reference_gather():
    """sigmaenv_opponent_fill against opponent_modeling's own loops (helper_training.py:1117-1137: obs[:, ego, -(K - j) * 2 : ...] = actions[arange(B),
    nearing[:, ego, j]]) on the oracle; the placeholder columns themselves are pinned on the reference trajectory cpm8_opponent_pad."""
    N, B, K = 8, 5, 2
    mp = load_map("cpm_entire")
    cfg = make_config(Parameters(n_agents=N, scenario_type="cpm_entire", is_obs_noise=False, is_using_opponent_modeling=True), mp, B)
    env = ob.OracleEnv(cfg, mp)
    env.get(capi.BUF_DONE, copy=False)[:] = 1
    env.auto_reset(1, 0, mp.list_first[0], mp.list_count[0])
    obs = env.get(capi.BUF_OBS)
    assert obs.shape == (B, N, 32 + 2 * K) and (obs[..., -2 * K:] == 0).all()
    tentative = np.random.default_rng(0).normal(size=(B, N, 2)).astype(np.float32)
    env.opponent_fill(tentative)
    near = env.get(capi.BUF_NEARING).astype(np.int64)
    want = obs.copy()
    for ego in range(N):
        for j in range(K):
            start = -(K - j) * 2
            end = start + 2
            want[:, ego, start:(end if end != 0 else None)] = tentative[np.arange(B), near[:, ego, j]]
    np.testing.assert_array_equal(env.get(capi.BUF_OBS), want)
    env.close()
