"""ctypes binding of the CPU oracle (oracle/libsigmaenv_oracle.so) for the test-suite.

Test infrastructure: imported by tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg only.
Uses the product's ABI definitions (``sigmarl_amd.capi``) so oracle and HIP library are driven identically.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from sigmarl_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libsigmaenv_oracle.so")

_f32p = C.POINTER(C.c_float)


def oracle_path(n_short_term: int = capi.N_SHORT_TERM) -> str:
    return ORACLE_SO if n_short_term == capi.N_SHORT_TERM else os.path.join(ORACLE_DIR, f"libsigmaenv_oracle_ns{int(n_short_term)}.so")


def build_oracle(force: bool = False, n_short_term: int = capi.N_SHORT_TERM) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("sigmaenv_oracle.c", "sigmaenv_cbf_oracle.inc")] + [os.path.join(ROOT, "include", f) for f in ("sigmaenv.h", "sigma_trig_f32.h", "sigmaenv_ref_weights.h")]
    so = oracle_path(n_short_term)
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", f"NS={int(n_short_term)}", os.path.basename(so)], stdout=subprocess.DEVNULL)
    return so


_libs = {}


def load_oracle(n_short_term: int = capi.N_SHORT_TERM) -> capi.Library:
    """The oracle built for ``n_points_short_term`` (a build constant of both libraries, include/sigmaenv.h)."""
    n_short_term = int(n_short_term or capi.N_SHORT_TERM)
    _lib = _libs.get(n_short_term)
    if _lib is None:
        build_oracle(n_short_term=n_short_term)
        extra = {
            "fn_bicycle": (None, [C.POINTER(capi.Config), C.c_int, C.c_void_p, C.c_void_p]),
            "fn_vertices": (None, [C.POINTER(capi.Config), C.c_int, C.c_void_p, C.c_void_p]),
            "fn_point_polyline": (None, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
            "fn_short_term": (None, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
            "fn_interx": (None, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
            "fn_mtv": (None, [C.c_int, C.c_void_p, C.c_void_p]),
            "fn_ego": (None, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
            "fn_wrap": (None, [C.c_int, C.c_void_p, C.c_void_p]),
            "fn_trig": (None, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
            "fn_pseudo_distance": (None, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
            "cbf_qp_ex": (C.c_int, [C.c_void_p] * 7),
            "fn_qp_vanish_tol": (C.c_double, [C.c_double]),
            "env0_reset_side_effect": (C.c_int, [C.c_void_p, C.c_int32]),
            "path_table": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(_f32p), C.POINTER(_f32p), C.POINTER(_f32p)]),
        }
        _lib = capi.Library(oracle_path(n_short_term), "sigmaenv_oracle_", extra)
        assert _lib.n_short_term() == n_short_term
        _libs[n_short_term] = _lib
    return _lib


def ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


_BUF_SPEC = {
    capi.BUF_STATE: (np.float32, lambda B, N, K, D: (B, N, 8)),
    capi.BUF_PREV_POS: (np.float32, lambda B, N, K, D: (B, N, 2)),
    capi.BUF_VERTICES: (np.float32, lambda B, N, K, D: (B, N, 5, 2)),
    capi.BUF_PATH: (np.int32, lambda B, N, K, D: (B, N, 4)),
    capi.BUF_SHORT_TERM: (np.float32, lambda B, N, K, D: (B, N, 3, 2)),
    capi.BUF_DIST_REF: (np.float32, lambda B, N, K, D: (B, N)),
    capi.BUF_DIST_LEFT: (np.float32, lambda B, N, K, D: (B, N, 5)),
    capi.BUF_DIST_RIGHT: (np.float32, lambda B, N, K, D: (B, N, 5)),
    capi.BUF_DIST_BOUND: (np.float32, lambda B, N, K, D: (B, N)),
    capi.BUF_CLOSEST: (np.int32, lambda B, N, K, D: (B, N, 3)),
    capi.BUF_DIST_AGENTS: (np.float32, lambda B, N, K, D: (B, N, N)),
    capi.BUF_COL_AGENTS: (np.uint8, lambda B, N, K, D: (B, N, N)),
    capi.BUF_COL_FLAGS: (np.uint8, lambda B, N, K, D: (B, N, 4)),
    capi.BUF_REWARD: (np.float32, lambda B, N, K, D: (B, N)),
    capi.BUF_REWARD_INFO: (np.float32, lambda B, N, K, D: (12, B, N)),
    capi.BUF_OBS: (np.float32, lambda B, N, K, D: (B, N, D)),
    capi.BUF_NEARING: (np.int32, lambda B, N, K, D: (B, N, K)),
    capi.BUF_DONE: (np.uint8, lambda B, N, K, D: (B,)),
    capi.BUF_TIMER: (np.int32, lambda B, N, K, D: (B, 4)),
    capi.BUF_ACTION: (np.float32, lambda B, N, K, D: (B, N, 2)),
    capi.BUF_CBF_NOMINAL: (np.float32, lambda B, N, K, D: (B, N, 2)),
}


def buf_spec(which, B, N, K, D, n_short_term=capi.N_SHORT_TERM):
    dt, shp = _BUF_SPEC[which]
    if which == capi.BUF_SHORT_TERM:
        return dt, (B, N, n_short_term, 2)
    return dt, shp(B, N, K, D)


class OracleEnv:
    """Host-memory twin of ``sigmarl_amd.env.SigmaEnv`` backed by the C oracle."""

    def __init__(self, cfg: capi.Config, map_table):
        self.n_short_term = int(getattr(cfg, "n_points_short_term", 0) or capi.N_SHORT_TERM)
        self.lib = load_oracle(self.n_short_term)
        self.cfg = cfg
        self.map = map_table
        self._map_struct = map_table.as_struct()
        self.B, self.N, self.K = cfg.n_envs, cfg.n_agents, cfg.n_nearing
        self.D = capi.obs_dim(self.K, int(getattr(self.cfg, "obs_flags", 0)), self.n_short_term, self.N)
        h = C.c_void_p()
        rc = self.lib.create(C.byref(cfg), C.byref(self._map_struct), 0, None, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_oracle_create failed: {rc}")
        self.h = h
        if (int(getattr(cfg, "obs_flags", 0)) & capi.OBS_BIRD_VIEW) and cfg.is_apply_mask and map_table.lanelet_tables() is not None:
            centers, neigh = map_table.lanelet_tables()  # the lanelet-relation mask of the bird-view observation (map_manager.py:41-118)
            assert self.lib.set_lanelets(self.h, int(centers.shape[0]), int(centers.shape[1]), ptr(centers), ptr(neigh)) == 0

    def close(self):
        if self.h:
            self.lib.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, env_idx, agent_idx, path_ids, state8, full_env):
        env_idx = np.ascontiguousarray(env_idx, np.int32)
        agent_idx = np.ascontiguousarray(agent_idx, np.int32)
        path_ids = np.ascontiguousarray(path_ids, np.int32).reshape(-1, 4)
        state8 = np.ascontiguousarray(state8, np.float32).reshape(-1, 8)
        n = len(env_idx)
        rc = self.lib.reset(self.h, n, ptr(env_idx), ptr(agent_idx), ptr(path_ids), ptr(state8), int(full_env))
        if rc != 0:
            raise RuntimeError(f"oracle reset failed: {rc} {self.lib.last_error(self.h)}")

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)
        rc = self.lib.step(self.h, ptr(a))
        if rc != 0:
            raise RuntimeError(f"oracle step failed: {rc}")

    def env0_reset_side_effect(self, agent: int):
        """The reference's `if env_index:` quirk (road_traffic.py:889-907): see sigmaenv_oracle_env0_reset_side_effect."""
        rc = self.lib.env0_reset_side_effect(self.h, int(agent))
        if rc != 0:
            raise RuntimeError(f"oracle env0_reset_side_effect failed: {rc}")

    def observe(self):
        assert self.lib.observe(self.h) == 0

    def set_scenario_lists(self, probabilities):
        n = len(probabilities)
        first = np.asarray([self.map.list_first[k + 1] for k in range(n)], np.int32)
        count = np.asarray([self.map.list_count[k + 1] for k in range(n)], np.int32)
        pr = np.asarray(probabilities, np.float32)
        assert self.lib.set_scenario_lists(self.h, n, ptr(first), ptr(count), ptr(pr)) == 0

    def opponent_fill(self, actions):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)
        assert self.lib.opponent_fill(self.h, ptr(a)) == 0

    def cbf_attach(self, cbf_cfg, seg_left, seg_right):
        seg_left = np.ascontiguousarray(seg_left, np.float32)
        seg_right = np.ascontiguousarray(seg_right, np.float32)
        self.cbf_cfg = cbf_cfg
        rc = self.lib.cbf_attach(self.h, C.byref(cbf_cfg), ptr(seg_left), ptr(seg_right), int(seg_left.shape[1]))
        if rc != 0:
            raise RuntimeError(f"oracle cbf_attach failed: {rc}")

    def cbf_inject_centers(self, centers):
        """Test hook: the covering-circle centres ([B,N,C,2] float32, e.g. the reference's own from a CBF golden) replace the computed ones; None clears."""
        self._centers = None if centers is None else np.ascontiguousarray(centers, np.float32)
        assert self.lib.cbf_inject_centers(self.h, ptr(self._centers) if self._centers is not None else None) == 0

    def cbf_rewards(self, actions, want_margins=True):
        from sigmarl_amd.cbf import split_cbf_margins

        a = np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)
        Cc = int(self.cbf_cfg.n_circles)
        m = np.full(2 * self.B * self.N * Cc + self.B * self.N * self.N * Cc * Cc, np.nan, np.float64) if want_margins else None
        rc = self.lib.cbf_rewards(self.h, ptr(a), ptr(m) if m is not None else None)
        if rc != 0:
            raise RuntimeError(f"oracle cbf_rewards failed: {rc}")
        return None if m is None else split_cbf_margins(m, self.B, self.N, Cc)

    def cbf_qp(self, actions, with_data=False):
        """(actions_safe f32 [B,N,2], u_opt f64 [B,N,2], info i32 [B,2]) and, with_data, the constraint rows [B,n_con,8] and u_nom."""
        a = np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)
        Cc = int(self.cbf_cfg.n_circles)
        ncon = self.N * Cc * 2 + (2 if int(self.cbf_cfg.is_grouping) else 1) * (self.N * (self.N - 1) // 2 * Cc * Cc)  # (rows beyond an env's count: i = -1)
        safe = np.zeros((self.B, self.N, 2), np.float32)
        u = np.zeros((self.B, self.N, 2), np.float64)
        info = np.zeros((self.B, 2), np.int32)
        con = np.zeros((self.B, ncon, 8), np.float64) if with_data else None
        unom = np.zeros((self.B, self.N, 2), np.float64) if with_data else None
        rc = self.lib.cbf_qp_ex(self.h, ptr(a), ptr(safe), ptr(u), ptr(info), ptr(con) if with_data else None, ptr(unom) if with_data else None)
        if rc != 0:
            raise RuntimeError(f"oracle cbf_qp failed: {rc}")
        return (safe, u, info, con, unom) if with_data else (safe, u, info)

    def cbf_groups(self):
        """[B,N] group index of every vehicle (grouped CBF-QPs; formed at the first cbf_qp call)."""
        g = np.zeros((self.B, self.N), np.int32)
        rc = self.lib.cbf_get_groups(self.h, ptr(g))
        if rc != 0:
            raise RuntimeError(f"oracle cbf_get_groups failed: {rc}")
        return g

    def cbf_regroup(self):
        self.lib.cbf_regroup(self.h)

    def auto_reset(self, seed, counter, path_first, path_count):
        rc = self.lib.auto_reset(self.h, int(seed), int(counter), int(path_first), int(path_count))
        if rc != 0:
            raise RuntimeError(f"oracle auto_reset failed: {rc}")

    def get(self, which, copy=True):
        p = C.c_void_p()
        nb = C.c_size_t()
        rc = self.lib.get(self.h, int(which), C.byref(p), C.byref(nb))
        if rc != 0:
            raise RuntimeError(f"oracle get({which}) failed: {rc}")
        dt, shp = buf_spec(which, self.B, self.N, self.K, self.D, self.n_short_term)
        n = int(np.prod(shp))
        assert n * np.dtype(dt).itemsize == nb.value, (which, shp, nb.value)
        if n == 0:
            return np.zeros(shp, dt)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).reshape(shp)
        return arr.copy() if copy else arr

    def path_table(self):
        P = C.c_int32()
        c, l, r = _f32p(), _f32p(), _f32p()
        assert self.lib.path_table(self.h, C.byref(P), C.byref(c), C.byref(l), C.byref(r)) == 0
        n = self.map.n_paths * P.value * 2
        shp = (self.map.n_paths, P.value, 2)
        return tuple(np.ctypeslib.as_array(x, shape=(n,)).reshape(shp).copy() for x in (c, l, r))
