"""Host side of the QP-free CBF margin reward.

Mirrors ``CBFQP.update_qp`` with ``is_solve_qp=False`` (reference ``sigmarl/cbf_qp.py:2534-2560``): nominal CBF constraint
margins (``compute_nominal_cbf_constraint_margins`` :2562-2760) and the three per-agent reward channels derived from them
(``compute_cbf_violation_rewards_from_margins`` :2762-2804), for every env of the batch in one launch instead of one Python
object per env (``helper_training.py:1620-1627``).
"""
from __future__ import annotations

import os

import numpy as np

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "maps", "pseudo")
SEG_FIELDS = 5  # cos, sin, m_b, m_t, length


def pseudo_tangent_vectors(poly: np.ndarray) -> np.ndarray:
    """``compute_pseudo_tangent_vector`` (reference ``sigmarl/helper_scenario.py:1369-1399``), float32."""
    p = np.asarray(poly, np.float32)
    t = np.zeros_like(p)
    if len(p) >= 2:
        t[0] = p[1] - p[0]
        t[-1] = p[-1] - p[-2]
    if len(p) >= 3:
        t[1:-1] = p[2:] - p[:-2]
    return t


def segment_tables(poly: np.ndarray) -> np.ndarray:
    """Per-segment constants of the pseudo distance: ``[n-1, 5]`` float32 = (cos, sin, m_b, m_t, length).

    Restates the map-only part of ``PseudoDistance.get_pseudo_distance`` (reference ``sigmarl/pseudo_distance.py:43-56,94-103,
    174-177``) in numpy float32, one rounding per reference op.  ``atan2`` / ``cos`` / ``sin`` are correctly rounded here
    while torch's vectorised ones are within 1 ulp, so a few entries can differ from the shipped assets by one ulp
    (``tests/golden/gen/gen_pseudo_tables.py`` prints the count); ``load_segment_tables`` prefers the asset.
    """
    f32 = np.float32
    p = np.asarray(poly, f32)
    t = pseudo_tangent_vectors(p)
    d = p[1:] - p[:-1]
    theta = np.arctan2(d[:, 1].astype(np.float64), d[:, 0].astype(np.float64)).astype(f32)
    c = np.cos(theta.astype(np.float64)).astype(f32)
    s = np.sin(theta.astype(np.float64)).astype(f32)

    def slope(tv):
        x = (c * tv[:, 0]) + (s * tv[:, 1])
        y = ((-s) * tv[:, 0]) + (c * tv[:, 1])
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(x != 0, y / x, f32(1e-8)).astype(f32)

    m_b, m_t = slope(t[:-1]), slope(t[1:])
    xx = (d[:, 0] * d[:, 0]).astype(f32)
    length = np.sqrt((xx.astype(np.float64) + d[:, 1].astype(np.float64) * d[:, 1].astype(np.float64)).astype(f32)).astype(f32)
    return np.stack([c, s, m_b, m_t, length], axis=-1).astype(f32)


def load_segment_tables(map_table) -> tuple[np.ndarray, np.ndarray]:
    """``(left, right)`` float32 ``[n_paths, S, 5]`` for a ``MapTable``: the shipped asset (produced by the reference's own torch
    ops, ``tests/golden/gen/gen_pseudo_tables.py``) when the map has one and it matches the table's paths, else ``segment_tables``."""
    n = map_table.n_paths
    S = max(int(map_table.n_left.max()), int(map_table.n_right.max())) - 1
    left = np.zeros((n, S, SEG_FIELDS), np.float32)
    right = np.zeros((n, S, SEG_FIELDS), np.float32)
    asset = os.path.join(_ASSETS, f"{map_table.name}.npz")
    if getattr(map_table, "from_asset", True) and os.path.exists(asset):
        z = np.load(asset)
        if z["left"].shape[0] == n and z["left"].shape[1] <= S and z["right"].shape[1] <= S:
            left[:, : z["left"].shape[1]] = z["left"]
            right[:, : z["right"].shape[1]] = z["right"]
            return left, right
    for p in range(n):
        nl, nr = int(map_table.n_left[p]), int(map_table.n_right[p])
        left[p, : nl - 1] = segment_tables(map_table.left[p, :nl])
        right[p, : nr - 1] = segment_tables(map_table.right[p, :nr])
    return left, right


def make_cbf_config(p, n_circles: int | None = None):
    """``sigmaenv_cbf_config_t`` from ``Parameters`` the way ``CBFQP.initialize_params`` / ``__init__`` derive the values
    (reference ``sigmarl/cbf_qp.py:326-433``, ``sigmarl/rectangle_approximation.py:45-70``)."""
    from . import capi

    A = capi.AGENTS
    c = capi.CbfConfig()
    C = int(p.n_circles_approximate_vehicle if n_circles is None else n_circles)
    if not 1 <= C <= capi.CBF_MAX_CIRCLES:
        raise ValueError(f"n_circles_approximate_vehicle must be 1..{capi.CBF_MAX_CIRCLES}")
    c.n_circles = C
    c.dt_taylor = float(2 * p.dt)  # r = 2, cbf_qp.py:370-371
    c.lambda_ttcbf = 0.5  # :404
    c.h_nom = float(p.h_nom)
    c.fd_step = 0.02  # :372-373
    c.safety_buffer = 0.0  # :403
    length, width = A["length"], A["width"]
    segment = length / C
    c.circle_radius = float(np.hypot(segment / 2, width / 2))  # rectangle_approximation.py:53-55
    step = length / C
    start = -length / 2 + step / 2
    for i in range(C):
        c.circle_x[i] = start + i * step  # :64-69
    c.l_r, c.l_wb = A["l_r"], A["l_wb"]
    c.min_speed, c.min_steering = A["min_speed"], A["min_steering"]
    return c


def split_cbf_margins(flat: np.ndarray, B: int, N: int, Cc: int):
    """(lane_left [B,N,C], lane_right [B,N,C], pair [B,N,N,C,C]) views of the flat margin record of ``sigmaenv_cbf_rewards``."""
    n1 = B * N * Cc
    return flat[:n1].reshape(B, N, Cc), flat[n1:2 * n1].reshape(B, N, Cc), flat[2 * n1:].reshape(B, N, N, Cc, Cc)
