"""Independent (numpy) optimality check of a centralized CBF-QP solution -- test infrastructure shared by the CPU and GPU tests.

The reference hands the QP of ``sigmarl/cbf_qp.py:733-929`` to cvxpy / OSQP, which do not exist in the build container, so a solution
cannot be compared with the reference's.  Instead the KKT conditions of the ORIGINAL problem (all variables: u, s_lane, s_pair, s_clf,
lambda) are verified for the returned u: the slacks / lambdas / multipliers are recovered per constraint row, then primal and dual
feasibility, complementarity and stationarity are evaluated.  The problem is strictly convex in u, so a KKT point is THE minimiser.
"""
import numpy as np


def recover_row(g, h, ws, wl):
    """Optimal (lambda, s) of one CBF row for a given g = A u + b0: min ws s^2 + wl lambda^2, g + h lambda + s >= 0, 0 <= lambda <= 1, s >= 0."""
    if wl <= 0:
        lam = 1.0 if h > 0 else 0.0
        return lam, max(0.0, -(g + h * lam))
    if g >= 0:
        return 0.0, 0.0
    d = -g
    if h <= 0:
        return 0.0, d
    D = ws * h * h + wl
    if ws * h * d <= D:  # interior lambda; s = d - h lambda written without the cancellation
        return ws * h * d / D, d * wl / D
    return 1.0, d - h


def kkt_residuals(u, unom, con, lo, hi, w, ws_lane, ws_pair, wl, n_lane, clf_e=None, clf_v=None, wc=1.0):
    """u, unom: [n]; con: [n_con, 8] rows (i, j, a0..a3, b0, h).  Returns a dict of residuals (all should be ~0) and the objective."""
    n = len(u)
    grad = 2.0 * (w ** 2) * (u - unom)  # d/du |(u - unom) W|^2
    obj = float(np.sum((w * (u - unom)) ** 2))
    worst_feas = worst_comp = worst_lam_stat = 0.0
    for r, row in enumerate(con):
        i, j = int(row[0]), int(row[1])
        idx = [2 * i, 2 * i + 1] + ([2 * j, 2 * j + 1] if j >= 0 else [])
        a = row[2:2 + len(idx)]
        g = float(a @ u[idx] + row[6])
        h = float(row[7])
        ws = ws_lane if r < n_lane else ws_pair
        lam, s = recover_row(g, h, ws, wl)
        mu = 2.0 * ws * s  # multiplier of the row: stationarity in s (s > 0) / mu = 0 (s = 0)
        slack = g + h * lam + s
        worst_feas = max(worst_feas, max(0.0, -slack) / max(1.0, abs(g)))
        worst_comp = max(worst_comp, abs(mu * slack) / max(1.0, mu))
        # stationarity in lambda: 2 wl lam - mu h - nu_lo + nu_hi = 0 with nu >= 0 complementary to the bounds
        t = 2.0 * wl * lam - mu * h
        if 0.0 < lam < 1.0:
            worst_lam_stat = max(worst_lam_stat, abs(t) / max(1.0, abs(mu * h)))
        elif lam <= 0.0:
            worst_lam_stat = max(worst_lam_stat, max(0.0, -t) / max(1.0, abs(mu * h)))  # nu_lo = t >= 0
        else:
            worst_lam_stat = max(worst_lam_stat, max(0.0, t) / max(1.0, abs(mu * h)))   # nu_hi = -t >= 0
        grad[idx] -= mu * a
        obj += ws * s * s + wl * lam * lam
    if clf_e is not None:
        c = clf_v - clf_e * u
        s_clf = np.maximum(c, 0.0)
        grad -= 2.0 * wc * s_clf * clf_e
        obj += float(wc * np.sum(s_clf ** 2))
    # stationarity in u with the box multipliers: free coordinates need grad = 0, at a bound the sign must push outward
    scale = np.maximum(1.0, np.abs(2.0 * (w ** 2) * (u - unom)))
    res = np.where((u <= lo) & (grad > 0), 0.0, np.where((u >= hi) & (grad < 0), 0.0, np.abs(grad)))
    # relative to the size of the terms that cancel in the gradient
    mag = np.full(n, 1.0)
    for r, row in enumerate(con):
        i, j = int(row[0]), int(row[1])
        idx = [2 * i, 2 * i + 1] + ([2 * j, 2 * j + 1] if j >= 0 else [])
        a = row[2:2 + len(idx)]
        g = float(a @ u[idx] + row[6])
        ws = ws_lane if r < n_lane else ws_pair
        _, s = recover_row(g, float(row[7]), ws, wl)
        mag[idx] = np.maximum(mag[idx], np.abs(2.0 * ws * s * a))
    box = float(max(np.max(lo - u), np.max(u - hi), 0.0))
    return dict(stationarity=float(np.max(res / np.maximum(scale, mag))), feasibility=worst_feas, complementarity=worst_comp,
                lambda_stationarity=worst_lam_stat, box=box, objective=obj)


def objective(u, unom, con, w, ws_lane, ws_pair, wl, n_lane, clf_e=None, clf_v=None, wc=1.0):
    return kkt_residuals(u, unom, con, np.full(len(u), -np.inf), np.full(len(u), np.inf), w, ws_lane, ws_pair, wl, n_lane, clf_e, clf_v, wc)["objective"]
