"""An independent solver of the ORIGINAL centralized CBF-QP (tests only).

The reference builds this problem in cvxpy and hands it to OSQP (``sigmarl/cbf_qp.py:733-929`` / ``:1223-1231``): per env, with
controls u in R^{2N}, one slack per CBF / CLF row and one lambda in [0, 1] per CBF row (2416 variables at N = 16, 3 circles)

    min  |(u - u_nom) diag(w)|^2 + w_lane |s_lane|^2 + w_pair |s_pair|^2 + w_clf |s_clf|^2 + w_lam |lambda|^2
    s.t. lo <= u <= hi;   A u + b0 + h lambda + s >= 0,  s >= 0,  0 <= lambda <= 1   (CBF rows);   clf_e u + s_clf >= clf_v, s_clf >= 0

cvxpy / OSQP do not exist in the build container, so the reference's own solution cannot be produced.  ``solve_original`` solves the
problem in ITS ORIGINAL FORM -- every slack and lambda an explicit variable, every inequality an explicit row, in the standard form OSQP
takes (min 1/2 x'Px + q'x, l <= Ax <= u) -- with a textbook Mehrotra predictor-corrector interior-point method on a sparse LU.  The problem
is strictly convex in all 2416 variables, so the limit is THE solution; it is reached to a relative gap of 1e-11, far below OSQP's 1e-5.
Nothing is shared with the product's solver (closed-form elimination of slacks / lambdas + projected Newton on the 2N controls), which
is what makes agreement between the two meaningful.
(Tried first and dropped: a plain numpy restatement of OSQP's ADMM iteration -- with the 1e9 slack weights it does not reach 1e-5 within
20 000 iterations without OSQP's compiled refactorisations -- and a primal-dual active-set iteration, which cycles on this problem.)
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def build_original_qp(con, unom, lo, hi, w, ws_lane, ws_pair, wl, n_lane, clf_e, clf_v, wc=1.0, ws_rows=None, wl_rows=None):
    """(P, q, A, l, u, n_u) of the original problem for one env.  Variable order: u [n], s [m], lambda [m], s_clf [n].
    ws_rows / wl_rows: per-row slack / lambda weights (the grouped problems: cross-group rows carry their own)."""
    n, m = len(unom), len(con)
    nv = 2 * n + 2 * m
    ws = np.where(np.arange(m) < n_lane, ws_lane, ws_pair).astype(np.float64) if ws_rows is None else np.asarray(ws_rows, np.float64)
    wlv = np.full(m, float(wl)) if wl_rows is None else np.asarray(wl_rows, np.float64)
    pd = np.concatenate([2.0 * w ** 2, 2.0 * ws, 2.0 * wlv, np.full(n, 2.0 * wc)])
    P = sp.diags(pd).tocsc()
    q = np.zeros(nv)
    q[:n] = -2.0 * (w ** 2) * unom
    rows, cols, vals = [], [], []
    for r, row in enumerate(con):  # A u + h lambda + s >= -b0
        i, j = int(row[0]), int(row[1])
        for k, c in enumerate((2 * i, 2 * i + 1)):
            rows.append(r); cols.append(c); vals.append(row[2 + k])
        if j >= 0:
            for k, c in enumerate((2 * j, 2 * j + 1)):
                rows.append(r); cols.append(c); vals.append(row[4 + k])
        rows.append(r); cols.append(n + r); vals.append(1.0)
        rows.append(r); cols.append(n + m + r); vals.append(row[7])
    for k in range(n):  # clf_e u + s_clf >= clf_v
        rows.append(m + k); cols.append(k); vals.append(clf_e[k])
        rows.append(m + k); cols.append(n + 2 * m + k); vals.append(1.0)
    nr = m + n
    for k in range(nv):  # bounds as identity rows: u box, s >= 0, 0 <= lambda <= 1, s_clf >= 0
        rows.append(nr + k); cols.append(k); vals.append(1.0)
    l = np.concatenate([-con[:, 6], np.asarray(clf_v, np.float64), lo, np.zeros(m), np.zeros(m), np.zeros(n)])
    u = np.concatenate([np.full(m, np.inf), np.full(n, np.inf), hi, np.full(m, np.inf), np.ones(m), np.full(n, np.inf)])
    A = sp.csr_matrix((vals, (rows, cols)), shape=(nr + nv, nv))
    return P, q, A, l, u, n


def solve_original(P, q, A, l, u, max_iter=120, tol=1e-10):
    """Mehrotra predictor-corrector interior-point method on  min 1/2 x'Px + q'x, l <= Ax <= u  (every finite side of every row one
    inequality G x <= h).  Returns (x, info); info carries the final primal / dual residuals and the complementarity gap."""
    fl, fu = np.isfinite(l), np.isfinite(u)
    G = sp.vstack([-A[np.flatnonzero(fl)], A[np.flatnonzero(fu)]]).tocsr()
    h = np.concatenate([-l[fl], u[fu]])
    nv, ni = P.shape[0], G.shape[0]
    Pd = P.tocsc()
    GT = G.T.tocsr()
    # start: the unconstrained minimiser pulled strictly inside (slacks of the inequalities and multipliers at 1-ish of the problem's scale)
    x = -q / P.diagonal()
    sx = np.maximum(h - G @ x, 1.0)
    z = np.ones(ni)
    scale = max(1.0, np.abs(q).max())
    it = 0
    best = (np.inf, x, z, sx)
    for it in range(1, max_iter + 1):
        rd = Pd @ x + q + GT @ z
        rp = G @ x + sx - h
        mu = float(sx @ z) / ni
        # a weakly active row (slack and multiplier both ~0) is resolved to ~sqrt(mu) only: the gap is driven to 1e-13, the dual residual
        # relative to the size of its terms (the 1e9-weighted slack rows) -- beyond that point fp64 has nothing left and the step degrades
        dscale = max(scale, np.abs(Pd @ x).max(), np.abs(GT @ z).max())
        merit = max(np.abs(rp).max(), mu * 1e4, np.abs(rd).max() / dscale)
        if merit < best[0]:
            best = (merit, x.copy(), z.copy(), sx.copy())
        if (np.abs(rp).max() <= 1e-9 and mu <= 1e-13 and np.abs(rd).max() <= 1e-9 * dscale) or (mu < 1e-9 and merit > 1e3 * best[0]):
            break
        W = z / sx
        if not np.isfinite(W).all() or W.max() > 1e30:  # the barrier is numerically exhausted: the iterate is as converged as fp64 allows
            break
        K = (Pd + GT @ sp.diags(W) @ G).tocsc()
        try:
            lu = spla.splu(K, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
        except RuntimeError:
            break

        def step(rc):
            rhs = -rd - GT @ (W * rp - rc / sx)
            dx = lu.solve(rhs)
            dx = dx + lu.solve(rhs - K @ dx)
            dz = W * (G @ dx + rp) - rc / sx
            ds = -(rc + sx * dz) / z
            return dx, ds, dz

        def max_step(v, dv):
            neg = dv < 0
            return min(1.0, float(np.min(-v[neg] / dv[neg]))) if neg.any() else 1.0

        dx, ds, dz = step(sx * z)  # predictor (affine scaling)
        ap, ad = max_step(sx, ds), max_step(z, dz)
        mu_aff = float((sx + ap * ds) @ (z + ad * dz)) / ni
        sigma = (mu_aff / mu) ** 3
        dx, ds, dz = step(sx * z + ds * dz - sigma * mu)  # corrector
        ap, ad = 0.995 * max_step(sx, ds), 0.995 * max_step(z, dz)
        a_ = min(ap, ad)
        x, sx, z = x + a_ * dx, sx + a_ * ds, z + a_ * dz
    _, x, z, sx = best  # (the best iterate: past ~1e-13 the barrier system is numerically exhausted and the residuals grow again)
    Ax = A @ x
    info = dict(iterations=it, primal=float(max(0.0, np.max(l - Ax), np.max(Ax - u))), dual=float(np.abs(Pd @ x + q + GT @ z).max() / max(scale, np.abs(Pd @ x).max(), np.abs(GT @ z).max())),
                gap=float(sx @ z) / ni / scale)
    return x, info
