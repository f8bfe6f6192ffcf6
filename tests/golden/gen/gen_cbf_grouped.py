#!/usr/bin/env python
"""Golden data of the GROUPED CBF-QPs (sigmarl/cbf_qp.py:193-310 `group_agents_k_nearest`, :1562-1856
`build_grouped_cbf_qps`, :1858-2281 `update_grouped_cbf_qps`, :2491-2532 `ttcbf_pair_affine_coeffs_cross`) -> tests/golden/cbf_grouped.npz.

BUILD CONTAINER ONLY.  cvxpy / OSQP are absent, so nothing is solved; the reference's own `CBFQP(is_grouping_agents=True)` is constructed
and `update_qp` run on the set states of tests/golden/cbf_functions.npz under a stand-in for cvxpy that only STORES what the reference
assigns (`Parameter.value`): the fixture holds the groups the reference forms, the neighbour lists of its cross-group rows and every
coefficient it hands to the group problems (intra-group pair rows, cross rows), i.e. the complete data of the problems it would solve.
Run through tests/golden/gen/gen_golden.py (`python tests/golden/gen/gen_golden.py cbf_grouped`), which also refreshes the manifest.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)


def _mk_cvxpy():
    """Third-party stand-in: expressions are inert, Variables / Parameters keep `.value` and `.shape`, Problem.solve does nothing."""
    m = types.ModuleType("cvxpy")

    class Expr:
        def _e(self, *a, **k):
            return Expr()
        __getitem__ = __matmul__ = __rmatmul__ = __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __neg__ = _e
        __le__ = __ge__ = __lt__ = __gt__ = __truediv__ = _e
        __hash__ = object.__hash__

        def __eq__(self, o):
            return Expr()

    class Leaf(Expr):
        def __init__(self, shape=(), **kw):
            self.shape = (shape,) if isinstance(shape, int) else tuple(shape)
            self.value = None

    class Problem:
        def __init__(self, obj, cons):
            self.status = "not_solved"
            self.solver_stats = types.SimpleNamespace()

        def solve(self, **kw):
            return None

    m.Variable, m.Parameter, m.Problem = Leaf, Leaf, Problem
    m.Minimize = lambda e: Expr()
    m.sum_squares = lambda e: Expr()
    m.SolverError = type("SolverError", (Exception,), {})
    m.OSQP, m.CLARABEL, m.SCS = "OSQP", "CLARABEL", "SCS"
    m.OPTIMAL, m.OPTIMAL_INACCURATE = "optimal", "optimal_inaccurate"
    return m


def main():
    sys.modules["cvxpy"] = _mk_cvxpy()  # before refshim: its finder only answers for modules that are not there yet
    sys.path.insert(0, HERE)
    import refshim
    refshim.install()
    from sigmarl.cbf_qp import CBFQP
    from sigmarl.helper_common import Parameters

    z = np.load(os.path.join(OUT, "cbf_functions.npz"))
    state_all, path_all, act_all, ref_all = z["p2_state"], z["p2_path"], z["p2_act"], z["g10_ref"]
    N, C = state_all.shape[1], 3
    envs = [0, 1, 2, 3, 5, 8, 13, 21, 30, 47]
    cases = [(2, 0.5, "rl"), (3, 0.5, "rl"), (4, 1.0, "rl"), (2, 1.0, "clf"), (5, 0.5, "clf")]  # max_group_size, observation_range, nominal controller
    S = len(envs) * len(cases)
    grp = np.zeros((S, N), np.int32)
    pair6 = np.full((S, N, N, C, C, 6), np.nan)
    cross4 = np.full((S, N, N, C, C, 4), np.nan)   # [local i, external j]: A_i (2), b0, h as handed to the cvxpy parameters
    nbr = np.zeros((S, N, N), np.int8)             # 1: j is in cross_groups[i]
    nom = np.zeros((S, N, 2), np.float32)          # world_state.nominal_action_{vel,steer} after the update
    unom = np.zeros((S, N, 2))                     # U_nom handed to the group problems
    meta = []
    k = 0
    for (m, rng, ctrl) in cases:
        p = Parameters(n_agents=N, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=len(envs), dt=0.05,
                       is_using_cbf_training=True, is_solve_qp=True, rew_method="cbf", is_challenging_initial_state_buffer=False,
                       is_grouping_agents=True, max_group_size=m, observation_range=rng, nom_controller_type=ctrl, adaptive_lambda=True)
        env = refshim.RefEnv(p, len(envs))
        sc = env.scenario
        ag = env.world.agents
        st = torch.from_numpy(state_all[envs])
        for i, a in enumerate(ag):
            a.state.pos = st[:, i, 0:2].clone(); a.state.rot = st[:, i, 2:3].clone()
            a.state.speed = st[:, i, 3:4].clone(); a.state.steering = st[:, i, 4:5].clone()
        ref = torch.zeros(len(envs), N, 6)
        ref[:, :, 4:6] = torch.from_numpy(ref_all[envs])
        fake = types.SimpleNamespace(base_env=types.SimpleNamespace(scenario_name=sc))
        for e in range(len(envs)):
            td = {("agents", "info", "path_id"): torch.from_numpy(path_all[envs]).clone(), ("agents", "action"): torch.from_numpy(act_all[envs]).clone(),
                  ("agents", "info", "ref"): ref.clone()}
            c = CBFQP(env=fake, env_idx=e)
            c.update_qp(td)
            groups = sc.inter_groups
            for g, mem in enumerate(groups):
                cache = c._group_qp_caches[g]
                for a_ in mem:
                    grp[k, a_] = g
                for il, i in enumerate(mem):
                    unom[k, i] = cache["U_nom"].value[il]
                    for jl, j in enumerate(mem):
                        if jl <= il:
                            continue
                        for ci in range(C):
                            for cj in range(C):
                                key = (il, jl, ci, cj)
                                pair6[k, i, j, ci, cj] = np.concatenate([cache["Aij_i"][key].value.reshape(2), cache["Aij_j"][key].value.reshape(2),
                                                                         cache["b0ij"][key].value.reshape(1), cache["hij"][key].value.reshape(1)])
                    for ei, j in enumerate(sc.cross_groups.get(i, [])):
                        nbr[k, i, j] = 1
                        for ci in range(C):
                            for cj in range(C):
                                key = (il, ei, ci, cj)
                                cross4[k, i, j, ci, cj] = np.concatenate([cache["Axc_i"][key].value.reshape(2), cache["b0xc"][key].value.reshape(1),
                                                                          cache["hxc"][key].value.reshape(1)])
            nom[k, :, 0] = sc.world_state.nominal_action_vel[e].numpy()
            nom[k, :, 1] = sc.world_state.nominal_action_steer[e].numpy()
            meta.append(dict(env=int(envs[e]), max_group_size=m, observation_range=rng, nominal=ctrl, rs=float(p.rs)))
            k += 1
    out = dict(grp=grp, pair6=pair6, cross4=cross4, nbr=nbr, nom=nom, unom=unom, meta_json=np.asarray(json.dumps(meta)))
    path = os.path.join(OUT, "cbf_grouped.npz")
    np.savez_compressed(path, **out)
    print("cbf_grouped", f"{os.path.getsize(path) / 1e6:.2f} MB", "samples", S, "intra pair rows", int(np.isfinite(pair6[..., 0]).sum()),
          "cross rows", int(np.isfinite(cross4[..., 0]).sum()), "neighbour links", int(nbr.sum()))


if __name__ == "__main__":
    main()
