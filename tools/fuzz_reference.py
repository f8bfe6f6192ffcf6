#!/usr/bin/env python
"""BUILD CONTAINER ONLY: fresh random rollouts of the imported reference (tests/golden/gen/refshim.py + gen_golden.run_traj) replayed through the
C oracle -- the check the committed goldens are instances of, on configurations they do not contain.  Every trajectory is generated in its own
interpreter (the generator's rule) into a scratch directory; nothing is committed.
Usage: python tools/fuzz_reference.py [--cases 12] [--seed 0] [--keep DIR]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEN = os.path.join(ROOT, "tests", "golden", "gen")

WORKER = r'''
import json, os, sys
sys.path.insert(0, {gen!r}); sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import gen_golden
gen_golden.OUT = {out!r}
kw = json.loads({kw!r})
gen_golden.run_traj(**kw)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--keep", default=None)
    ap.add_argument("--cbf", action="store_true", help="mix in rollouts with the CBF margin reward (tens of minutes each)")
    ap.add_argument("--full-obs", action="store_true", help="every case in bird view with the full observation (is_partial_observation=False; random switches and n_nearing)")
    ap.add_argument("--only", type=int, default=-1, help="run only this case of the sequence (the others just consume their random draws)")
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    import traj_replay as tr

    rng = np.random.default_rng(args.seed)
    out = args.keep or tempfile.mkdtemp(prefix="fuzz_ref_")
    os.makedirs(out, exist_ok=True)
    maps = [("cpm_entire", 12), ("cpm_entire", 12), ("intersection_1", 5), ("on_ramp_1", 5), ("roundabout_1", 5), ("interchange_2", 5), ("cpm_mixed", 4),
            ("intersection_5", 6), ("on_ramp_2_multilane", 6), ("interchange_1", 6), ("roundabout_2", 5), ("intersection_8", 6)]
    rews = ["distance", "ttc", "sparse", "distance_sparse", "ttc_sparse"]
    worst = {}
    for k in range(args.cases):
        scen, nmax = maps[rng.integers(len(maps))]
        N = int(rng.integers(2, nmax + 1))
        B = int(rng.integers(2, 4))
        kw = dict(name=f"fuzz{k}", T=int(rng.integers(12, 28)), B=B, seed=int(rng.integers(1000, 100000)), mode_pattern=[int(v) for v in rng.integers(0, 2, B)],
                  n_agents=N, scenario_type=scen, dt=float(rng.choice([0.05, 0.1])), is_use_mtv_distance=bool(rng.integers(2)), rew_method=str(rews[rng.integers(len(rews))]),
                  is_testing_mode=bool(rng.integers(5) == 0), is_apply_mask=bool(rng.integers(3) == 0), max_steps=int(rng.integers(8, 40)))
        if scen == "cpm_mixed":
            kw["cpm_scenario_probabilities"] = [1.0, 0.0, 0.0]
        if rng.integers(3) == 0:
            kw["reset_agent_fixed_duration"] = 0.5
        ns = int(rng.choice([3, 3, 2, 5]))  # a build constant of the oracle too: the variants the test-suite builds
        if ns != 3:
            kw["n_points_short_term"] = ns
        if rng.integers(5) == 0:
            kw["is_using_opponent_modeling"] = True
        if rng.integers(3) == 0:
            kw.update(is_obs_steering=bool(rng.integers(2)), is_observe_ref_path_other_agents=bool(rng.integers(2)), is_observe_vertices=bool(rng.integers(2)),
                      is_observe_distance_to_agents=bool(rng.integers(2)), is_observe_distance_to_center_line=bool(rng.integers(2)),
                      is_observe_distance_to_boundaries=bool(rng.integers(2)))
            if rng.integers(3) == 0:
                kw.update(is_ego_view=False, is_apply_mask=bool(rng.integers(2)))  # (with the mask: the lanelet-relation mask on OSM maps)
        if args.full_obs:  # round 4: the full observation (observation_provider_rt.py:756-851): bird view, all agents, n_nearing chunks -- only shapes the reference's reshape takes
            from sigmarl_amd import capi
            from sigmarl_amd.params import Parameters as _P, obs_flags as _of
            for _ in range(20):
                cand = dict(is_ego_view=False, is_partial_observation=False, n_nearing_agents_observed=int(rng.integers(1, 5)), is_apply_mask=bool(rng.integers(2)),
                            is_obs_steering=bool(rng.integers(2)), is_observe_ref_path_other_agents=bool(rng.integers(2)), is_observe_vertices=bool(rng.integers(2)),
                            is_observe_distance_to_agents=bool(rng.integers(2)), is_observe_distance_to_center_line=bool(rng.integers(2)),
                            is_observe_distance_to_boundaries=bool(rng.integers(2)))
                try:
                    capi.obs_dim(min(cand["n_nearing_agents_observed"], N - 1), _of(_P(n_agents=N, n_points_short_term=ns, **cand)), ns, N)
                    kw.update(cand)
                    break
                except ValueError:
                    continue
        if args.cbf and rng.integers(4) == 0:  # the CBF margin reward of the reference in front of every step (cbf_qp.py:2534-2804, one Python object per env: VERY slow)
            kw.update(hook="cbf", rew_method=str(rng.choice(["cbf", "cbf_sparse"])), is_using_cbf_training=True, is_solve_qp=False,
                      nom_controller_type=str(rng.choice(["rl", "clf"])), T=int(rng.integers(6, 12)), n_agents=min(N, 4))
            for key in ("is_obs_steering", "is_observe_ref_path_other_agents", "is_observe_vertices", "is_observe_distance_to_agents", "is_observe_distance_to_center_line",
                        "is_observe_distance_to_boundaries", "is_ego_view", "reset_agent_fixed_duration"):
                kw.pop(key, None)
            kw["is_testing_mode"] = False
        if args.only >= 0 and k != args.only:
            continue
        code = WORKER.format(gen=GEN, root=ROOT, out=out, kw=json.dumps(kw))
        try:
            r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONHASHSEED="0"), timeout=600)
        except subprocess.TimeoutExpired:  # (the reference's unbounded rejection loop can spin forever on a crowded map)
            print(f"case {k}: reference run did not finish in 600 s ({kw})")
            continue
        if r.returncode != 0:
            print(f"case {k}: reference run failed ({kw}):\n{r.stderr[-1500:]}")
            continue
        tr.GOLDEN_DIR = out
        z, meta = tr.load_fixture(f"fuzz{k}")
        cfg, mp = tr.config_from_meta(meta)
        env = ob.OracleEnv(cfg, mp)
        rep = tr.replay(env, z, meta, mp)
        env.close()
        ok = rep.total_mismatch() == 0 and all(v <= 1e-5 for v in rep.max_abs.values()) and rep.cbf_ok()
        for key, v in rep.max_abs.items():
            worst[key] = max(worst.get(key, 0.0), float(v))
        print(f"case {k}: {scen} N={N} B={B} T={kw['T']} {'mtv' if kw['is_use_mtv_distance'] else 'c2c'} {kw['rew_method']} "
              f"{'testing ' if kw['is_testing_mode'] else ''}{'cbf-' + kw['nom_controller_type'] + ' ' if kw.get('hook') else ''}mismatches {rep.total_mismatch()} max err {max(rep.max_abs.values()):.2e} {'ok' if ok else 'FAIL ' + str(rep)}", flush=True)
    print("worst fp32 error per buffer:", {k: f"{v:.2e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})


if __name__ == "__main__":
    main()
