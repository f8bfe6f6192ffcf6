#!/usr/bin/env python
"""Histogram fixture of the REFERENCE's reset sampler (build container only; imports /root/reference through refshim).

world_state_rt_sim.py:215-311 draws, per agent, a path (uniform over the scenario's paths), a centre-line point (uniform on [3, n/2) in
training; from the path's beginning with a range that grows with the tries in testing mode) and rejects starts closer than
reset_agent_min_distance to the agents placed before; the speed is uniform on [0, max_speed).  The device-side sampler follows the same
rule from a counter-based generator: parity is distributional.  This script records the reference's marginals on the CPM map with 16 agents:
  path_counts[40], point_frac_counts[20] (position of the point inside its allowed range), speed_counts[20], first-agent vs last-agent
  point histograms (the rejection makes later agents' draws conditional), min_spacing (smallest centre distance seen), n_env_resets
for training mode and for testing mode.  Deterministic: torch.manual_seed(20260928).  One command: python tests/golden/gen/gen_reset_distribution.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
from sigmarl.helper_common import Parameters  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "reset_distribution.npz")


def collect(testing, n_rounds, B=32, N=16):
    torch.manual_seed(20260928 + int(testing))
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=B, is_testing_mode=testing,
                   is_challenging_initial_state_buffer=False)
    env = refshim.RefEnv(p, B)
    sc = env.scenario
    paths = sc.map.parser.reference_paths
    n_pts = np.asarray([pp["center_line"].shape[0] for pp in paths])
    rp = sc.world_state.ref_paths_agent_related
    path_c = np.zeros(len(paths), np.int64)
    frac_c, speed_c = np.zeros(20, np.int64), np.zeros(20, np.int64)
    first_pt, last_pt = np.zeros(20, np.int64), np.zeros(20, np.int64)
    point_raw = np.zeros(64, np.int64)
    min_sp = np.inf
    n = 0
    for r in range(n_rounds):
        for e in range(B):
            sc.env_reset_world_at(e)
        pid = rp.path_id.numpy().astype(np.int64)
        pt = rp.point_id.numpy().astype(np.int64)
        pos = torch.stack([a.state.pos for a in env.world.agents], dim=1).numpy()
        spd = torch.stack([a.state.speed for a in env.world.agents], dim=1).squeeze(-1).numpy()
        half = n_pts[pid] // 2
        frac = (pt - 3) / np.maximum(1, half - 3)
        np.add.at(path_c, pid.ravel(), 1)
        np.add.at(frac_c, np.minimum(19, (frac.ravel() * 20).astype(np.int64)), 1)
        np.add.at(speed_c, np.minimum(19, (spd.ravel() / 1.0 * 20).astype(np.int64)), 1)
        np.add.at(first_pt, np.minimum(19, (frac[:, 0] * 20).astype(np.int64)), 1)
        np.add.at(last_pt, np.minimum(19, (frac[:, -1] * 20).astype(np.int64)), 1)
        np.add.at(point_raw, np.minimum(63, pt.ravel()), 1)
        d = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1)) + np.eye(N)[None] * 1e9
        min_sp = min(min_sp, float(d.min()))
        n += B
    return dict(path_counts=path_c, point_frac_counts=frac_c, speed_counts=speed_c, first_agent_point=first_pt, last_agent_point=last_pt,
                point_raw_counts=point_raw, min_spacing=np.float64(min_sp), n_env_resets=np.int64(n),
                min_distance=np.float64(float(sc.constants.reset_agent_min_distance)))


MIXED_PROBS = [0.5, 0.3, 0.2]


def collect_mixed(n_rounds, B=32, N=2):
    """cpm_mixed (world_state_rt_sim.py:313-358): every env reset draws its sub-scenario from cpm_scenario_probabilities, then the agents' paths from
    that sub-scenario's list.  Histograms: sub-scenario, (sub-scenario, list-local path) pairs, point position, speed.  Two agents: with four the reference's
    unbounded rejection loop practically never terminates on the merge lists (4 short paths, minimum spacing 0.27 m)."""
    import signal

    class _Stuck(Exception):
        pass

    def _alarm(signum, frame):
        raise _Stuck()

    # The reference's rejection loop has no bound (world_state_rt_sim.py:215-311): on the merge lists (4 short, converging paths) a first agent placed at the
    # merge point leaves the second one NO feasible start and the loop never ends.  A reset that does not finish within 0.5 s (normal: 5 ms) is abandoned
    # and left out of the histograms; the device sampler's bounded loop falls back to an overlapping start in that case, which the check excludes as well.
    signal.signal(signal.SIGALRM, _alarm)
    p = Parameters(n_agents=N, scenario_type="cpm_mixed", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=B, is_testing_mode=False,
                   is_challenging_initial_state_buffer=False, cpm_scenario_probabilities=MIXED_PROBS)
    env = None
    for attempt in range(50):  # (the constructor resets every env once: retry with the next seed if one of those resets is stuck)
        torch.manual_seed(20260929 + attempt)
        signal.setitimer(signal.ITIMER_REAL, 30.0)
        try:
            env = refshim.RefEnv(p, B)
            signal.setitimer(signal.ITIMER_REAL, 0.0)
            break
        except _Stuck:
            continue
    assert env is not None
    sc = env.scenario
    rp = sc.world_state.ref_paths_agent_related
    mr = sc.world_state.ref_paths_map_related
    lists = [mr.long_term_intersection, mr.long_term_merge_in, mr.long_term_merge_out]
    counts = [len(l) for l in lists]
    offs = np.concatenate([[0], np.cumsum(counts)])
    scen_c = np.zeros(3, np.int64)
    pair_c = np.zeros(int(offs[-1]), np.int64)
    frac_c, speed_c = np.zeros(20, np.int64), np.zeros(20, np.int64)
    n = n_stuck = 0
    for r in range(n_rounds):
        for e in range(B):
            signal.setitimer(signal.ITIMER_REAL, 0.5)
            try:
                sc.env_reset_world_at(e)
                signal.setitimer(signal.ITIMER_REAL, 0.0)
            except _Stuck:
                n_stuck += 1
                continue
            sid = rp.scenario_id[e].numpy().astype(np.int64)
            pid = rp.path_id[e].numpy().astype(np.int64)
            pt = rp.point_id[e].numpy().astype(np.int64)
            assert (sid == sid[0]).all() and 1 <= sid[0] <= 3
            spd = torch.stack([a.state.speed[e] for a in env.world.agents]).reshape(-1).numpy()
            n_pts = np.asarray([lists[sid[i] - 1][pid[i]]["center_line"].shape[0] for i in range(N)])
            half = n_pts // 2
            frac = (pt - 3) / np.maximum(1, half - 3)
            scen_c[sid[0] - 1] += 1
            np.add.at(pair_c, offs[sid - 1] + pid, 1)
            np.add.at(frac_c, np.minimum(19, (frac * 20).astype(np.int64)), 1)
            np.add.at(speed_c, np.minimum(19, (spd / 1.0 * 20).astype(np.int64)), 1)
            n += 1
    return dict(scenario_counts=scen_c, pair_counts=pair_c, list_counts=np.asarray(counts, np.int64), point_frac_counts=frac_c, speed_counts=speed_c,
                n_env_resets=np.int64(n), n_stuck=np.int64(n_stuck), n_agents=np.int64(N), probabilities=np.asarray(MIXED_PROBS, np.float64),
                min_distance=np.float64(float(sc.constants.reset_agent_min_distance)))


def main():
    if "--mixed-only" in sys.argv:  # add the cpm_mixed histograms to the committed fixture without redrawing the others
        out = dict(np.load(OUT))
        for k, v in collect_mixed(128).items():
            out[f"mixed_{k}"] = v
        print("mixed", {k: (v if np.ndim(v) == 0 else v.sum()) for k, v in out.items() if k.startswith("mixed")})
        np.savez_compressed(OUT, **out)
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
        return
    out = {}
    for k, v in collect_mixed(128).items():
        out[f"mixed_{k}"] = v
    for tag, testing, rounds in (("train", False, 48), ("test", True, 16)):
        for k, v in collect(testing, rounds).items():
            out[f"{tag}_{k}"] = v
        print(tag, {k: (v if np.ndim(v) == 0 else v.sum()) for k, v in out.items() if k.startswith(tag)})
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
