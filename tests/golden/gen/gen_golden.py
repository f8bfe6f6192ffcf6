"""Generate golden vectors by RUNNING the reference (this container only).

Usage:  python tests/golden/gen/gen_golden.py [name ...]

Writes ``tests/golden/*.npz``.  Every array is produced by code under
``/root/reference`` imported through ``refshim`` (third-party stand-ins only).
The fixtures are data: inputs (initial states, actions, recorded reset draws)
and expected outputs (every hot-path tensor after each step).

Trajectory fixture layout (``traj_*.npz``), T steps, B envs, N agents:
  meta_json                      parameters used
  init_*                         state after the initial reset (+ derived tensors)
  act[T,B,N,2]                   raw actions fed to the env
  post_*[T,...]                  tensors after reward/observation/info, BEFORE done()
  done[T,B]
  next_*[T,...]                  tensors after done() and the resets of step t
  ev_*                           flat list of reset events (step, kind, env, agent)
                                 with the post-reset state of every agent of that env
"""
from __future__ import annotations

import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()

from sigmarl.helper_common import Parameters  # noqa: E402

OUT = os.path.abspath(os.path.join(HERE, ".."))

REWARD_FIELDS = [
    "rew_progress", "rew_reach_goal", "rew_speed", "rew_centerline", "rew_near_other_agents",
    "rew_near_left_lane", "rew_near_right_lane", "rew_collide_other_agents", "rew_collide_lane",
    "rew_energy_acceleration", "rew_energy_steering", "rew_total",
]


def np_(t):
    return t.detach().cpu().numpy().copy()


def snapshot(env, prefix, with_obs=None, rew=None):
    sc = env.scenario
    ws = sc.world_state
    ag = env.world.agents
    d = {}
    d[prefix + "pos"] = np_(torch.stack([a.state.pos for a in ag], dim=1))
    d[prefix + "rot"] = np_(torch.stack([a.state.rot for a in ag], dim=1).squeeze(-1))
    d[prefix + "vel"] = np_(torch.stack([a.state.vel for a in ag], dim=1))
    d[prefix + "speed"] = np_(torch.stack([a.state.speed for a in ag], dim=1).squeeze(-1))
    d[prefix + "steering"] = np_(torch.stack([a.state.steering for a in ag], dim=1).squeeze(-1))
    d[prefix + "sideslip"] = np_(torch.stack([a.state.sideslip_angle for a in ag], dim=1).squeeze(-1))
    d[prefix + "vertices"] = np_(ws.vertices)
    d[prefix + "dist_agents"] = np_(ws.distances.agents)
    d[prefix + "dist_ref"] = np_(ws.distances.ref_paths)
    d[prefix + "dist_left"] = np_(ws.distances.left_boundaries)
    d[prefix + "dist_right"] = np_(ws.distances.right_boundaries)
    d[prefix + "dist_bound"] = np_(ws.distances.boundaries)
    d[prefix + "cp_ref"] = np_(ws.distances.closest_point_on_ref_path)
    d[prefix + "cp_left"] = np_(ws.distances.closest_point_on_left_b)
    d[prefix + "cp_right"] = np_(ws.distances.closest_point_on_right_b)
    d[prefix + "short_term"] = np_(ws.ref_paths_agent_related.short_term)
    d[prefix + "col_agents"] = np_(ws.collisions.with_agents)
    d[prefix + "col_lane"] = np_(ws.collisions.with_lanelets)
    d[prefix + "col_entry"] = np_(ws.collisions.with_entry_segments)
    d[prefix + "col_exit"] = np_(ws.collisions.with_exit_segments)
    d[prefix + "scenario_id"] = np_(ws.ref_paths_agent_related.scenario_id)
    d[prefix + "path_id"] = np_(ws.ref_paths_agent_related.path_id)
    d[prefix + "point_id"] = np_(ws.ref_paths_agent_related.point_id)
    d[prefix + "timer_step"] = np_(sc.timer.step)
    d[prefix + "prev_pos"] = np_(sc.state_buffer.get_latest(n=1)[:, :, 0:2])
    if with_obs is not None:
        d[prefix + "obs"] = np_(torch.stack(with_obs, dim=1))
        d[prefix + "nearing_idx"] = np_(sc.observation_provider.observations.nearing_agents_indices)
        cur = sc.observation_provider.map.current_lanelet_idx  # bird view + is_apply_mask only (map_manager.py:41-92): the agents' lanelets
        if isinstance(cur, torch.Tensor) and cur.numel():
            d[prefix + "current_lanelet"] = np_(cur.reshape(cur.shape[0], -1)).astype(np.int32)
    if rew is not None:
        d[prefix + "reward"] = np_(torch.stack(rew, dim=1))
        for f in REWARD_FIELDS:
            d[prefix + f] = np_(getattr(sc.reward_info, f))
        d[prefix + "num_task_tries"] = np_(sc.num_task_tries)
        d[prefix + "task_success_times"] = np_(sc.task_success_times)
    return d


def follow_policy(obs, gen, B, N, mode):
    """obs: list of N tensors [B,32] (default obs layout, ego view).  mode[b]: 0 random, 1 follow."""
    o = torch.stack(obs, dim=1)  # [B,N,32]
    pt = o[:, :, 3:5]  # 2nd short-term point (ego frame, normalised)
    ang = torch.atan2(pt[..., 1], pt[..., 0])
    steer_f = (1.6 * ang).clamp(-0.5, 0.5) + (torch.rand(B, N, generator=gen) - 0.5) * 0.06
    v_f = 0.55 + 0.45 * torch.rand(B, N, generator=gen)
    v_r = torch.rand(B, N, generator=gen) * 1.3 - 0.1  # exercises the action clamp
    steer_r = torch.rand(B, N, generator=gen) * 1.3 - 0.65  # exercises the steering clamp
    m = mode.view(B, 1).expand(B, N)
    v = torch.where(m == 1, v_f, v_r)
    s = torch.where(m == 1, steer_f, steer_r)
    return torch.stack([v, s], dim=-1).to(torch.float32)


class CbfHook:
    """Runs the reference's QP-free CBF margin path before every env step, as ``cbf_constrained_centralized_policy`` does after
    the policy (helper_training.py:1620-1627 -> cbf_qp.py:2534-2560), and records inputs, margins and the three reward channels."""

    def __init__(self, env, B, N):
        import types
        from sigmarl.cbf_qp import CBFQP

        self.env, self.B, self.N = env, B, N
        fake = types.SimpleNamespace(base_env=types.SimpleNamespace(scenario_name=env.scenario))
        self.ctl = [CBFQP(env=fake, env_idx=e) for e in range(B)]
        self.C = int(env.scenario.parameters.n_circles_approximate_vehicle)

    def meta(self):
        c = self.ctl[0]
        return dict(cbf_circle_radius=float(c.circle_radius), cbf_centers=[float(x) for x, _ in c.rec_cir_approx.centers],
                    cbf_dt_taylor=float(c.dt_taylor), cbf_lambda=float(c.lambda_ttcbf), cbf_h_nom=float(c.parameters.h_nom),
                    cbf_fd_step=float(c.dx), cbf_n_circles=self.C, numpy_version=np.__version__, torch_version=torch.__version__)

    def __call__(self, act):
        env, B, N, C = self.env, self.B, self.N, self.C
        sc = env.scenario
        ag = env.world.agents
        td = {("agents", "info", "path_id"): sc.world_state.ref_paths_agent_related.path_id.clone(), ("agents", "action"): act.clone(),
              ("agents", "info", "ref"): sc.world_state.ref_paths_agent_related.short_term.reshape(B, N, -1).clone()}
        rec = {
            "cbf_in_state": np_(torch.stack([torch.cat([a.state.pos, a.state.rot, a.state.speed, a.state.steering], dim=-1) for a in ag], dim=1)),
            "cbf_in_path": np_(td[("agents", "info", "path_id")]).astype(np.int32),
        }
        L = np.zeros((B, N, C)); R = np.zeros((B, N, C)); P = np.full((B, N, N, C, C), np.nan)
        # the float32 covering-circle centres exactly as the reference computes them (get_circle_centers, cbf_qp.py:527-573, on the state vector
        # compute_nominal_cbf_constraint_margins builds, :2586-2600): torch's cos / sin are within 1 ulp of the correctly rounded ones the oracle and the HIP
        # path use, and that ulp can flip an fp16 rounding of the pseudo distance downstream -- the parity tests can inject these instead
        cen = np.zeros((B, N, C, 2), np.float32)
        for e in range(B):
            for i in range(N):
                st_i = torch.cat([ag[i].state.pos[e], ag[i].state.rot[e], ag[i].state.speed[e], ag[i].state.steering[e]], dim=-1)
                cen[e, i] = np_(self.ctl[e].get_circle_centers(st_i)[:, 0:2])
        rec["cbf_centers"] = cen
        for e in range(B):
            c = self.ctl[e]
            c.time_pseudo_dis = 0
            m = c.compute_nominal_cbf_constraint_margins(td)
            L[e], R[e] = m["lane_L_margin"], m["lane_R_margin"]
            for (i, j, ci, cj), g in m["pair_margin"].items():
                P[e, i, j, ci, cj] = g
            c.update_qp(td)  # writes reward_info.rew_near_{left_lane,right_lane,other_agents}[e]
        rec["cbf_lane_left"], rec["cbf_lane_right"], rec["cbf_pair"] = L, R, P
        ri = sc.reward_info
        rec["cbf_rew"] = np.stack([np_(ri.rew_near_left_lane), np_(ri.rew_near_right_lane), np_(ri.rew_near_other_agents)], axis=0)
        return rec


def injected_start(scenario_type, n_agents, first_point=3, stride=3):
    """Deterministic initial state for more agents than the reference's rejection sampler can place (BASELINE
    config 4): agent i on path i mod n_paths, centre-line point first_point + stride * (i // n_paths), yaw of the centre line there.
    Returned in the form Parameters.predefined_ref_path_idx / Parameters.init_state take (world_state_rt_sim.py:99-126)."""
    probe = refshim.RefEnv(Parameters(n_agents=2, scenario_type=scenario_type, is_obs_noise=False, num_vmas_envs=1,
                                      is_challenging_initial_state_buffer=False), 1)
    paths = probe.scenario.map.parser.reference_paths
    jit = np.random.default_rng(4).uniform(-1.0, 1.0, (n_agents, 3))  # a few millimetres / milliradians: no two distances tie exactly
    idx, st = [], []                                                   # (torch.topk's order among exactly equal distances is unspecified)
    for i in range(n_agents):
        pid = i % len(paths)
        cl, yaw = paths[pid]["center_line"], paths[pid]["center_line_yaw"].reshape(-1)
        k = min(first_point + stride * (i // len(paths)), cl.shape[0] - 8)
        idx.append(pid)
        st.append([float(cl[k, 0]) + 4e-3 * jit[i, 0], float(cl[k, 1]) + 4e-3 * jit[i, 1], float(yaw[min(k, yaw.shape[0] - 1)]) + 2e-2 * jit[i, 2]])
    return idx, st


def run_traj(name, T, B, seed, mode_pattern, no_reset=False, hook=None, inject=None, **pkw):
    torch.manual_seed(seed)
    np.random.seed(seed)
    import random as _random
    _random.seed(seed)
    gen = torch.Generator().manual_seed(seed + 1000)
    kw = dict(
        is_obs_noise=False, is_apply_mask=False, max_steps=128, num_vmas_envs=B,
        is_challenging_initial_state_buffer=False, is_testing_mode=False,
    )
    kw.update(pkw)
    if inject is not None:
        kw["predefined_ref_path_idx"], kw["init_state"] = injected_start(kw["scenario_type"], kw["n_agents"], **inject)
        torch.manual_seed(seed)
    p = Parameters(**kw)
    env = refshim.RefEnv(p, B)
    sc = env.scenario
    N = env.n_agents
    mode = torch.tensor([mode_pattern[b % len(mode_pattern)] for b in range(B)])

    events = []  # (step, kind, env, agent) + per-env state arrays
    cur_step = [-1]
    orig_reset = sc.reset_world_at

    def logged_reset(env_index=None, agent_index=None):
        orig_reset(env_index=env_index, agent_index=agent_index)
        if env_index is None:
            return
        e = int(env_index)
        ag = env.world.agents
        rp = sc.world_state.ref_paths_agent_related
        events.append(dict(
            step=cur_step[0], kind=0 if agent_index is not None else 1, env=e,
            agent=int(agent_index) if agent_index is not None else -1,
            pos=np_(torch.stack([a.state.pos[e] for a in ag])),
            rot=np_(torch.stack([a.state.rot[e] for a in ag]).squeeze(-1)),
            vel=np_(torch.stack([a.state.vel[e] for a in ag])),
            speed=np_(torch.stack([a.state.speed[e] for a in ag]).squeeze(-1)),
            steering=np_(torch.stack([a.state.steering[e] for a in ag]).squeeze(-1)),
            sideslip=np_(torch.stack([a.state.sideslip_angle[e] for a in ag]).squeeze(-1)),
            scenario_id=np_(rp.scenario_id[e]), path_id=np_(rp.path_id[e]), point_id=np_(rp.point_id[e]),
        ))

    sc.reset_world_at = logged_reset

    obs = env.observe()
    out = {}
    out.update(snapshot(env, "init_", with_obs=obs))
    hook_obj = CbfHook(env, B, N) if hook == "cbf" else None
    steps = []
    for t in range(T):
        cur_step[0] = t
        act = follow_policy(obs, gen, B, N, mode)
        hook_rec = hook_obj(act) if hook_obj is not None else {}
        env.set_actions(act)
        env.world.step()
        rew = [sc.reward(a).clone() for a in env.world.agents]
        obs = [sc.observation(a).clone() for a in env.world.agents]
        info = [refshim.TorchUtils.recursive_clone(sc.info(a)) for a in env.world.agents]
        rec = {"act": np_(act)}
        rec.update(hook_rec)
        rec.update(snapshot(env, "post_", with_obs=obs, rew=rew))
        rec["post_act_clamped"] = np_(torch.stack([a.action.u for a in env.world.agents], dim=1))
        for key in ["rot", "distance_left_b", "distance_right_b", "is_collision_with_agents", "rew_total", "rew_near_other_agents", "rew_collide_lane"]:
            rec["post_info_" + key] = np_(torch.stack([info[i][key].reshape(B, -1)[:, 0] if info[i][key].ndim > 1 else info[i][key] for i in range(N)], dim=1))
        done = sc.done().clone()
        rec["done"] = np_(done)
        if not no_reset:
            for e in torch.where(done)[0].tolist():
                env.reset_env(e)
        if done.any() or any(ev["step"] == t for ev in events):
            obs = env.observe()
        rec.update(snapshot(env, "next_", with_obs=obs))
        steps.append(rec)
    for k in steps[0]:
        out[k] = np.stack([s[k] for s in steps], axis=0)
    ne = len(events)
    out["ev_step"] = np.asarray([e["step"] for e in events], np.int32).reshape(ne)
    out["ev_kind"] = np.asarray([e["kind"] for e in events], np.int32).reshape(ne)
    out["ev_env"] = np.asarray([e["env"] for e in events], np.int32).reshape(ne)
    out["ev_agent"] = np.asarray([e["agent"] for e in events], np.int32).reshape(ne)
    for k in ["pos", "rot", "vel", "speed", "steering", "sideslip", "scenario_id", "path_id", "point_id"]:
        shp = events[0][k].shape if ne else ((N, 2) if k in ("pos", "vel") else (N,))
        out["ev_" + k] = np.stack([e[k] for e in events], axis=0) if ne else np.zeros((0,) + shp, np.float32)
    meta = dict(kw)
    meta.update(dict(
        n_agents=N, T=T, B=B, seed=seed, dt=p.dt, no_reset=bool(no_reset), rew_method=p.rew_method, scenario_type=p.scenario_type,
        is_use_mtv_distance=p.is_use_mtv_distance,
        thresholds=dict(
            near_boundary_low=float(sc.thresholds.near_boundary_low), near_boundary_high=float(sc.thresholds.near_boundary_high),
            near_other_agents_low=float(sc.thresholds.near_other_agents_low), near_other_agents_high=float(sc.thresholds.near_other_agents_high),
            ttc_low=float(sc.thresholds.ttc_low), ttc_high=float(sc.thresholds.ttc_high),
        ),
        penalties=dict(
            near_boundary=float(sc.penalties.near_boundary), near_other_agents=float(sc.penalties.near_other_agents),
            collide_with_agents=float(sc.penalties.collide_with_agents), collide_with_boundaries=float(sc.penalties.collide_with_boundaries),
        ),
        rewards=dict(progress=float(sc.rewards.progress), reach_goal=float(sc.rewards.reach_goal)),
        n_events=ne, n_done=int(out["done"].sum()),
        n_col_agents=int(out["post_col_agents"].any(-1).sum()), n_col_lane=int(out["post_col_lane"].sum()),
        n_exit=int(out["post_col_exit"].sum()), n_entry=int(out["post_col_entry"].sum()),
    ))
    if hook_obj is not None:
        meta.update(hook_obj.meta())
    out["meta_json"] = np.asarray(json.dumps(meta))
    path = os.path.join(OUT, f"traj_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: meta[k] for k in ["n_events", "n_done", "n_col_agents", "n_col_lane", "n_exit", "n_entry"]},
          f"{os.path.getsize(path)/1e6:.2f} MB")


# --------------------------------------------------------------------------- #
# function-level goldens (direct calls into sigmarl/helper_scenario.py etc.)
# --------------------------------------------------------------------------- #
def gen_functions():
    from sigmarl.constants import AGENTS
    from sigmarl.dynamics import KinematicBicycleModel
    from sigmarl.helper_scenario import (
        angle_eliminate_two_pi, get_distances_between_agents, get_perpendicular_distances,
        get_rectangle_vertices, get_short_term_reference_path, interX,
        transform_from_global_to_local_coordinate,
    )
    from sigmarl.helper_training import WorldCustom
    from sigmarl.helper_common import Vehicle
    from sigmarl.interX_original import interX as interX_np
    import refshim as rs

    g = torch.Generator().manual_seed(7)
    out = {}

    # G1: WorldCustom.step (helper_training.py:797-861) + KinematicBicycleModel (dynamics.py:62-192)
    Bn = 4096
    for dt_name, dt in (("dt05", 0.05), ("dt10", 0.1)):
        world = WorldCustom(Bn, torch.device("cpu"), x_semidim=torch.tensor(4.5), y_semidim=torch.tensor(4.0), dt=dt)
        veh = Vehicle(
            name="a", shape=rs.Box(AGENTS["length"], AGENTS["width"]), u_range=[AGENTS["max_speed"], AGENTS["max_steering"]],
            u_multiplier=[1, 1], max_speed=AGENTS["max_speed"],
            dynamics=KinematicBicycleModel(
                l_f=AGENTS["l_f"], l_r=AGENTS["l_r"], max_speed=AGENTS["max_speed"], min_speed=AGENTS["min_speed"],
                max_steering=AGENTS["max_steering"], min_steering=AGENTS["min_steering"], max_acc=AGENTS["max_acc"],
                min_acc=AGENTS["min_acc"], max_steering_rate=AGENTS["max_steering_rate"],
                min_steering_rate=AGENTS["min_steering_rate"], device="cpu"),
        )
        world.add_agent(veh)
        pos = torch.rand(Bn, 2, generator=g) * 4
        rot = (torch.rand(Bn, 1, generator=g) * 2 - 1) * 7.0
        speed = torch.rand(Bn, 1, generator=g) * 1.2 - 0.1
        steering = (torch.rand(Bn, 1, generator=g) * 2 - 1) * 0.6
        u = torch.stack([torch.rand(Bn, generator=g) * 1.6 - 0.3, (torch.rand(Bn, generator=g) * 2 - 1) * 0.8], dim=-1)
        veh.state.pos, veh.state.rot, veh.state.speed, veh.state.steering = pos.clone(), rot.clone(), speed.clone(), steering.clone()
        veh.action.u = u.clone()
        world.step()
        out[f"g1_{dt_name}_in"] = np_(torch.cat([pos, rot, speed, steering, u], dim=-1))
        out[f"g1_{dt_name}_out"] = np_(torch.cat(
            [veh.state.pos, veh.state.rot, veh.state.speed, veh.state.steering, veh.state.vel, veh.state.sideslip_angle, veh.action.u], dim=-1))

    # G2: get_rectangle_vertices (helper_scenario.py:695-826)
    c = torch.rand(2048, 2, generator=g) * 4
    y = (torch.rand(2048, 1, generator=g) * 2 - 1) * 7
    out["g2_in"] = np_(torch.cat([c, y], dim=-1))
    out["g2_out"] = np_(get_rectangle_vertices(c, y, AGENTS["width"], AGENTS["length"], True))

    # G5: get_distances_between_agents c2c / mtv (helper_scenario.py:999-1145) on a relative-pose grid
    # (the known-answer generator of mtv_based_sm_predictor.py:181-235, reduced) + random poses incl. overlaps
    xs = torch.linspace(-0.5, 0.5, 15)
    ys = torch.linspace(-0.5, 0.5, 15)
    ps = torch.linspace(-math.pi, math.pi, 13)
    gx, gy, gp = torch.meshgrid(xs, ys, ps, indexing="ij")
    rel = torch.stack([gx.reshape(-1), gy.reshape(-1), gp.reshape(-1)], dim=-1)
    extra = torch.cat([(torch.rand(3000, 2, generator=g) - 0.5) * 0.6, (torch.rand(3000, 1, generator=g) * 2 - 1) * math.pi], dim=-1)
    rel = torch.cat([rel, extra], dim=0)
    nb = rel.shape[0]
    ego_pos = torch.rand(nb, 2, generator=g) * 3 + 0.5
    ego_rot = (torch.rand(nb, 1, generator=g) * 2 - 1) * math.pi
    oth_pos = ego_pos + rel[:, 0:2]
    oth_rot = ego_rot + rel[:, 2:3]
    v0 = get_rectangle_vertices(ego_pos, ego_rot, AGENTS["width"], AGENTS["length"], True)
    v1 = get_rectangle_vertices(oth_pos, oth_rot, AGENTS["width"], AGENTS["length"], True)
    verts = torch.stack([v0, v1], dim=1)  # [nb,2,5,2]
    out["g5_vertices"] = np_(verts)
    out["g5_mtv"] = np_(get_distances_between_agents(verts, "mtv", True, torch.tensor(4.5), torch.tensor(4.0)))
    out["g5_c2c"] = np_(get_distances_between_agents(torch.stack([ego_pos, oth_pos]), "c2c", True, torch.tensor(4.5), torch.tensor(4.0)))
    out["g5_pos"] = np_(torch.stack([ego_pos, oth_pos], dim=1))
    # G6: interX rectangle/rectangle (helper_scenario.py:1148-1229), cross-checked with interX_original.py
    hit = interX(v0, v1, False)
    out["g6_hit"] = np_(hit)
    chk = np.zeros(nb, bool)
    for b in range(0, nb, 7):
        chk[b] = bool(interX_np(v0[b].numpy().T.astype(np.float64), v1[b].numpy().T.astype(np.float64)))
    out["g6_hit_original_every7"] = chk

    # G3/G4/G6b: point->polyline, short-term path, rectangle/boundary interX on CPM paths (padded like world_state_rt.py:313-420)
    p = Parameters(n_agents=2, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=1, is_use_mtv_distance=False)
    env = refshim.RefEnv(p, 1)
    ws = env.scenario.world_state
    paths = env.scenario.map.parser.reference_paths
    ext = ws.ref_paths_map_related.point_extended_all
    q_pts, q_path, r_d, r_i, r_dl, r_il, r_dr, r_ir, r_st, r_hitl, r_hitr, q_rot = [], [], [], [], [], [], [], [], [], [], [], []
    tables = {}
    for pid in [0, 5, 17, 39]:
        rp = paths[pid]
        ws._reset_agent_related_ref_path(0, 0, rp, pid, ext)
        ra = ws.ref_paths_agent_related
        lt, lb, rb = ra.long_term[0, 0].clone(), ra.left_boundary[0, 0].clone(), ra.right_boundary[0, 0].clone()
        n_lt, n_l, n_r = ra.n_points_long_term[0, 0].clone(), ra.n_points_left_b[0, 0].clone(), ra.n_points_right_b[0, 0].clone()
        tables[f"g3_path{pid}_long_term"] = np_(lt)
        tables[f"g3_path{pid}_left"] = np_(lb)
        tables[f"g3_path{pid}_right"] = np_(rb)
        tables[f"g3_path{pid}_n"] = np.asarray([int(n_lt), int(n_l), int(n_r), int(bool(rp["is_loop"]))], np.int32)
        n = int(n_lt)
        cl = rp["center_line"]
        M = 400
        idx = torch.randint(0, n, (M,), generator=g)
        idx[:40] = torch.cat([torch.arange(0, 20), torch.arange(n - 20, n)])  # the loop seam
        pts = cl[idx] + (torch.rand(M, 2, generator=g) - 0.5) * 0.16
        rot = ra.long_term.new_tensor(rp["center_line_yaw"][idx.clamp(max=rp["center_line_yaw"].shape[0] - 1)]).reshape(M, 1) + (torch.rand(M, 1, generator=g) - 0.5) * 0.8
        nn = n_lt.repeat(M)
        d, i = get_perpendicular_distances(pts, lt.unsqueeze(0).expand(M, -1, -1), nn)
        dl, il = get_perpendicular_distances(pts, lb.unsqueeze(0).expand(M, -1, -1), n_l.repeat(M))
        dr, ir = get_perpendicular_distances(pts, rb.unsqueeze(0).expand(M, -1, -1), n_r.repeat(M))
        st, _ = get_short_term_reference_path(
            lt.unsqueeze(0).expand(M, -1, -1), i.to(torch.int32), 3, torch.device("cpu"),
            torch.tensor(bool(rp["is_loop"])).repeat(M), nn, 2)
        vv = get_rectangle_vertices(pts, rot, AGENTS["width"], AGENTS["length"], True)
        hl = interX(vv, lb.unsqueeze(0).expand(M, -1, -1), False)
        hr = interX(vv, rb.unsqueeze(0).expand(M, -1, -1), False)
        q_pts.append(np_(pts)); q_rot.append(np_(rot)); q_path.append(np.full(M, pid, np.int32))
        r_d.append(np_(d)); r_i.append(np_(i)); r_dl.append(np_(dl)); r_il.append(np_(il)); r_dr.append(np_(dr)); r_ir.append(np_(ir))
        r_st.append(np_(st)); r_hitl.append(np_(hl)); r_hitr.append(np_(hr))
    out.update(tables)
    out["g3_pts"] = np.concatenate(q_pts); out["g3_rot"] = np.concatenate(q_rot); out["g3_path"] = np.concatenate(q_path)
    out["g3_d_ref"] = np.concatenate(r_d); out["g3_i_ref"] = np.concatenate(r_i)
    out["g3_d_left"] = np.concatenate(r_dl); out["g3_i_left"] = np.concatenate(r_il)
    out["g3_d_right"] = np.concatenate(r_dr); out["g3_i_right"] = np.concatenate(r_ir)
    out["g4_short_term"] = np.concatenate(r_st)
    out["g6_hit_left"] = np.concatenate(r_hitl); out["g6_hit_right"] = np.concatenate(r_hitr)

    # G8: ego transform + angle wrap (helper_scenario.py:1241-1289)
    pi_ = torch.rand(1024, 2, generator=g) * 4
    pj = pi_.unsqueeze(1) + (torch.rand(1024, 6, 2, generator=g) - 0.5) * 1.5
    pj[:, 0] = pi_  # self-transform (atan2(0,0))
    ri = (torch.rand(1024, 1, generator=g) * 2 - 1) * 9
    out["g8_pos_i"], out["g8_pos_j"], out["g8_rot_i"] = np_(pi_), np_(pj), np_(ri)
    out["g8_rel"] = np_(transform_from_global_to_local_coordinate(pi_, pj, ri))
    ang = (torch.rand(4096, generator=g) * 2 - 1) * 20
    out["g8_angle_in"] = np_(ang)
    out["g8_angle_out"] = np_(angle_eliminate_two_pi(ang.clone()))

    path = os.path.join(OUT, "functions.npz")
    np.savez_compressed(path, **out)
    print("functions", f"{os.path.getsize(path)/1e6:.2f} MB", "rect hits", int(out["g6_hit"].sum()), "lane hits", int(out["g6_hit_left"].sum()), int(out["g6_hit_right"].sum()),
          "neg mtv", int((out["g5_mtv"][:, 0, 1] < 0).sum()), "zero mtv", int((out["g5_mtv"][:, 0, 1] == 0).sum()))
    agree = (out["g6_hit"][::7] == chk[::7]).mean()
    print("interX vs interX_original agreement on sampled cases:", agree)


def gen_cbf_functions():
    """Function-level goldens of the CBF margin path: PseudoDistance.get_distance (fp16 bit patterns) on many points, and
    compute_nominal_cbf_constraint_margins / compute_cbf_violation_rewards_from_margins on directly set states (CPM, 16 agents)."""
    import types
    from sigmarl.cbf_qp import CBFQP

    g = torch.Generator().manual_seed(31)
    out = {}
    B, N = 48, 16
    p = Parameters(n_agents=N, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=B, dt=0.05,
                   is_using_cbf_training=True, is_solve_qp=False, rew_method="cbf", is_challenging_initial_state_buffer=False)
    env = refshim.RefEnv(p, B)
    sc = env.scenario
    pd = sc.map_pseudo_distance
    paths = sc.map.parser.reference_paths
    # P1: get_distance on points scattered around the centre lines (incl. far outside the lane)
    q_pts, q_path, q_l, q_r = [], [], [], []
    for pid in [0, 3, 11, 17, 26, 39]:
        cl = paths[pid]["center_line"]
        M = 1500
        idx = torch.randint(0, cl.shape[0], (M,), generator=g)
        spread = torch.where(torch.rand(M, 1, generator=g) < 0.85, torch.tensor(0.12), torch.tensor(0.6))
        pts = (cl[idx] + (torch.rand(M, 2, generator=g) - 0.5) * 2 * spread).to(torch.float32)
        dl, dr = pd.get_distance(pid, pts)
        q_pts.append(np_(pts)); q_path.append(np.full(M, pid, np.int32))
        q_l.append(dl.view(np.uint16).copy()); q_r.append(dr.view(np.uint16).copy())
    out["p1_pts"], out["p1_path"] = np.concatenate(q_pts), np.concatenate(q_path)
    out["p1_left_f16"], out["p1_right_f16"] = np.concatenate(q_l), np.concatenate(q_r)
    # P2: margins on directly set states
    ag = env.world.agents
    path_id = torch.randint(0, len(paths), (B, N), generator=g)
    state = torch.zeros(B, N, 5)
    for b in range(B):
        for i in range(N):
            rp = paths[int(path_id[b, i])]
            cl, yaw = rp["center_line"], rp["center_line_yaw"].reshape(-1)
            k = int(torch.randint(1, cl.shape[0] - 1, (1,), generator=g))
            off = (torch.rand(2, generator=g) - 0.5) * (0.08 if b % 4 else 0.2)
            state[b, i, 0:2] = cl[k] + off
            state[b, i, 2] = yaw[min(k, yaw.shape[0] - 1)] + (torch.rand(1, generator=g) - 0.5) * (0.6 if b % 3 else 2.5)
            state[b, i, 3] = torch.rand(1, generator=g) * 1.2 - 0.2
            state[b, i, 4] = (torch.rand(1, generator=g) - 0.5) * 1.0
    # a few deliberately close pairs (pair-margin violations)
    for b in range(0, B, 2):
        state[b, 1, 0:2] = state[b, 0, 0:2] + torch.tensor([0.11, 0.05])
        state[b, 1, 2] = state[b, 0, 2] + 0.4
        path_id[b, 1] = path_id[b, 0]
    state = state.to(torch.float32)
    for i, a in enumerate(ag):
        a.state.pos = state[:, i, 0:2].clone(); a.state.rot = state[:, i, 2:3].clone()
        a.state.speed = state[:, i, 3:4].clone(); a.state.steering = state[:, i, 4:5].clone()
    act = torch.stack([torch.rand(B, N, generator=g) * 1.8 - 0.6, (torch.rand(B, N, generator=g) - 0.5) * 1.4], dim=-1).to(torch.float32)
    td = {("agents", "info", "path_id"): path_id.clone(), ("agents", "action"): act.clone()}
    fake = types.SimpleNamespace(base_env=types.SimpleNamespace(scenario_name=sc))
    C = int(p.n_circles_approximate_vehicle)
    L = np.zeros((B, N, C)); R = np.zeros((B, N, C)); P = np.full((B, N, N, C, C), np.nan)
    for e in range(B):
        c = CBFQP(env=fake, env_idx=e)
        c.time_pseudo_dis = 0
        m = c.compute_nominal_cbf_constraint_margins(td)
        L[e], R[e] = m["lane_L_margin"], m["lane_R_margin"]
        for (i, j, ci, cj), gval in m["pair_margin"].items():
            P[e, i, j, ci, cj] = gval
        c.update_qp(td)
    # G10: the constraint data of the centralized QP on the same states -- the ADAPTIVE branches of
    # ttcbf_lane_affine_coeffs / ttcbf_pair_affine_coeffs (cbf_qp.py:2337-2447: A, b0 and h kept apart, as update_centralized_cbf_qp
    # :1103-1180 fills them into the cvxpy parameters), the nominal controls of both controller types (:1062-1090) and the CLF errors
    # (:442-459) for reference points given as inputs.  Called on the very methods; nothing is solved (cvxpy / OSQP are absent).
    g10_lane = np.zeros((B, N, C, 2, 4))
    g10_pair = np.full((B, N, N, C, C, 6), np.nan)
    g10_unom = np.zeros((B, N, 2))
    g10_ref = (state[..., 0:2] + (torch.rand(B, N, 2, generator=g) - 0.5) * 1.2).to(torch.float32)
    g10_clf = np.zeros((B, N, 4))  # e_head, e_speed, u1_nom, u2_nom of the "clf" controller
    for e in range(B):
        c = CBFQP(env=fake, env_idx=e)
        c.time_pseudo_dis = 0
        c.adaptive_lambda = True
        d_safe = float(2.0 * c.circle_radius + c.safety_buffer)
        states = [torch.cat([ag[i].state.pos[e], ag[i].state.rot[e], ag[i].state.speed[e], ag[i].state.steering[e]], dim=-1) for i in range(N)]
        circles = [c.get_circle_centers(s_) for s_ in states]
        kins = [c.linearized_center_kinematics_coeffs(s_) for s_ in states]
        for i in range(N):
            _, u_nom_i = c.rl_action_to_u(rl_actions=act[e, i].clone(), v=states[i][3], steering=states[i][4])
            g10_unom[e, i] = u_nom_i.detach().cpu().numpy().reshape(2)
            e_h, e_v = c._clf_errors_for_agent(states[i], g10_ref[e, i])
            g10_clf[e, i] = (e_h, e_v, np.clip(c.k_clf_speed * e_v, c.a_min, c.a_max), np.clip(c.k_clf_heading * e_h, c.steering_rate_min, c.steering_rate_max))
            for ci in range(C):
                smL, gL, HL, smR, gR, HR = c.estimate_agent_2_lane_safety_margin(circles[i][ci][0:2], int(path_id[e, i]))
                for side, (sm, gg, HH) in enumerate(((smL, gL, HL), (smR, gR, HR))):
                    A_, b0_, h_ = c.ttcbf_lane_affine_coeffs(kins[i], ci, sm, gg, HH, c.dt_taylor, None)
                    g10_lane[e, i, ci, side] = (A_[0, 0], A_[0, 1], b0_[0], h_[0])
        for i in range(N - 1):
            for j in range(i + 1, N):
                for ci in range(C):
                    for cj in range(C):
                        delta = circles[i][ci][0:2] - circles[j][cj][0:2]
                        Ai, Aj, b0_, h_ = c.ttcbf_pair_affine_coeffs(kins[i], kins[j], ci, cj, float(delta[0].item()), float(delta[1].item()),
                                                                     d_safe * d_safe, c.dt_taylor, None)
                        g10_pair[e, i, j, ci, cj] = (Ai[0, 0], Ai[0, 1], Aj[0, 0], Aj[0, 1], b0_[0], h_[0])
    out["g10_lane"], out["g10_pair"], out["g10_unom"], out["g10_ref"], out["g10_clf"] = g10_lane, g10_pair, g10_unom, np_(g10_ref), g10_clf
    ri = sc.reward_info
    out["p2_state"], out["p2_path"], out["p2_act"] = np_(state), np_(path_id).astype(np.int32), np_(act)
    out["p2_lane_left"], out["p2_lane_right"], out["p2_pair"] = L, R, P
    cen = np.zeros((B, N, C, 2), np.float32)  # the reference's own float32 circle centres of these states (see CbfHook)
    c0 = CBFQP(env=fake, env_idx=0)
    for e in range(B):
        for i in range(N):
            st_i = torch.cat([ag[i].state.pos[e], ag[i].state.rot[e], ag[i].state.speed[e], ag[i].state.steering[e]], dim=-1)
            cen[e, i] = np_(c0.get_circle_centers(st_i)[:, 0:2])
    out["p2_centers"] = cen
    out["p2_rew"] = np.stack([np_(ri.rew_near_left_lane), np_(ri.rew_near_right_lane), np_(ri.rew_near_other_agents)], axis=0)
    out["meta_json"] = np.asarray(json.dumps(dict(B=B, N=N, dt=p.dt, h_nom=p.h_nom, n_circles=C, scenario_type="cpm_entire",
                                                  numpy_version=np.__version__, torch_version=torch.__version__)))
    path = os.path.join(OUT, "cbf_functions.npz")
    np.savez_compressed(path, **out)
    print("cbf_functions", f"{os.path.getsize(path)/1e6:.2f} MB", "neg L/R/P", int((L < 0).sum()), int((R < 0).sum()), int((P < 0).sum()),
          "rew nonzero", (out["p2_rew"] != 0).sum(axis=(1, 2)), "1000s", int((out["p1_left_f16"] == np.float16(1000).view(np.uint16)).sum()))


def gen_adversarial():
    """Adversarial inputs for the collision PRUNING of the HIP path: the strict-sign interX predicate
    (helper_scenario.py:1148-1229) evaluated by the reference on configurations where a pruned far segment / far rectangle could matter --
    rectangle edges exactly collinear with long straight boundary stretches (so that far boundary segments are collinear with an edge),
    offsets of a few ulps around touching, and vehicles in line whose side edges are collinear at centre distances around and far beyond
    the circumcircle sum.  Stored: poses, the reference's vertices and its hit flags (rectangle x left / right boundary of the path,
    rectangle x rectangle).  The oracle (full scan) is held to these flags, the HIP kernels to the oracle on the same poses."""
    from sigmarl.constants import AGENTS
    from sigmarl.helper_scenario import get_rectangle_vertices, interX

    p = Parameters(n_agents=2, scenario_type="cpm_entire", is_obs_noise=False, is_apply_mask=False, num_vmas_envs=1, is_use_mtv_distance=False)
    env = refshim.RefEnv(p, 1)
    ws = env.scenario.world_state
    paths = env.scenario.map.parser.reference_paths
    ext = ws.ref_paths_map_related.point_extended_all
    W, L = AGENTS["width"], AGENTS["length"]
    g = torch.Generator().manual_seed(41)
    poses, pids, verts, hl, hr = [], [], [], [], []
    for pid in range(len(paths)):
        rp = paths[pid]
        ws._reset_agent_related_ref_path(0, 0, rp, pid, ext)
        ra = ws.ref_paths_agent_related
        lb, rb = ra.left_boundary[0, 0].clone(), ra.right_boundary[0, 0].clone()
        nl, nr = int(ra.n_points_left_b[0, 0]), int(ra.n_points_right_b[0, 0])
        for side, (poly, n) in enumerate(((lb, nl), (rb, nr))):
            d = poly[1:n] - poly[: n - 1]
            for ax in (0, 1):  # stretches where one coordinate is EXACTLY constant
                same = (d[:, ax] == 0).numpy()
                k = 0
                while k < len(same):
                    if not same[k]:
                        k += 1
                        continue
                    k0 = k
                    while k < len(same) and same[k]:
                        k += 1
                    if k - k0 < 8:
                        continue
                    c0 = float(poly[k0, ax])              # the constant coordinate of the stretch
                    other = 1 - ax
                    lo_, hi_ = float(poly[k0, other]), float(poly[k, other])
                    yaw = 0.0 if ax == 1 else math.pi / 2  # vehicle along the stretch
                    for frac in (0.15, 0.5, 0.85):
                        along = lo_ + frac * (hi_ - lo_)
                        for sgn in (-1.0, 1.0):
                            for off in (0.0, 1e-7, -1e-7, 6e-8, -6e-8, 1e-6, -1e-6, 1e-4, -1e-4, 3e-3):
                                for dyaw in (0.0, 1e-7, -1e-7, 1e-4):
                                    pos = [0.0, 0.0]
                                    pos[ax] = np.float32(c0 + sgn * (W / 2) + off)
                                    pos[other] = np.float32(along)
                                    poses.append([pos[0], pos[1], np.float32(yaw + dyaw)])
                                    pids.append(pid)
                    break
                else:
                    continue
            if len(poses) > 6000:
                break
        if len(poses) > 6000:
            break
    poses = torch.tensor(np.asarray(poses, np.float32))
    pids = np.asarray(pids, np.int32)
    vv = get_rectangle_vertices(poses[:, 0:2], poses[:, 2:3], W, L, True)
    hits_l, hits_r = np.zeros(len(pids), bool), np.zeros(len(pids), bool)
    for pid in np.unique(pids):
        ws._reset_agent_related_ref_path(0, 0, paths[pid], int(pid), ext)
        ra = ws.ref_paths_agent_related
        m = torch.from_numpy(pids == pid)
        M = int(m.sum())
        hits_l[m.numpy()] = np_(interX(vv[m], ra.left_boundary[0, 0].unsqueeze(0).expand(M, -1, -1), False))
        hits_r[m.numpy()] = np_(interX(vv[m], ra.right_boundary[0, 0].unsqueeze(0).expand(M, -1, -1), False))
    out = dict(b_pose=np_(poses), b_path=pids, b_vertices=np_(vv), b_hit_left=hits_l, b_hit_right=hits_r)
    # vehicles in line: collinear side edges, centre distances around 2 R (R = circumradius) and far beyond; small lateral / yaw perturbations
    R2 = 2.0 * math.sqrt((L / 2) ** 2 + (W / 2) ** 2)
    pa, pb = [], []
    for base_yaw in (0.0, math.pi / 2, 0.7):
        for dist in (0.2, 0.2200001, 0.22, 0.2199999, 0.23, R2 - 1e-6, R2, R2 + 1e-6, R2 + 9e-5, R2 + 1.1e-4, 0.3, 0.6, 1.5, 3.0):
            for lat in (0.0, 1e-7, -1e-7, 1e-6, W, W + 1e-7, W - 1e-7):
                for dyaw in (0.0, 1e-7, 1e-4):
                    c, s_ = math.cos(base_yaw), math.sin(base_yaw)
                    a = [2.0, 2.0, base_yaw]
                    b = [2.0 + dist * c - lat * s_, 2.0 + dist * s_ + lat * c, base_yaw + dyaw]
                    pa.append(a); pb.append(b)
    pa, pb = torch.tensor(np.asarray(pa, np.float32)), torch.tensor(np.asarray(pb, np.float32))
    va = get_rectangle_vertices(pa[:, 0:2], pa[:, 2:3], W, L, True)
    vb = get_rectangle_vertices(pb[:, 0:2], pb[:, 2:3], W, L, True)
    out.update(r_pose_a=np_(pa), r_pose_b=np_(pb), r_vertices_a=np_(va), r_vertices_b=np_(vb), r_hit=np_(interX(va, vb, False)))
    path = os.path.join(OUT, "adversarial.npz")
    np.savez_compressed(path, **out)
    print("adversarial", f"{os.path.getsize(path)/1e6:.2f} MB", "boundary cases", len(pids), "paths", len(np.unique(pids)), "hits L/R", int(hits_l.sum()), int(hits_r.sum()),
          "pairs", len(pa), "pair hits", int(out["r_hit"].sum()))


TRAJS = {
    "cpm16_c2c": dict(T=32, B=4, seed=11, mode_pattern=[1, 0, 1, 1], n_agents=16, scenario_type="cpm_entire", dt=0.05,
                      is_use_mtv_distance=False, rew_method="distance"),
    "cpm16_mtv": dict(T=32, B=4, seed=12, mode_pattern=[1, 1, 0, 1], n_agents=16, scenario_type="cpm_entire", dt=0.1,
                      is_use_mtv_distance=True, rew_method="ttc_sparse"),
    "intersection4_c2c": dict(T=64, B=2, seed=13, mode_pattern=[1, 1], n_agents=4, scenario_type="intersection_1", dt=0.1,
                              is_use_mtv_distance=False, rew_method="distance_sparse"),
    "onramp6_mtv": dict(T=64, B=3, seed=14, mode_pattern=[1, 1, 0], n_agents=6, scenario_type="on_ramp_1", dt=0.1,
                        is_use_mtv_distance=True, rew_method="ttc"),
    # done envs are NOT reset: long runs with persistent collisions and agents far off their lanes
    "cpm16_c2c_noreset": dict(T=40, B=3, seed=16, mode_pattern=[1, 0, 1], no_reset=True, n_agents=16, scenario_type="cpm_entire",
                              dt=0.05, is_use_mtv_distance=False, rew_method="ttc"),
    "cpm8_mtv_noreset": dict(T=40, B=3, seed=17, mode_pattern=[0, 1, 1], no_reset=True, n_agents=8, scenario_type="cpm_entire",
                             dt=0.1, is_use_mtv_distance=True, rew_method="distance_sparse"),
    "cpmmixed4_c2c": dict(T=48, B=4, seed=15, mode_pattern=[1, 0, 1, 1], n_agents=4, scenario_type="cpm_mixed", dt=0.05,
                          is_use_mtv_distance=False, rew_method="sparse", cpm_scenario_probabilities=[1.0, 0.0, 0.0]),
    # rew_method "cbf" with the QP-free margin reward (is_solve_qp=False): cbf_qp.py:2534-2804 run before every step
    "cpm16_cbf": dict(T=12, B=2, seed=21, mode_pattern=[1, 0], hook="cbf", n_agents=16, scenario_type="cpm_entire", dt=0.05,
                      is_use_mtv_distance=False, rew_method="cbf", is_using_cbf_training=True, is_solve_qp=False),
    "intersection4_cbf": dict(T=48, B=3, seed=22, mode_pattern=[1, 1, 0], hook="cbf", n_agents=4, scenario_type="intersection_1", dt=0.1,
                              is_use_mtv_distance=True, rew_method="cbf_sparse", is_using_cbf_training=True, is_solve_qp=False),
    # Parameters' default is_apply_mask=True on the CPM map (no neighbouring-lanelet table there: the distance mask only)
    "cpm16_mask": dict(T=24, B=3, seed=24, mode_pattern=[1, 0, 1], n_agents=16, scenario_type="cpm_entire", dt=0.05,
                       is_use_mtv_distance=False, rew_method="distance", is_apply_mask=True),
    # is_apply_mask on OSM maps (whose parser lists neighbouring lanelets): in ego view the lanelet mask stays empty -- the agents' lanelets
    # are only computed in the bird-view branch (observation_provider_rt.py:537-588) -- so these pin the distance mask there as well
    "intersection4_mask": dict(T=40, B=3, seed=25, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="intersection_1", dt=0.1,
                               is_use_mtv_distance=False, rew_method="distance", is_apply_mask=True),
    "roundabout6_mask": dict(T=40, B=2, seed=26, mode_pattern=[1, 0], n_agents=6, scenario_type="roundabout_2", dt=0.1,
                             is_use_mtv_distance=True, rew_method="ttc", is_apply_mask=True),
    # Parameters.reset_agent_fixed_duration: every env is done whenever t = step * dt hits a multiple of it (road_traffic.py:1388-1397)
    "cpm8_fixed_reset": dict(T=32, B=4, seed=27, mode_pattern=[1, 1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05,
                             is_use_mtv_distance=False, rew_method="distance", reset_agent_fixed_duration=0.5),
    "intersection4_fixed_testing": dict(T=48, B=3, seed=28, mode_pattern=[1, 0, 1], n_agents=4, scenario_type="intersection_1", dt=0.1,
                                        is_use_mtv_distance=False, rew_method="distance", is_testing_mode=True, reset_agent_fixed_duration=1.5),
    # non-default observation switches (observation_provider_rt.py:803-925): steering + the neighbours' reference paths (with the distance
    # mask on); position / rotation / length / width instead of vertices, without the distance columns
    "cpm8_obs_steer_ref": dict(T=24, B=3, seed=29, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                               rew_method="distance", is_apply_mask=True, is_obs_steering=True, is_observe_ref_path_other_agents=True),
    "intersection4_obs_novert": dict(T=32, B=3, seed=30, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="intersection_1", dt=0.1,
                                     is_use_mtv_distance=True, rew_method="ttc", is_observe_vertices=False, is_observe_distance_to_agents=False,
                                     is_observe_distance_to_center_line=False, is_obs_steering=True),
    # bird view (is_ego_view=False): world-frame observation normalised by the world size, own position / rotation in front
    "cpm8_birdview": dict(T=24, B=3, seed=31, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                          rew_method="distance", is_ego_view=False, is_apply_mask=False, is_observe_ref_path_other_agents=True),
    "intersection4_birdview_novert": dict(T=24, B=3, seed=32, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="intersection_1", dt=0.1,
                                          is_use_mtv_distance=True, rew_method="ttc", is_ego_view=False, is_apply_mask=False, is_observe_vertices=False,
                                          is_obs_steering=True),
    # boundary points instead of boundary distances (is_observe_distance_to_boundaries=False), ego view and bird view
    "cpm8_boundary_points": dict(T=24, B=3, seed=33, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                                 rew_method="distance", is_observe_distance_to_boundaries=False),
    "onramp4_boundary_points_bird": dict(T=32, B=3, seed=34, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="on_ramp_1", dt=0.1, is_use_mtv_distance=False,
                                         rew_method="distance", is_observe_distance_to_boundaries=False, is_ego_view=False, is_apply_mask=False),
    # bird view WITH is_apply_mask: the lanelet-relation mask (observation_provider_rt.py:577-665, map_manager.py:41-118) -- on OSM maps the parser's
    # neighbouring-lanelet table masks observed neighbours whose lanelet the ego's lanelet does not list; on the CPM map the table is empty
    "intersection4_birdview_mask": dict(T=40, B=3, seed=35, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="intersection_1", dt=0.1, is_use_mtv_distance=False,
                                        rew_method="distance", is_ego_view=False, is_apply_mask=True),
    "roundabout6_birdview_mask": dict(T=32, B=3, seed=36, mode_pattern=[1, 0, 1], n_agents=6, scenario_type="roundabout_2", dt=0.1, is_use_mtv_distance=True,
                                      rew_method="ttc", is_ego_view=False, is_apply_mask=True, is_observe_vertices=False),
    "cpm8_birdview_mask": dict(T=24, B=3, seed=37, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                               rew_method="distance", is_ego_view=False, is_apply_mask=True),
    # opponent modelling: the observation row ends with n_nearing x n_actions zero placeholders (observation_provider_rt.py:606-611)
    "cpm8_opponent_pad": dict(T=24, B=3, seed=38, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                              rew_method="distance", is_using_opponent_modeling=True),
    # n_points_short_term != 3 (road_traffic.py:316, :536-543; helper_scenario.py:892-957): longer / shorter short-term reference path, observation
    # width 4 + 2 n + 11 k, progress-reward weights linspace(1, 0.2, n) / sum
    "cpm8_ns5": dict(T=24, B=3, seed=39, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                     rew_method="distance", n_points_short_term=5),
    "intersection4_ns2": dict(T=40, B=3, seed=40, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="intersection_1", dt=0.1, is_use_mtv_distance=True,
                              rew_method="ttc", n_points_short_term=2),
    # the remaining map families (the oracle is otherwise pinned on cpm / intersection_1 / on_ramp_1 / roundabout_2 trajectories): interchange, the larger
    # intersections, the small roundabout, the multi-lane on-ramp -- training and testing mode, every distance / reward combination once more
    "interchange6_mtv": dict(T=48, B=3, seed=41, mode_pattern=[1, 1, 0], n_agents=6, scenario_type="interchange_2", dt=0.1, is_use_mtv_distance=True,
                             rew_method="ttc"),
    "intersection5_6_testing": dict(T=48, B=3, seed=42, mode_pattern=[1, 0, 1], n_agents=6, scenario_type="intersection_5", dt=0.1, is_use_mtv_distance=False,
                                    rew_method="distance_sparse", is_testing_mode=True),
    "roundabout1_5_c2c": dict(T=48, B=3, seed=43, mode_pattern=[1, 1, 0], n_agents=5, scenario_type="roundabout_1", dt=0.1, is_use_mtv_distance=False,
                              rew_method="ttc_sparse"),
    "onramp2_6_mask": dict(T=48, B=3, seed=44, mode_pattern=[1, 0, 1], n_agents=6, scenario_type="on_ramp_2_multilane", dt=0.1, is_use_mtv_distance=True,
                           rew_method="distance", is_apply_mask=True),
    "interchange1_8_birdview": dict(T=32, B=3, seed=45, mode_pattern=[1, 1, 0], n_agents=8, scenario_type="interchange_1", dt=0.1, is_use_mtv_distance=False,
                                    rew_method="distance", is_ego_view=False, is_apply_mask=True, is_obs_steering=True),
    # second round of cross-combinations: testing mode with mtv / ttc on the multi-lane on-ramp, the CBF margin reward on an interchange, the merge
    # sub-scenarios of cpm_mixed (two agents: more do not fit the merge lists, tests/golden/gen/gen_reset_distribution.py; seed 51: with 48 and 50 the
    # reference's unbounded rejection loop never returns -- a first agent on the merge point leaves the second no feasible start), bird view without vertices
    "onramp2_8_testing_mtv": dict(T=40, B=3, seed=46, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="on_ramp_2_multilane", dt=0.1, is_use_mtv_distance=True,
                                  rew_method="ttc", is_testing_mode=True),
    "interchange2_6_cbf": dict(T=24, B=2, seed=47, mode_pattern=[1, 0], hook="cbf", n_agents=6, scenario_type="interchange_2", dt=0.1,
                               is_use_mtv_distance=False, rew_method="cbf", is_using_cbf_training=True, is_solve_qp=False),
    "cpmmixed2_merge": dict(T=48, B=6, seed=51, mode_pattern=[1, 1, 0], n_agents=2, scenario_type="cpm_mixed", dt=0.05, is_use_mtv_distance=False,
                            rew_method="distance", cpm_scenario_probabilities=[0.2, 0.4, 0.4]),
    "intersection8_6_bird_novert": dict(T=40, B=3, seed=49, mode_pattern=[1, 1, 0], n_agents=6, scenario_type="intersection_8", dt=0.1, is_use_mtv_distance=True,
                                        rew_method="distance_sparse", is_ego_view=False, is_observe_vertices=False, is_apply_mask=True, is_observe_distance_to_agents=False),
    # full observation (is_partial_observation=False, observation_provider_rt.py:756-851): bird view only (the ego view raises in the reference); every
    # agent's row carries ALL agents' features, cut into n_nearing_agents_observed chunks by the reference's reshape; the mutual distances are all zero
    "intersection4_full_bird": dict(T=32, B=3, seed=52, mode_pattern=[1, 1, 0], n_agents=4, scenario_type="intersection_1", dt=0.1, is_use_mtv_distance=False,
                                    rew_method="distance", is_ego_view=False, is_partial_observation=False, is_apply_mask=False),
    "intersection4_full_bird_novert": dict(T=32, B=3, seed=53, mode_pattern=[1, 0, 1], n_agents=4, scenario_type="intersection_1", dt=0.1, is_use_mtv_distance=True,
                                           rew_method="ttc", is_ego_view=False, is_partial_observation=False, is_apply_mask=True, is_observe_vertices=False,
                                           is_obs_steering=True),
    "roundabout6_full_bird_k3": dict(T=32, B=3, seed=54, mode_pattern=[1, 1, 0], n_agents=6, scenario_type="roundabout_2", dt=0.1, is_use_mtv_distance=True,
                                     rew_method="distance_sparse", is_ego_view=False, is_partial_observation=False, is_apply_mask=False, n_nearing_agents_observed=3,
                                     is_observe_ref_path_other_agents=True, is_observe_distance_to_center_line=False),
    "cpm8_full_bird_pad": dict(T=24, B=3, seed=55, mode_pattern=[1, 0, 1], n_agents=8, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False,
                               rew_method="distance", is_ego_view=False, is_partial_observation=False, is_apply_mask=False, is_using_opponent_modeling=True,
                               is_observe_distance_to_agents=False, is_observe_distance_to_boundaries=False),
    # BASELINE config 4: 32 agents on the on-ramp map.  The reference's rejection sampler cannot place them, so the
    # start is injected (Parameters.predefined_ref_path_idx / init_state); vehicles overlap from the first step on, every env is "done" at
    # every step and none is reset: non-reset steps only, as the survey prescribes for this configuration
    "onramp32_c2c": dict(T=32, B=3, seed=27, mode_pattern=[1, 0, 1], no_reset=True, inject=dict(first_point=3, stride=3), n_agents=32,
                         scenario_type="on_ramp_1", dt=0.05, is_use_mtv_distance=False, rew_method="distance"),
    # "clf" nominal controller (cbf_qp.py:2616-2628): the margins are evaluated at a P controller's action instead of the policy's
    "onramp4_cbf_clf": dict(T=24, B=3, seed=23, mode_pattern=[1, 0, 1], hook="cbf", n_agents=4, scenario_type="on_ramp_1", dt=0.05,
                            is_use_mtv_distance=False, rew_method="cbf", is_using_cbf_training=True, is_solve_qp=False, nom_controller_type="clf"),
}

def gen_initial_reset():
    """The reference's INITIAL reset from a seeded torch generator: `torch.manual_seed(s)` right before `env_reset_world_at(None)` (every import
    and make_world done -- importing road_traffic itself consumes draws), so that a caller of the mirrored surface who seeds at the same point
    must get the same start states (road_traffic.py:832-834, world_state_rt_sim.py:215-311)."""
    out = {}
    cases = [("cpm16", dict(n_agents=16, scenario_type="cpm_entire"), 6, 5), ("intersection4", dict(n_agents=4, scenario_type="intersection_1"), 8, 6),
             ("cpm8_testing", dict(n_agents=8, scenario_type="cpm_entire", is_testing_mode=True), 4, 7)]
    for nme, kw, B, seed in cases:
        p = Parameters(is_obs_noise=False, is_apply_mask=False, max_steps=128, num_vmas_envs=B, is_challenging_initial_state_buffer=False, **kw)
        refshim.install()
        from sigmarl.scenarios.road_traffic import ScenarioRoadTraffic

        sc = ScenarioRoadTraffic()
        sc.parameters = p
        world = sc.env_make_world(B, torch.device("cpu"))
        torch.manual_seed(seed)
        sc.env_reset_world_at(None)
        rp = sc.world_state.ref_paths_agent_related
        out[nme + "_path_id"] = np_(rp.path_id).astype(np.int32)
        out[nme + "_point_id"] = np_(rp.point_id).astype(np.int32) if hasattr(rp, "point_id") else np.zeros((B, len(world.agents)), np.int32)
        out[nme + "_pos"] = np.stack([np_(a.state.pos) for a in world.agents], axis=1).astype(np.float32)
        out[nme + "_rot"] = np.stack([np_(a.state.rot)[:, 0] for a in world.agents], axis=1).astype(np.float32)
        out[nme + "_speed"] = np.stack([np_(a.state.speed)[:, 0] for a in world.agents], axis=1).astype(np.float32)
        out[nme + "_meta"] = np.asarray([B, len(world.agents), seed], np.int32)
    np.savez_compressed(os.path.join(OUT, "initial_reset.npz"), **out)
    print("initial_reset:", {k: v.shape for k, v in out.items() if k.endswith("_pos")})


def content_hash(path):
    """sha256 over the fixture's CONTENT (sorted keys, dtype, shape, raw bytes): independent of the zip container's timestamps."""
    import hashlib

    z = np.load(path)
    h = hashlib.sha256()
    for k in sorted(z.files):
        a = np.ascontiguousarray(z[k])
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    return h.hexdigest()


def generate_one(nme):
    if nme == "functions":
        gen_functions()
    elif nme == "cbf_functions":
        gen_cbf_functions()
    elif nme == "adversarial":
        gen_adversarial()
    elif nme == "initial_reset":
        gen_initial_reset()
    elif nme == "cbf_grouped":  # needs its own cvxpy stand-in installed before the reference is imported: own script
        import subprocess

        subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen_cbf_grouped.py")])
    else:
        run_traj(nme, **TRAJS[nme])


if __name__ == "__main__":
    # THE recipe for the committed fixtures:   python tests/golden/gen/gen_golden.py
    # Every fixture is generated in its OWN interpreter (fresh RNG / module state, PYTHONHASHSEED=0), so the bytes do not depend on which
    # fixtures were generated before it; then tests/golden/MANIFEST.json (content hashes) is rewritten.  `--one <name>` is the worker form.
    import subprocess

    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        generate_one(sys.argv[2])
        sys.exit(0)
    names = sys.argv[1:] or (["functions", "cbf_functions", "adversarial", "initial_reset", "cbf_grouped"] + list(TRAJS))
    envv = dict(os.environ, PYTHONHASHSEED="0")
    for nme in names:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", nme], env=envv)
    man_path = os.path.join(OUT, "MANIFEST.json")
    man = json.load(open(man_path)) if os.path.exists(man_path) else {}
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            man[f] = content_hash(os.path.join(OUT, f))
    json.dump(man, open(man_path, "w"), indent=1, sort_keys=True)
    print("manifest:", man_path, len(man), "fixtures")
