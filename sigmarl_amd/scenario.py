"""Host-side mirror of the reference's VMAS plugin surface, backed by the fused HIP step.

Same names, argument meaning and error behaviour as the reference for THIS path, so that ``sigmarl/mappo_cavs.py`` can
consume it as a drop-in (``scenario = ScenarioRoadTraffic(); scenario.parameters = parameters; VmasEnv(scenario=scenario, ...)``):

  ScenarioRoadTraffic      <- sigmarl/scenarios/road_traffic.py:72   (make_world :104, reset_world_at :816, reward :925,
                                                                       observation :1334, done :1368, info :1489)
  WorldCustom              <- sigmarl/helper_training.py:791          (step :797 -> ONE fused launch: sigmaenv_step)
  Vehicle / VehicleState   <- sigmarl/helper_common.py:290-430        (tensors are zero-copy views of the device state buffer)

VMAS drives a scenario as ``world.step()`` then ``reward(a)`` for all agents, ``observation(a)`` for all agents, ``info(a)``,
``done()`` (sigmarl's ``if agent_index == 0`` guards rely on that order).  Here ``world.step()`` runs the whole fused kernel
(dynamics -> distances/collisions -> rewards -> observations -> done flags); the callbacks only hand out views of the results
and ``done()`` performs the host-driven per-agent resets the reference performs there (RNG stays in torch, as in the reference).

vmas itself is absent from the build image (and cannot be installed: no network), so what is verified is this: ``ScenarioRoadTraffic`` derives
from ``vmas.simulator.scenario.BaseScenario`` when vmas is importable and from a stand-in with the same five entry points otherwise;
``WorldCustom`` and ``Vehicle`` are plain classes (NOT subclasses of ``vmas.simulator.core.World`` / ``Agent``) that carry every attribute
``vmas.simulator.environment.Environment`` touches on them while it resets, sets actions, steps and collects results -- ``world.policy_agents``,
``world.dim_c`` / ``dim_p``, ``world.to``, ``agent.action_size``, ``agent.silent`` / ``movable`` / ``action_script``, ``agent.action.u``,
``action.u_range_tensor`` / ``u_multiplier_tensor`` / ``u_noise`` -- and ``tests/vmas_env_shim.py`` restates that driver (attribute accesses of
VMAS 1.4.3's ``Environment.__init__ / reset / reset_at / step / _set_action / get_from_scenario / done``, from its published source; third-party,
unverifiable here) and ``tests/test_gpu_scenario.py`` runs the mirror under it.  It has never run under the real ``VmasEnv``.
A CUDA/HIP device is mandatory -- there is no CPU fallback.
"""
from __future__ import annotations

import math
import time
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import capi
from .env import SigmaEnv
from .maps import load_map
from .params import Parameters, make_config

try:  # pragma: no cover - vmas is not installed in the build image
    from vmas.simulator.scenario import BaseScenario as _VmasBaseScenario
except Exception:  # noqa: BLE001
    _VmasBaseScenario = None
# The reference's WorldCustom / Vehicle / VehicleState ARE vmas classes (helper_training.py:791, helper_common.py:382, :290): with vmas importable the classes below derive
# from the same bases, so that an ``isinstance(world, vmas.simulator.core.World)`` in VMAS / TorchRL / user code holds.  The bases' allocating ``__init__`` is NOT called
# (the state stays a view of the library's buffers); every attribute the mirror keeps on its instances is declared at class level below, which takes the bases'
# read-only properties of the same names (Entity.name, Agent.u_range, World.x_semidim, TorchVectorizedObject.batch_dim ...) out of the way
# (tests/test_host.py::test_mirror_classes_derive_from_the_vmas_bases, with a stand-in vmas package of the published attribute names).
try:  # pragma: no cover
    from vmas.simulator.core import Agent as _VmasAgent, AgentState as _VmasAgentState, World as _VmasWorld
except Exception:  # noqa: BLE001
    _VmasAgent = _VmasAgentState = _VmasWorld = None
_WorldBase, _AgentBase, _AgentStateBase = (_VmasWorld or object), (_VmasAgent or object), (_VmasAgentState or object)


class _BaseScenarioStandIn:
    """The slice of ``vmas.simulator.scenario.BaseScenario`` that VMAS' Environment calls."""

    def __init__(self):
        self._world = None

    @property
    def world(self):
        return self._world

    def env_make_world(self, batch_dim, device, **kwargs):
        self._world = self.make_world(batch_dim, device, **kwargs)
        return self._world

    def env_reset_world_at(self, env_index):
        self.world.reset(env_index)
        self.reset_world_at(env_index)

    def pre_step(self):
        return

    def post_step(self):
        return

    def env_process_action(self, agent):
        return

    def extra_render(self, env_index: int = 0):
        return []


BaseScenario = _VmasBaseScenario or _BaseScenarioStandIn

AGENTS = capi.AGENTS


class Box:
    def __init__(self, length: float, width: float):
        self.length, self.width = length, width


class KinematicBicycleModel:
    """Constants holder mirroring ``sigmarl/dynamics.py:11-60``; the integration itself runs inside the fused kernel."""

    def __init__(self, **kw):
        self.l_f, self.l_r = kw.get("l_f", AGENTS["l_f"]), kw.get("l_r", AGENTS["l_r"])
        self.l_wb = self.l_f + self.l_r
        for k in ("max_speed", "min_speed", "max_steering", "min_steering", "max_acc", "min_acc", "max_steering_rate", "min_steering_rate"):
            setattr(self, k, kw.get(k, AGENTS[k]))
        self.device = kw.get("device", "cuda")

    @property
    def needed_action_size(self) -> int:
        return 2


class _Action:
    """``vmas.simulator.core.Action`` as ``Environment._set_action`` uses it: ``u`` (None before the first step, road_traffic.py:1505,
    observation_provider_rt.py:948), the per-dimension range / multiplier tensors, no action noise."""

    def __init__(self, u_range, u_multiplier, device):
        self.u = None
        self.c = None
        self.u_range, self.u_multiplier, self.u_noise = u_range, u_multiplier, 0.0
        self.action_size = len(u_range)
        self.u_range_tensor = torch.tensor([float(x) for x in u_range], dtype=torch.float32, device=device)
        self.u_multiplier_tensor = torch.tensor([float(x) for x in u_multiplier], dtype=torch.float32, device=device)
        self.u_noise_tensor = torch.zeros(self.action_size, dtype=torch.float32, device=device)

    def to(self, device):
        for k in ("u_range_tensor", "u_multiplier_tensor", "u_noise_tensor"):
            setattr(self, k, getattr(self, k).to(device))


class VehicleState(_AgentStateBase):
    """pos/rot/vel + speed/steering/sideslip_angle (helper_common.py:290-379) as views of ``SIGMAENV_BUF_STATE[:, i]``."""

    batch_dim = device = None  # (instance attributes here; properties with a set-once rule on vmas' TorchVectorizedObject)

    def __init__(self, state_row: torch.Tensor):
        self._s = state_row  # [B, 8] strided view: x, y, psi, speed, steering, vx, vy, sideslip
        self.batch_dim, self.device = state_row.shape[0], state_row.device
        # what the bases' own accessors read (EntityState / AgentState: no angular velocity, communication, force or torque here)
        self._batch_dim, self._device = self.batch_dim, self.device
        self._ang_vel = self._c = self._force = self._torque = None

    pos = property(lambda self: self._s[:, 0:2])
    rot = property(lambda self: self._s[:, 2:3])
    speed = property(lambda self: self._s[:, 3:4])
    steering = property(lambda self: self._s[:, 4:5])
    vel = property(lambda self: self._s[:, 5:7])
    sideslip_angle = property(lambda self: self._s[:, 7:8])


class Vehicle(_AgentBase):
    """``Vehicle(Agent)`` of helper_common.py:382-430.  Direct state writes do NOT refresh the derived tensors; go through
    ``ScenarioRoadTraffic.reset_world_at`` / ``SigmaEnv.reset`` for that."""

    # instance attributes of the mirror (read-only properties on vmas' Entity / Agent: see the note at the imports)
    name = shape = color = collide = render_action = u_range = u_multiplier = max_speed = dynamics = batch_dim = device = None
    action_size = silent = movable = rotatable = action_script = is_scripted_ai = obs_range = obs_noise = discrete_action_nvec = None

    def __init__(self, name, state_row, shape, u_range, max_speed, dynamics, color=None):
        self.name = name
        self.shape = shape
        self.color = color
        self.collide = False
        self.render_action = False
        self.u_range = u_range
        self.u_multiplier = [1, 1]
        self.max_speed = max_speed
        self.dynamics = dynamics
        self._state = VehicleState(state_row)
        self.batch_dim = self._batch_dim = state_row.shape[0]
        self.device = self._device = state_row.device
        self._action = _Action(u_range, self.u_multiplier, self.device)
        # what vmas' Environment reads on an agent while it sets actions and collects results (Agent(..., dynamics=...) derives action_size from
        # dynamics.needed_action_size; road_traffic.py:788-813 passes collide=False, render_action=False, no action script, silent by default)
        self.action_size = dynamics.needed_action_size
        self.silent, self.movable, self.rotatable, self.action_script, self.is_scripted_ai = True, True, True, None, False
        self.obs_range, self.obs_noise, self.discrete_action_nvec = None, None, [3] * self.action_size

    def to(self, device):  # TorchVectorizedObject.to: the state lives in the library's device buffers, only the small tensors move
        self._action.to(device)

    @property
    def state(self):
        return self._state

    @property
    def action(self):
        return self._action

    def _set(self, view, value, batch_index):
        value = torch.as_tensor(value, dtype=torch.float32, device=view.device)
        if batch_index is None:
            view.copy_(value.expand_as(view) if value.ndim < view.ndim or value.shape[0] != view.shape[0] else value)
        else:
            view[batch_index] = value

    def set_pos(self, pos, batch_index=None):
        self._set(self.state.pos, pos, batch_index)

    def set_rot(self, rot, batch_index=None):
        self._set(self.state.rot, rot, batch_index)

    def set_vel(self, vel, batch_index=None):
        self._set(self.state.vel, vel, batch_index)

    def set_speed(self, speed, batch_index=None):
        self._set(self.state.speed, speed, batch_index)

    def set_steering(self, steering, batch_index=None):
        self._set(self.state.steering, steering, batch_index)

    def set_sideslip_angle(self, sideslip_angle, batch_index=None):
        self._set(self.state.sideslip_angle, sideslip_angle, batch_index)


class WorldCustom(_WorldBase):
    """``WorldCustom(World)`` of helper_training.py:791-861: ``step()`` is ONE fused HIP launch over agents x envs."""

    batch_dim = device = dt = x_semidim = y_semidim = parameters = dim_c = dim_p = None  # instance attributes of the mirror (properties on vmas' World)

    def __init__(self, env: SigmaEnv, dt: float, x_semidim, y_semidim):
        self._env = env
        self.batch_dim = self._batch_dim = env.B
        self.device = self._device = env.device
        self.dt = dt
        self.x_semidim, self.y_semidim = x_semidim, y_semidim
        self._agents = []
        self.parameters = None
        self._actions = torch.zeros((env.B, env.N, 2), dtype=torch.float32, device=env.device)
        self._scenario = None
        self._clamped_views = None
        self.dim_c, self.dim_p = 0, 2  # no communication channel, planar positions (vmas World defaults; Environment._set_action reads dim_c)

    def to(self, device):
        """``TorchVectorizedObject.to``: the env shard lives on the GPU it was created on; moving it is not possible."""
        if torch.device(device) != self.device and not (torch.device(device).type == "cuda" and torch.device(device).index in (None, self.device.index)):
            raise RuntimeError(f"the environment lives on {self.device}; create the scenario on the target device instead")
        for a in self._agents:
            a.to(self.device)

    @property
    def scripted_agents(self):
        return []

    def zero_grad(self):
        return

    @property
    def agents(self):
        return self._agents

    @property
    def entities(self):
        return self._agents

    @property
    def policy_agents(self):
        return self._agents

    def add_agent(self, agent):
        self._agents.append(agent)

    def reset(self, env_index):
        """VMAS zeroes the entity states here; the full reset (incl. that) happens in ``reset_world_at``."""
        return

    def step(self):
        us = [a.action.u for a in self._agents]
        if any(u is None for u in us):
            raise ValueError("agent.action.u is None: VMAS sets the actions before world.step()")
        torch.stack(us, dim=1, out=self._actions)  # ONE launch for the [B, N, 2] action block (N copies cost the host 17 % of a surface step)
        self._env.step(self._actions)
        if self._clamped_views is None:  # WorldCustom.step clamps entity.action.u in place (helper_training.py:807-818): per-agent views of the clamped block, built once
            clamped = self._env.buffer(capi.BUF_ACTION)
            self._clamped_views = [clamped[:, i] for i in range(len(self._agents))]
        for a, v in zip(self._agents, self._clamped_views):
            a.action.u = v
        if self._scenario is not None:
            self._scenario._obs_dirty = False
            self._scenario._obs_served.clear()  # (a step's observations are new ones)
            self._scenario._auto_reset_done_this_step = False
            self._scenario._info_batch = None


class ScenarioRoadTraffic(BaseScenario):
    """Drop-in for ``sigmarl.scenarios.road_traffic.ScenarioRoadTraffic`` (callbacks :104-1635; rendering is out of scope)."""

    def make_world(self, batch_dim: int, device, **kwargs):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("sigmarl_amd.ScenarioRoadTraffic runs on an MI355X only (device='cuda:<i>'); there is no CPU fallback")
        if not hasattr(self, "parameters") or self.parameters is None:
            self.parameters = Parameters(
                n_agents=kwargs.pop("n_agents", 4), scenario_type=kwargs.get("scenario_type", "cpm_entire"),
                dt=kwargs.pop("dt", 0.05), is_obs_noise=kwargs.pop("is_obs_noise", True), is_apply_mask=kwargs.pop("is_apply_mask", False),
                is_use_mtv_distance=kwargs.pop("is_use_mtv_distance", False), is_testing_mode=kwargs.pop("is_testing_mode", False),
            )
        p = self.parameters
        if "n_agents" in kwargs:  # VmasEnv(..., n_agents=...) forwards it (mappo_cavs.py:170-177)
            p.n_agents = int(kwargs.pop("n_agents"))
        make_world_scenario_type = kwargs.pop("scenario_type", "cpm_entire")  # the reference's normaliser quirk, see params.make_config
        self.map = load_map(p.scenario_type)
        if abs(float(p.lane_width) - self.map.parser_lane_width) > 1e-12 and "cpm" not in p.scenario_type:
            raise NotImplementedError(
                f"map table of {p.scenario_type!r} was parsed with lane_width={self.map.parser_lane_width}; Parameters.lane_width={p.lane_width} needs the map compiler (SURVEY.md 8f-2)")
        cfg = make_config(p, self.map, batch_dim, make_world_scenario_type)
        self.env = SigmaEnv(cfg=cfg, map_table=self.map, device=device)
        if p.scenario_type == "cpm_mixed":  # the sub-scenario distribution of the device-side resets (world_state_rt_sim.py:330-343)
            self.env.set_scenario_lists(list(p.cpm_scenario_probabilities))
        self.n_agents = p.n_agents
        self.agent_width, self.agent_length = AGENTS["width"], AGENTS["length"]
        self.max_speed = AGENTS["max_speed"]
        self.max_steering = torch.tensor(AGENTS["max_steering"], device=device, dtype=torch.float32)
        self.world_x_dim, self.world_y_dim = self.map.world_x_dim, self.map.world_y_dim
        self.i_iter = 0
        env = self.env
        world = WorldCustom(env, p.dt, torch.tensor(self.world_x_dim, device=device), torch.tensor(self.world_y_dim, device=device))
        world.parameters = p
        world._scenario = self
        state = env.buffer(capi.BUF_STATE)
        for i in range(self.n_agents):
            world.add_agent(Vehicle(
                name=f"agent_{i}", state_row=state[:, i], shape=Box(AGENTS["length"], AGENTS["width"]),
                u_range=[self.max_speed, self.max_steering], max_speed=self.max_speed, dynamics=KinematicBicycleModel(device=device)))
        self._world = world
        self._build_views()
        self._obs_dirty = True
        self._obs_served = set()  # agents whose current observation has been handed out (observation(): a second request draws new sensor noise)
        # device_side_resets: serve BOTH the per-agent reset requests and the resets of finished envs inside done() with the
        # device sampler (sigmaenv_auto_reset) instead of the reference's host loops over torch's generator.  The callbacks'
        # results are the same tensors; only the random stream differs (distributional parity).  Off by default.
        self.device_side_resets = bool(kwargs.pop("device_side_resets", getattr(self, "device_side_resets", False)))
        self._auto_reset_done_this_step = False
        self.stored_observations = [None] * self.n_agents
        self._info_buf, self._info_batch = None, None
        self._info_cache, self._info_cache_empty = [None] * self.n_agents, [None] * self.n_agents
        self._reward_views = None
        self._lanelet_table = torch.zeros((self.map.n_paths, max(1, self.map.n_lanelets_all)), dtype=torch.int32, device=device)
        ids = torch.as_tensor(self.map.lanelet_ids)
        self._lanelet_table[:, : ids.shape[1]] = ids.to(device)
        return world

    # ---- views mirroring the attribute tree consumers read (helper_training.py:596-642, cbf_qp.py:1286-1395) --------------
    def _build_views(self):
        e, B, N, dev = self.env, self.env.B, self.env.N, self.env.device
        cl = e.buffer(capi.BUF_CLOSEST)
        pth = e.buffer(capi.BUF_PATH)
        cf = e.buffer(capi.BUF_COL_FLAGS)
        distances = SimpleNamespace(
            type="mtv" if self.parameters.is_use_mtv_distance else "c2c", agents=e.buffer(capi.BUF_DIST_AGENTS),
            left_boundaries=e.buffer(capi.BUF_DIST_LEFT), right_boundaries=e.buffer(capi.BUF_DIST_RIGHT),
            boundaries=e.buffer(capi.BUF_DIST_BOUND), ref_paths=e.buffer(capi.BUF_DIST_REF),
            closest_point_on_ref_path=cl[..., 0], closest_point_on_left_b=cl[..., 1], closest_point_on_right_b=cl[..., 2])
        collisions = SimpleNamespace(
            with_agents=e.buffer(capi.BUF_COL_AGENTS), with_lanelets=cf[..., 0], with_entry_segments=cf[..., 1], with_exit_segments=cf[..., 2])
        ref_paths = SimpleNamespace(short_term=e.buffer(capi.BUF_SHORT_TERM), scenario_id=pth[..., 1], path_id=pth[..., 2], point_id=pth[..., 3])
        act = e.buffer(capi.BUF_ACTION)
        nom = e.buffer(capi.BUF_CBF_NOMINAL)
        # with a CBF controller in the loop (training or testing) world_state.nominal_action_* is what CBFQP left behind
        # (cbf_qp.py:1315-1379); without one info() mirrors the applied action (road_traffic.py:1522-1540)
        pp = self.parameters
        qp = bool((pp.is_using_cbf_training or pp.is_using_cbf_testing) and pp.is_solve_qp)
        self.world_state = SimpleNamespace(
            distances=distances, collisions=collisions, ref_paths_agent_related=ref_paths, vertices=e.buffer(capi.BUF_VERTICES), world=self._world,
            nominal_action_vel=(nom if qp else act)[..., 0], nominal_action_steer=(nom if qp else act)[..., 1],
            applied_action_vel=act[..., 0], applied_action_steer=act[..., 1])
        self.observation_provider = SimpleNamespace(observations=SimpleNamespace(nearing_agents_indices=e.buffer(capi.BUF_NEARING)))
        ri = e.buffer(capi.BUF_REWARD_INFO)
        self.reward_info = SimpleNamespace(**{name: ri[k] for k, name in enumerate(capi.REWARD_INFO_FIELDS)})
        tm = e.buffer(capi.BUF_TIMER)
        self.timer = SimpleNamespace(step=tm[:, 0], start=time.time(), end=0)
        self.num_task_tries, self.task_success_times = tm[:, 1], tm[:, 2]
        lw3 = float(self.env.cfg.lane_width) * 3
        self.normalizers = SimpleNamespace(
            pos=torch.tensor([self.agent_length * 10] * 2, device=dev), pos_world=torch.tensor([self.world_x_dim, self.world_y_dim], device=dev, dtype=torch.float32),
            v=torch.tensor(self.max_speed, device=dev), rot=torch.tensor(2 * math.pi, device=dev), steering=self.max_steering,
            distance_lanelet=torch.tensor(lw3, device=dev), distance_ref=torch.tensor(lw3, device=dev), distance_agent=torch.tensor(self.agent_length * 10, device=dev))
        self.constants = SimpleNamespace(
            empty_action_vel=torch.zeros((B, N), device=dev), empty_action_steering=torch.zeros((B, N), device=dev),
            reset_agent_min_distance=torch.tensor(self.agent_length ** 2 + self.agent_width ** 2, dtype=torch.float32).sqrt() * 1.5)

    # ---- reset (road_traffic.py:816-923; sampling restated from world_state_rt_sim.py:143-358, RNG = torch's global generator) --
    def _scenario_lists(self, env_index, agent_index):
        p = self.parameters
        if p.scenario_type != "cpm_mixed":
            return 0
        if agent_index is not None:
            return int(self.env.buffer(capi.BUF_PATH)[env_index, agent_index, 1].item())
        probs = torch.tensor(p.cpm_scenario_probabilities, dtype=torch.float32)
        return int(torch.multinomial(probs, 1, replacement=True).item()) + 1

    def _sample_start(self, positions: np.ndarray, agent_index: int, list_id: int, single: bool):
        """One agent's (path_id, point_id): the rejection loop of world_state_rt_sim.py:215-311 (draw order preserved)."""
        mp, p = self.map, self.parameters
        first, count = mp.list_first[list_id], mp.list_count[list_id]
        min_d2 = float(self.constants.reset_agent_min_distance) ** 2
        random_count, end_point_idx = 0, 3
        while True:
            random_count += 1
            path_id = int(torch.randint(0, count, (1,)).item())
            gp = first + path_id
            num_points = int(mp.n_center[gp])
            if p.is_testing_mode:
                end_point_idx += random_count
            else:
                end_point_idx = int(num_points / 2)
            end_point_idx = min(end_point_idx, int(num_points / 2))
            point_id = int(torch.randint(3, end_point_idx, (1,)).item())
            pos = mp.center[gp, point_id]
            positions[agent_index] = pos
            if not single and agent_index == 0:
                return gp, path_id, point_id
            others = positions[: agent_index + 1] if not single else positions
            d2 = ((positions[agent_index] - others) ** 2).sum(-1).astype(np.float32)
            d2[agent_index] = d2.max() + 1
            if d2.min() >= min_d2:
                return gp, path_id, point_id

    def reset_world_at(self, env_index: Optional[int] = None, agent_index: Optional[int] = None):
        p, env, mp, N = self.parameters, self.env, self.map, self.n_agents
        self._info_batch = None
        if agent_index is not None:
            assert env_index is not None
            agent_index = int(agent_index)
        if env_index is not None and agent_index is None and self._auto_reset_done_this_step:
            return  # device_side_resets: done() already reset every finished env of this step on the GPU
        if env_index is None:
            if p.predefined_ref_path_idx is None:
                if self.device_side_resets:
                    # vectorised initial reset by the device-side sampler (same rule, counter-based RNG instead of torch's generator; on cpm_mixed
                    # every env draws its sub-scenario from cpm_scenario_probabilities, SigmaEnv.set_scenario_lists)
                    env.buffer(capi.BUF_DONE).fill_(1)
                    env.auto_reset(seed=int(getattr(p, "random_seed", 0)))
                    self._obs_dirty = False
                    return
                if p.scenario_type == "cpm_mixed":
                    for e in range(env.B):
                        self.reset_world_at(e)
                    return
                # default: the reference's own loop over the envs (road_traffic.py:832-834) drawing from torch's global generator in the
                # reference's order, so a caller that seeds torch gets the reference's initial states
            envs = range(env.B)
        else:
            envs = [int(env_index)]
        env_idx, agent_idx, ids, st8 = [], [], [], []
        state_host = env.buffer(capi.BUF_STATE).cpu().numpy() if agent_index is not None else None
        for e in envs:
            list_id = self._scenario_lists(e, agent_index)
            positions = np.zeros((N, 2), np.float32) if agent_index is None else state_host[e, :, 0:2].copy()
            for i in (range(N) if agent_index is None else [agent_index]):
                if agent_index is None and p.predefined_ref_path_idx is not None:  # deterministic start (world_state_rt_sim.py:99-126)
                    path_id = int(p.predefined_ref_path_idx[i])
                    gp = mp.list_first[list_id] + path_id
                    x, y, rot = [float(v) for v in p.init_state[i][0:3]]
                    point_id, speed = 0, 0.0
                else:
                    gp, path_id, point_id = self._sample_start(positions, i, list_id, agent_index is not None)
                    x, y = [float(v) for v in mp.center[gp, point_id]]
                    rot = float(mp.yaw[gp, min(point_id, int(mp.n_yaw[gp]) - 1)])
                    speed = float(torch.rand(1, dtype=torch.float32).item() * self.max_speed)
                rot32, sp32 = np.float32(rot), np.float32(speed)
                vx = np.float32(sp32 * np.float32(math.cos(float(rot32))))
                vy = np.float32(sp32 * np.float32(math.sin(float(rot32))))
                env_idx.append(e); agent_idx.append(i)
                ids.append((gp, list_id, path_id, point_id))
                st8.append((x, y, rot32, sp32, 0.0, vx, vy, 0.0))
        env.reset(env_idx, agent_idx, np.asarray(ids, np.int32), np.asarray(st8, np.float32), full_env=agent_index is None)
        self._obs_dirty = True
        self._obs_served.clear()

    # ---- callbacks -------------------------------------------------------------------------------------------------
    def _index(self, agent) -> int:
        i = getattr(agent, "_sigma_index", None)
        if i is None:
            i = agent._sigma_index = self.world.agents.index(agent)
        return i

    def reward(self, agent):
        """[B] fp32, already computed by the fused step (road_traffic.py:925-1253)."""
        if self._reward_views is None:
            r = self.env.reward
            self._reward_views = [r[:, i] for i in range(self.n_agents)]
        return self._reward_views[self._index(agent)]

    def observation(self, agent):
        """[B, obs_dim] fp32 (road_traffic.py:1334-1366); uniform noise as observation_provider_rt.py:613-618 when enabled (device side).

        The noise is drawn ON THE DEVICE from the counter-based generator keyed on (Parameters.random_seed, the env's episodes_reset and timer.step, env, agent,
        column) -- a pure function of the env's own counters, so the fused / separate / T-step launches, any sharding and the rollout record all see the same
        values -- plus, for an observation that is requested AGAIN without a step or reset in between, the number of that re-observation on the handle: like the
        reference's ``torch.rand_like`` per call (observation_provider_rt.py:613-618) a second ``observation(agent)`` returns fresh noise.  What still differs:
        seeding torch does not control the noise -- ``Parameters.random_seed`` does."""
        i = self._index(agent)
        if self._obs_dirty:
            if i == 0:
                self.env.observe()
                self._obs_dirty = False
                self._obs_served.clear()
        elif i in self._obs_served and self.parameters.is_obs_noise:
            self.env.observe()  # taken again at the same counters: sigmaenv_observe salts the draws with its call count
            self._obs_served.clear()
        self._obs_served.add(i)
        obs = self.env.obs[:, i]  # incl. the sensor noise: added on the device (sigmaenv_config_t.obs_noise_level), so the rollout record and the
        self.stored_observations[i] = obs  # on-device actor see the same noisy observation the trainer does
        return obs

    def done(self):
        """[B] bool (road_traffic.py:1368-1487) + the per-agent resets the reference performs here (:1435-1447, :1456-1473)."""
        is_done = self.env.done.to(torch.bool)
        self._info_batch = None
        if self.device_side_resets:
            # cpm_mixed: every finished env draws its sub-scenario (path list) from cpm_scenario_probabilities and keeps it for its per-agent resets
            # (world_state_rt_sim.py:313-358): SigmaEnv.default_paths() hands the device sampler the scenario lists instead of one path range
            self.env.auto_reset(seed=int(getattr(self.parameters, "random_seed", 0)))
            self._auto_reset_done_this_step = True
            self._obs_dirty = False  # the reset kernel refreshed the observations of every touched env
            self._obs_served.clear()
            return is_done
        if self.parameters.is_testing_mode or self.parameters.scenario_type != "cpm_entire":
            req = self.env.buffer(capi.BUF_COL_FLAGS)[..., 3]
            if bool(req.any()):
                idx = torch.nonzero(req).cpu().tolist()
                for e, a in idx:
                    self.reset_world_at(env_index=e, agent_index=a)
        return is_done

    def _info_refresh(self):
        """The derived entries of info() for ALL agents, written IN PLACE into buffers that live as long as the scenario (same ops, same values as one call per agent:
        they are elementwise in the agent).  The per-agent dicts of info() hold views of these buffers and of the library's, so they are built once."""
        ws, nz, B, N = self.world_state, self.normalizers, self.env.B, self.n_agents
        state = self.env.buffer(capi.BUF_STATE)
        ib = self._info_buf
        if ib is None:
            dev = self.env.device
            f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
            ib = self._info_buf = dict(
                pos_nom=f32(B, N, 2), rot=f32(B, N, 1), rot_nom=f32(B, N, 1), vel_nom=f32(B, N, 2), ref_nom=f32(B, N, ws.ref_paths_agent_related.short_term.shape[2], 2),
                distance_ref_nom=f32(B, N), dl=f32(B, N), dl_nom=f32(B, N), dr=f32(B, N), dr_nom=f32(B, N), act_v_nom=f32(B, N), act_s_nom=f32(B, N),
                col_agents=torch.empty((B, N), dtype=torch.bool, device=dev), col_lane=torch.empty((B, N), dtype=torch.bool, device=dev),
                goal=torch.empty((B, N), dtype=torch.bool, device=dev), col_any=torch.empty((B, N, N), dtype=torch.bool, device=dev),
                lanelet_ids=torch.empty((B * N, self._lanelet_table.shape[1]), dtype=self._lanelet_table.dtype, device=dev),
                arg=torch.empty((B, N), dtype=torch.long, device=dev), wrap=torch.empty((B, N, 1), dtype=torch.bool, device=dev), rot_m=f32(B, N, 1))
        two_pi = 2 * math.pi
        torch.remainder(state[..., 2:3], two_pi, out=ib["rot"])
        torch.gt(ib["rot"], math.pi, out=ib["wrap"])
        torch.sub(ib["rot"], two_pi, out=ib["rot_m"])
        torch.where(ib["wrap"], ib["rot_m"], ib["rot"], out=ib["rot"])  # angle_eliminate_two_pi, helper_scenario.py:1276-1289
        torch.div(state[..., 0:2], nz.pos_world, out=ib["pos_nom"])
        torch.div(ib["rot"], nz.rot, out=ib["rot_nom"])
        torch.div(state[..., 5:7], nz.v, out=ib["vel_nom"])
        torch.div(ws.ref_paths_agent_related.short_term, nz.pos_world, out=ib["ref_nom"])
        torch.div(ws.distances.ref_paths, nz.distance_ref, out=ib["distance_ref_nom"])
        torch.min(ws.distances.left_boundaries, dim=-1, out=(ib["dl"], ib["arg"]))
        torch.min(ws.distances.right_boundaries, dim=-1, out=(ib["dr"], ib["arg"]))
        torch.div(ib["dl"], nz.distance_lanelet, out=ib["dl_nom"])
        torch.div(ib["dr"], nz.distance_lanelet, out=ib["dr_nom"])
        torch.ne(ws.collisions.with_agents, 0, out=ib["col_any"])
        torch.any(ib["col_any"], dim=-1, out=ib["col_agents"])
        torch.ne(ws.collisions.with_lanelets, 0, out=ib["col_lane"])
        torch.ne(ws.collisions.with_exit_segments, 0, out=ib["goal"])
        torch.index_select(self._lanelet_table, 0, self.env.buffer(capi.BUF_PATH)[..., 0].reshape(-1).long(), out=ib["lanelet_ids"])
        act = self.env.buffer(capi.BUF_ACTION)  # (agent.action.u after a step: the clamped action)
        torch.div(act[..., 0], nz.v, out=ib["act_v_nom"])
        torch.div(act[..., 1], nz.steering, out=ib["act_s_nom"])
        self._info_batch = ib

    def _info_views(self, i: int, empty: bool):
        ws, ib, B = self.world_state, self._info_buf, self.env.B
        st = self.world.agents[i].state
        act = self.env.buffer(capi.BUF_ACTION)
        act_v = self.constants.empty_action_vel[:, i] if empty else act[:, i, 0]
        act_s = self.constants.empty_action_steering[:, i] if empty else act[:, i, 1]
        info = {
            "pos": st.pos, "pos_nom": ib["pos_nom"][:, i], "rot": ib["rot"][:, i], "rot_nom": ib["rot_nom"][:, i],
            "vel": st.vel, "vel_nom": ib["vel_nom"][:, i],
            "act_vel": act_v, "act_vel_nom": act_v if empty else ib["act_v_nom"][:, i],
            "act_steer": act_s, "act_steer_nom": act_s if empty else ib["act_s_nom"][:, i],
            "ref": ws.ref_paths_agent_related.short_term[:, i].reshape(B, -1), "ref_nom": ib["ref_nom"][:, i].reshape(B, -1),
            "distance_ref": ws.distances.ref_paths[:, i], "distance_ref_nom": ib["distance_ref_nom"][:, i],
            "distance_left_b": ib["dl"][:, i], "distance_left_b_nom": ib["dl_nom"][:, i],
            "distance_right_b": ib["dr"][:, i], "distance_right_b_nom": ib["dr_nom"][:, i],
            "is_collision_with_agents": ib["col_agents"][:, i],
            "is_collision_with_lanelets": ib["col_lane"][:, i],
            "is_reach_goal": ib["goal"][:, i],
            "ref_lanelet_ids": ib["lanelet_ids"].view(B, self.n_agents, -1)[:, i], "path_id": ws.ref_paths_agent_related.path_id[:, i],
            "applied_action_vel": ws.applied_action_vel[:, i], "applied_action_steer": ws.applied_action_steer[:, i],
            "nominal_action_vel": ws.nominal_action_vel[:, i], "nominal_action_steer": ws.nominal_action_steer[:, i],
        }
        rew = {name: getattr(self.reward_info, name)[:, i] for name in capi.REWARD_INFO_FIELDS}
        return info, rew

    def info(self, agent) -> Dict[str, torch.Tensor]:
        """The 27 + 12 entries of road_traffic.py:1489-1635 (+ 2 with is_using_prioritized_marl).

        The derived entries are elementwise in the agent: they are computed for ALL agents in one batched op each, when the first agent asks after a step or reset
        (VMAS collects the infos of all agents back to back, tests/vmas_env_shim.py), IN PLACE into buffers that live as long as the scenario -- so every entry of
        every agent is a view that is built ONCE (624 view constructions per step cost the host 16 % of a surface step; the values are the same ops on the same
        inputs).  A consumer that keeps an info tensor across steps must clone it -- vmas' Environment does."""
        i = self._index(agent)
        empty = agent.action.u is None
        if getattr(self, "_info_batch", None) is None:
            self._info_refresh()
        cache = self._info_cache_empty if empty else self._info_cache
        if cache[i] is None:
            cache[i] = self._info_views(i, empty)
        info = dict(cache[i][0])
        cv = self.world._clamped_views
        if not empty and not (cv is not None and agent.action.u is cv[i]):
            # agent.action.u is not the clamped block the last step left behind (a caller assigned an action and asks before stepping): report what is there
            nz = self.normalizers
            act_v, act_s = agent.action.u[:, 0], agent.action.u[:, 1]
            info.update(act_vel=act_v, act_vel_nom=act_v / nz.v, act_steer=act_s, act_steer_nom=act_s / nz.steering)
        if getattr(self.parameters, "is_using_prioritized_marl", False):
            # the two extra entries of prioritised MARL (road_traffic.py:1513-1520, :1616-1625): the observation padded with the placeholders of the
            # neighbours' actions for the base policy, and the observation itself for the priority-assignment policy
            obs = self.stored_observations[i] if self.stored_observations[i] is not None else self.env.obs[:, i]
            info["base_observation"] = torch.nn.functional.pad(obs.clone(), (0, self.parameters.n_nearing_agents_observed * 2))
            info["priority_observation"] = obs.clone()
        info.update(cache[i][1])  # (the RewardInfo fields last, as the reference's dict orders them)
        return info


def make_scenario(parameters: Parameters) -> ScenarioRoadTraffic:
    """``scenario = ScenarioRoadTraffic(); scenario.parameters = parameters`` as mappo_cavs.py:168-169 does."""
    sc = ScenarioRoadTraffic()
    sc.parameters = parameters
    return sc
