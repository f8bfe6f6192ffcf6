"""``Parameters``: host-side mirror of the reference's configuration object.

Same field names, defaults and JSON round-trip as ``sigmarl/helper_common.py:26-287`` so that
``sigmarl/config.json`` and ``main_training.py``-style overrides load unchanged.  Only the fields the
environment-step path reads are interpreted here (``make_config``); the trainer-side fields are carried
verbatim for the consumer (``mappo_cavs.py``).
"""
from __future__ import annotations

import json

from . import capi

# (name, default) in the reference's keyword order
_FIELDS = [
    ("n_agents", 4), ("dt", 0.05), ("device", "cpu"), ("scenario_name", "road_traffic"),
    ("n_iters", 250), ("num_epochs", 30), ("minibatch_size", 512), ("lr", 2e-4), ("lr_min", 1e-5),
    ("max_grad_norm", 1.0), ("clip_epsilon", 0.2), ("gamma", 0.99), ("lmbda", 0.9), ("entropy_eps", 1e-4),
    ("max_steps", 128), ("num_vmas_envs", 32), ("scenario_type", "intersection_1"),
    ("episode_reward_mean_current", 0.00), ("episode_reward_intermediate", -1e3),
    ("is_prb", False), ("is_challenging_initial_state_buffer", False), ("cpm_scenario_probabilities", [1.0, 0.0, 0.0]),
    ("n_steps_stored", 10), ("n_points_short_term", 3), ("is_partial_observation", True),
    ("n_nearing_agents_observed", 2), ("is_ego_view", True), ("is_apply_mask", True),
    ("is_observe_distance_to_agents", True), ("is_observe_distance_to_boundaries", True),
    ("is_observe_distance_to_center_line", True), ("is_observe_vertices", True), ("is_obs_noise", True),
    ("obs_noise_level", 0.05), ("is_observe_ref_path_other_agents", False), ("is_use_mtv_distance", True),
    ("is_visualize_short_term_path", True), ("is_visualize_lane_boundary", False), ("is_real_time_rendering", False),
    ("is_visualize_extra_info", True), ("render_title", ""), ("is_save_intermediate_model", True),
    ("is_load_model", False), ("is_load_final_model", False), ("model_name", None), ("where_to_save", "outputs/"),
    ("is_continue_train", False), ("is_save_eval_results", True), ("is_load_out_td", False),
    ("is_testing_mode", False), ("is_save_simulation_video", False), ("is_using_opponent_modeling", False),
    ("is_using_prioritized_marl", False), ("prioritization_method", "marl"), ("is_communication_noise", False),
    ("communication_noise_level", 0.1), ("is_using_cbf_testing", False), ("is_using_cbf_training", False),
    ("is_using_centralized_cbf", False), ("is_apply_cbf_action", False), ("is_solve_qp", True),
    ("experiment_type", "simulation"), ("is_obs_steering", False), ("predefined_ref_path_idx", None),
    ("init_state", None), ("random_seed", 0), ("is_using_pseudo_distance", False),
    ("n_circles_approximate_vehicle", 3), ("lane_width", 0.25), ("reset_agent_fixed_duration", 0),
    ("is_grouping_agents", False), ("max_group_size", 2), ("observation_range", 0.5), ("nom_controller_type", "rl"),
    ("adaptive_lambda", False), ("rs", 0.5), ("h_nom", 0.2), ("rew_method", "distance"), ("reward_progress", 0.10),
    ("threshold_near_boundary_high", 0.02), ("threshold_near_boundary_low", 0),
    ("threshold_near_other_agents_c2c_high", 0.3), ("threshold_near_other_agents_c2c_low", 0),
    ("ttc_low", 0), ("ttc_high", 3.75), ("penalty_near_boundary", -0.2), ("penalty_near_other_agents", -0.2),
]
_DEFAULTS = dict(_FIELDS)


class Parameters:
    """Keyword-configured parameter bag (reference: ``sigmarl/helper_common.py:26-256``)."""

    def __init__(self, **kwargs):
        unknown = set(kwargs) - set(_DEFAULTS)
        if unknown:
            raise TypeError(f"Parameters got unexpected keyword(s): {sorted(unknown)}")
        for name, default in _FIELDS:
            value = kwargs.get(name, default)
            if isinstance(default, list) and name not in kwargs:
                value = list(default)
            setattr(self, name, value)
        if self.model_name is None and self.scenario_name is not None:
            self.model_name = f"reward{self.episode_reward_mean_current:.2f}"  # helper_common.py:20-23,254-255

    @property
    def frames_per_batch(self):
        return self.num_vmas_envs * self.max_steps

    @property
    def total_frames(self):
        return self.frames_per_batch * self.n_iters

    def to_dict(self):
        return self.__dict__

    @classmethod
    def from_dict(cls, d):
        return cls(**d)

    @classmethod
    def from_json(cls, path):
        with open(path, "r") as f:
            return cls(**json.load(f))


def obs_flags(p: Parameters) -> int:
    """``sigmaenv_config_t.obs_flags`` of the observation switches (observation_provider_rt.py:803-925)."""
    f = 0
    if p.is_obs_steering:
        f |= capi.OBS_STEERING
    if p.is_observe_ref_path_other_agents:
        f |= capi.OBS_REF_OTHERS
    if not p.is_observe_vertices:
        f |= capi.OBS_NO_VERTICES
    if not p.is_observe_distance_to_agents:
        f |= capi.OBS_NO_DIST_AGENTS
    if not p.is_observe_distance_to_center_line:
        f |= capi.OBS_NO_DIST_CENTER
    if not p.is_ego_view:
        f |= capi.OBS_BIRD_VIEW
    if not p.is_observe_distance_to_boundaries:
        f |= capi.OBS_BOUNDARY_POINTS
    if getattr(p, "is_using_opponent_modeling", False):
        f |= capi.OBS_OPPONENT_PAD
    if not p.is_partial_observation:
        f |= capi.OBS_FULL
    return f


def check_supported(p: Parameters) -> None:
    """Raise for observation/feature flags the fused step does not implement (fail loudly, never silently differ)."""
    bad = []
    # bird view with is_apply_mask (the lanelet-relation mask, observation_provider_rt.py:577-665 / map_manager.py:41-118) IS built: SigmaEnv hands the
    # map's lanelet tables to the library (sigmaenv_set_lanelets); maps without a neighbour table (the CPM map) mask by distance only, as in the reference
    # is_partial_observation=False (every agent observes ALL agents, observation_provider_rt.py:756-800) IS built for the bird view (capi.OBS_FULL); in the
    # ego view the reference itself raises (its reshape of the per-agent length / width scalars to [batch, n_nearing, -1] fails), and so does this mirror
    if not p.is_partial_observation:
        if p.is_ego_view:
            bad.append("is_partial_observation=False with is_ego_view=True (raises in the reference as well)")
        else:
            try:
                capi.obs_dim(min(int(p.n_nearing_agents_observed), int(p.n_agents) - 1), obs_flags(p), int(p.n_points_short_term), int(p.n_agents))
            except ValueError as exc:
                bad.append(str(exc))
    # is_apply_mask: in ego view (the only view built) only the DISTANCE criterion is live in the reference -- the lanelet of every agent
    # (MapManager.determine_current_lanelet) is only computed in the bird-view branch of update_state (observation_provider_rt.py:537-588),
    # so current_lanelet_idx stays empty and determine_masked_agents_by_lanelets masks nobody (map_manager.py:21,102-118) on every map
    # built through the separate observation kernel (capi.OBS_*): is_obs_steering, is_observe_ref_path_other_agents, is_observe_vertices=False,
    # is_observe_distance_to_agents=False, is_observe_distance_to_center_line=False
    # n_points_short_term is a build constant of the library: every value in 1 .. 8 has its own build (capi.variant_path; `make NS=k`)
    if not (1 <= int(p.n_points_short_term) <= capi.MAX_SHORT_TERM):
        bad.append(f"n_points_short_term={p.n_points_short_term} (1 .. {capi.MAX_SHORT_TERM})")
    if p.is_challenging_initial_state_buffer:
        bad.append("is_challenging_initial_state_buffer=True")
    # is_using_opponent_modeling IS built: the placeholder columns (observation_provider_rt.py:606-611, capi.OBS_OPPONENT_PAD) and the gather of the
    # neighbours' tentative actions into them (SigmaEnv.opponent_fill; helper_training.py:1117-1137).
    # is_using_prioritized_marl: the environment's share is two extra info() entries (road_traffic.py:1513-1520, :1616-1625), built in scenario.info();
    # priority assignment and action propagation are the trainer's.  is_using_pseudo_distance is read nowhere in the reference outside Parameters
    # (helper_common.py:117, :229): accepted, no effect -- as there.
    if p.is_using_cbf_training or p.is_using_cbf_testing or "cbf" in p.rew_method:
        # built: the centralized QP (is_solve_qp=True), the grouped QPs (is_grouping_agents) and the QP-free margin reward
        # (sigmarl/cbf_qp.py:2534-2560).  The reference's grouped update only runs with is_solve_qp=True (its coefficient builders receive
        # lam=None and raise otherwise, :1993-1998 with :2371-2398) and without a cap on the cross-group neighbours.
        if p.is_grouping_agents and not p.is_solve_qp:
            bad.append("is_grouping_agents=True with is_solve_qp=False (raises in the reference as well)")
        if p.is_grouping_agents and getattr(p, "max_cross_neighbors", None) is not None:
            bad.append("max_cross_neighbors")
        if p.is_grouping_agents and int(p.max_group_size) < 1:
            bad.append("max_group_size < 1")
        if p.nom_controller_type not in ("rl", "clf"):
            bad.append(f"nom_controller_type={p.nom_controller_type!r}")
        if not 1 <= int(p.n_circles_approximate_vehicle) <= capi.CBF_MAX_CIRCLES:
            bad.append(f"n_circles_approximate_vehicle={p.n_circles_approximate_vehicle}")
    if bad:
        raise NotImplementedError("sigmarl_amd: unsupported configuration for the fused environment step: " + "; ".join(bad))


# ``_init_params`` reads the lane width for its NORMALISERS from ``SCENARIOS[kwargs.pop("scenario_type", "cpm_entire")]``
# (road_traffic.py:116-123).  ``mappo_cavs.py:170-177`` passes no ``scenario_type`` kwarg to VmasEnv, so the normaliser
# ``distance_lanelet`` is 3 * 0.15 for EVERY map when driven through the trainer; pass ``make_world_scenario_type`` to
# mirror a caller that does forward the kwarg.
CPM_LANE_WIDTH = 0.15


def make_config(p: Parameters, map_table, n_envs: int, make_world_scenario_type: str = "cpm_entire", env_index_base: int = 0) -> capi.Config:
    """Thresholds/penalties exactly as ``ScenarioRoadTraffic._init_params`` derives them when ``scenario.parameters``
    is pre-set (``sigmarl/scenarios/road_traffic.py:132-175,214-270``)."""
    check_supported(p)
    A = capi.AGENTS
    r_p_normalizer = 100
    mtv = bool(p.is_use_mtv_distance)
    n_agents = int(p.n_agents)
    c = capi.Config()
    c.abi_version = capi.ABI_VERSION
    c.n_envs = int(n_envs)
    c.env_index_base = int(env_index_base)  # this handle's first env in the whole batch (shard_range begin): keys the device-side random draws
    c.n_agents = n_agents
    c.distance_type = capi.DIST_MTV if mtv else capi.DIST_C2C
    c.rew_flags = capi.rew_flags_from_method(p.rew_method, bool(p.is_solve_qp))
    c.is_testing_mode = int(bool(p.is_testing_mode))
    c.has_entry_exit = int(p.scenario_type != "cpm_entire")
    c.max_steps = int(p.max_steps)
    c.n_nearing = min(int(p.n_nearing_agents_observed), n_agents - 1)
    c.dt = p.dt
    c.length, c.width, c.l_f, c.l_r = A["length"], A["width"], A["l_f"], A["l_r"]
    c.max_speed, c.max_steering = A["max_speed"], A["max_steering"]
    c.min_acc, c.max_acc = A["min_acc"], A["max_acc"]
    c.min_steering_rate, c.max_steering_rate = A["min_steering_rate"], A["max_steering_rate"]
    c.world_x_dim, c.world_y_dim = map_table.world_x_dim, map_table.world_y_dim
    if make_world_scenario_type in ("cpm_entire", "cpm_mixed"):
        c.lane_width = CPM_LANE_WIDTH
    else:
        from .maps import load_map

        c.lane_width = load_map(make_world_scenario_type).lane_width
    c.reward_progress = p.reward_progress if p.reward_progress is not None else 0.1
    c.reward_reach_goal = 100 / r_p_normalizer
    c.penalty_near_boundary = p.penalty_near_boundary if p.penalty_near_boundary is not None else -0.2
    c.penalty_near_other_agents = p.penalty_near_other_agents if p.penalty_near_other_agents is not None else -0.2
    c.penalty_collide_with_agents = -100 / r_p_normalizer
    c.penalty_collide_with_boundaries = -100 / r_p_normalizer
    c.threshold_near_boundary_low = p.threshold_near_boundary_low if p.threshold_near_boundary_low is not None else 0
    c.threshold_near_boundary_high = p.threshold_near_boundary_high if p.threshold_near_boundary_high is not None else 0.02
    if mtv:  # road_traffic.py:264-270,632-649
        c.threshold_near_other_agents_low = 0
        c.threshold_near_other_agents_high = A["length"]
    else:
        c.threshold_near_other_agents_low = p.threshold_near_other_agents_c2c_low if p.threshold_near_other_agents_c2c_low is not None else 0
        c.threshold_near_other_agents_high = p.threshold_near_other_agents_c2c_high if p.threshold_near_other_agents_c2c_high is not None else 0.3
    c.is_apply_mask = int(bool(p.is_apply_mask))
    c.distance_mask_agents = A["length"] * 5  # road_traffic.py:663
    c.reset_agent_fixed_duration = float(p.reset_agent_fixed_duration or 0.0)  # road_traffic.py:1388-1397
    c.obs_flags = obs_flags(p)
    c.n_points_short_term = int(p.n_points_short_term)
    # observation noise (observation_provider_rt.py:613-618) is added on the device, from the counter-based generator seeded by Parameters.random_seed
    c.obs_noise_level = float(p.obs_noise_level) if p.is_obs_noise else 0.0
    c.obs_noise_seed_lo, c.obs_noise_seed_hi = int(p.random_seed or 0) & 0xFFFFFFFF, (int(p.random_seed or 0) >> 32) & 0xFFFFFFFF
    c.penalty_deviate_from_cbf_vel = c.penalty_deviate_from_cbf_steer = -5 / r_p_normalizer  # road_traffic.py:238-243
    c.ttc_low = p.ttc_low if p.ttc_low is not None else 0
    c.ttc_high = p.ttc_high if p.ttc_high is not None else 3.75
    return c
