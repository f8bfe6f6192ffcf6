"""ctypes mirror of ``include/sigmaenv.h`` (the C-ABI of the HIP environment step).

The structures here are the single Python-side definition of the ABI; the product
binds ``libsigmaenv.so`` with them and the tests bind the CPU oracle
(``oracle/libsigmaenv_oracle.so``, prefix ``sigmaenv_oracle_``) with the very same
definitions, so both sides are driven through identical calls.

The product path fails loudly when the HIP library is missing: there is no CPU
fallback (``load_library`` raises).
"""
from __future__ import annotations

import ctypes as C
import math
import os

ABI_VERSION = 5
DIST_C2C, DIST_MTV = 0, 1
REW_DISTANCE, REW_TTC, REW_EXACT_SPARSE, REW_HAS_SPARSE, REW_CBF, REW_CBF_QP = 1, 2, 4, 8, 16, 32
CBF_MAX_CIRCLES = 4
N_SHORT_TERM = 3  # the default build; MAX_SHORT_TERM: the largest n_points_short_term a build exists for (include/sigmaenv.h)
MAX_SHORT_TERM = 8
MAX_NEARING = 4
N_REWARD_INFO = 12

(BUF_STATE, BUF_PREV_POS, BUF_VERTICES, BUF_PATH, BUF_SHORT_TERM, BUF_DIST_REF, BUF_DIST_LEFT, BUF_DIST_RIGHT,
 BUF_DIST_BOUND, BUF_CLOSEST, BUF_DIST_AGENTS, BUF_COL_AGENTS, BUF_COL_FLAGS, BUF_REWARD, BUF_REWARD_INFO, BUF_OBS,
 BUF_NEARING, BUF_DONE, BUF_TIMER, BUF_ACTION, BUF_CBF_NOMINAL) = range(21)

REWARD_INFO_FIELDS = (
    "rew_progress", "rew_reach_goal", "rew_speed", "rew_centerline", "rew_near_other_agents", "rew_near_left_lane",
    "rew_near_right_lane", "rew_collide_other_agents", "rew_collide_lane", "rew_energy_acceleration",
    "rew_energy_steering", "rew_total",
)  # RewardInfo declaration order, sigmarl/helper_scenario.py:101-114


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("n_envs", C.c_int32), ("n_agents", C.c_int32), ("distance_type", C.c_int32),
        ("rew_flags", C.c_int32), ("is_testing_mode", C.c_int32), ("has_entry_exit", C.c_int32), ("max_steps", C.c_int32),
        ("n_nearing", C.c_int32), ("envs_per_group", C.c_int32),
        ("dt", C.c_float), ("length", C.c_float), ("width", C.c_float), ("l_f", C.c_float), ("l_r", C.c_float),
        ("max_speed", C.c_float), ("max_steering", C.c_float), ("min_acc", C.c_float), ("max_acc", C.c_float),
        ("min_steering_rate", C.c_float), ("max_steering_rate", C.c_float),
        ("world_x_dim", C.c_float), ("world_y_dim", C.c_float), ("lane_width", C.c_float),
        ("reward_progress", C.c_float), ("reward_reach_goal", C.c_float),
        ("penalty_near_boundary", C.c_float), ("penalty_near_other_agents", C.c_float),
        ("penalty_collide_with_agents", C.c_float), ("penalty_collide_with_boundaries", C.c_float),
        ("threshold_near_boundary_low", C.c_float), ("threshold_near_boundary_high", C.c_float),
        ("threshold_near_other_agents_low", C.c_float), ("threshold_near_other_agents_high", C.c_float),
        ("ttc_low", C.c_float), ("ttc_high", C.c_float),
        ("penalty_deviate_from_cbf_vel", C.c_float), ("penalty_deviate_from_cbf_steer", C.c_float),
        ("is_apply_mask", C.c_int32), ("distance_mask_agents", C.c_float), ("obs_flags", C.c_int32), ("reset_agent_fixed_duration", C.c_float),
        ("env_index_base", C.c_int32), ("obs_noise_level", C.c_float), ("obs_noise_seed_lo", C.c_uint32), ("obs_noise_seed_hi", C.c_uint32),
        ("n_points_short_term", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


class Map(C.Structure):
    _fields_ = [
        ("n_paths", C.c_int32), ("stride_points", C.c_int32),
        ("center", C.c_void_p), ("left", C.c_void_p), ("right", C.c_void_p), ("yaw", C.c_void_p),
        ("n_center", C.c_void_p), ("n_left", C.c_void_p), ("n_right", C.c_void_p), ("is_loop", C.c_void_p),
    ]


class CbfConfig(C.Structure):
    """``sigmaenv_cbf_config_t``."""

    _fields_ = [
        ("n_circles", C.c_int32), ("nominal", C.c_int32),
        ("dt_taylor", C.c_double), ("lambda_ttcbf", C.c_double), ("h_nom", C.c_double), ("fd_step", C.c_double),
        ("safety_buffer", C.c_double), ("circle_radius", C.c_double), ("circle_x", C.c_double * CBF_MAX_CIRCLES),
        ("l_r", C.c_double), ("l_wb", C.c_double), ("min_speed", C.c_float), ("min_steering", C.c_float),
        ("steering_rate_max", C.c_double), ("k_clf_speed", C.c_double), ("k_clf_heading", C.c_double), ("ref_speed", C.c_double),
        ("qp_w_acc", C.c_double), ("qp_w_steer", C.c_double), ("qp_w_lane", C.c_double), ("qp_w_pair", C.c_double), ("qp_w_clf", C.c_double),
        ("qp_w_lambda", C.c_double), ("lam_clf", C.c_double), ("is_apply_cbf_action", C.c_int32), ("is_grouping", C.c_int32),
        ("max_group_size", C.c_int32), ("reserved4", C.c_int32), ("observation_range", C.c_double), ("rs", C.c_double),
        ("qp_w_cross", C.c_double), ("qp_w_lambda_cross", C.c_double),
    ]


# vehicle constants of the reference, sigmarl/constants.py:628-647
AGENTS = {
    "width": 0.107, "length": 0.22, "l_f": 0.075, "l_r": 0.075, "l_wb": 0.15,
    "max_speed": 1.0, "min_speed": -0.5,
    "max_steering": 31 * math.pi / 180, "min_steering": -31 * math.pi / 180,
    "max_acc": 5.0, "min_acc": -5.0, "max_steering_rate": math.pi / 2, "min_steering_rate": -math.pi / 2,
    "n_actions": 2,
}


def rew_flags_from_method(rew_method: str, is_solve_qp: bool = True) -> int:
    """Decode ``Parameters.rew_method`` the way sigmarl/scenarios/road_traffic.py:1056-1151 tests the string."""
    f = 0
    if "cbf" in rew_method:
        f |= REW_CBF_QP if is_solve_qp else REW_CBF
    if "distance" in rew_method:
        f |= REW_DISTANCE
    if "ttc" in rew_method:
        f |= REW_TTC
    if rew_method == "sparse":
        f |= REW_EXACT_SPARSE
    if "sparse" in rew_method:
        f |= REW_HAS_SPARSE
    return f


KERNEL_STEP, KERNEL_CBF_QP, KERNEL_CBF_MARGIN, KERNEL_MLP32, KERNEL_ACTOR_BF16 = range(5)
MLP32_EXACT, MLP32_SPLIT = 0, 1  # sigmaenv_mlp32_set_mode
KERNEL_NAMES = ("sigmaenv_step_wave_kernel", "cbf::sigmaenv_cbf_qp_kernel", "cbf::sigmaenv_cbf_kernel", "sigmaenv_mlp32_kernel", "sigmaenv_actor_kernel")
SCENARIO_LISTS = -1  # path_count of a device-side reset that draws from the handle's sub-scenario lists (cpm_mixed)
OBS_STEERING, OBS_REF_OTHERS, OBS_NO_VERTICES, OBS_NO_DIST_AGENTS, OBS_NO_DIST_CENTER, OBS_BIRD_VIEW, OBS_BOUNDARY_POINTS, OBS_OPPONENT_PAD = 1, 2, 4, 8, 16, 32, 64, 128
OBS_FULL = 256  # Parameters.is_partial_observation == False (bird view only): every agent observes ALL agents (include/sigmaenv.h)


def obs_dim(n_nearing: int, obs_flags: int = 0, n_short_term: int = N_SHORT_TERM, n_agents: int | None = None) -> int:
    """``sigmaenv_obs_dim_full``: [own] speed, (steering), short-term path, (centre-line distance), two boundary distances; per observed
    neighbour vertices (or position / rotation / length / width), velocity, (steering), (distance), (its short-term path).  With ``OBS_FULL``
    (needs ``n_agents``): the [others] part is every feature tensor of ALL agents, cut into ``n_nearing`` chunks (observation_provider_rt.py:756-851);
    raises ``ValueError`` where the reference's reshape raises."""
    s, r = int(bool(obs_flags & OBS_STEERING)), int(bool(obs_flags & OBS_REF_OTHERS))
    own = 1 + s + 2 * n_short_term + (0 if obs_flags & OBS_NO_DIST_CENTER else 1) + (20 if obs_flags & OBS_BOUNDARY_POINTS else 2) + (4 if obs_flags & OBS_BIRD_VIEW else 0)
    pad = 2 * n_nearing if obs_flags & OBS_OPPONENT_PAD else 0
    if obs_flags & OBS_FULL:
        if n_agents is None:
            raise ValueError("obs_dim: OBS_FULL needs n_agents")
        if not obs_flags & OBS_BIRD_VIEW or n_nearing < 1:
            raise ValueError("full observation (is_partial_observation=False) exists in bird view only (is_ego_view=False), with n_nearing_agents_observed >= 1")
        widths = ([2, 1, 1, 1] if obs_flags & OBS_NO_VERTICES else [8]) + [2] + [1] * s + ([] if obs_flags & OBS_NO_DIST_AGENTS else [n_agents]) + [2 * n_short_term] * r
        # the reference reshapes ALL nine feature tensors to [B, n_nearing, -1] -- position 2, rotation 1, velocity 2, reference path 2 NS, vertices 8, distance N, length 1,
        # width 1, steering 1 -- whether or not they end up in the row (observation_provider_rt.py:790-816): the width-1 tensors make N % n_nearing == 0 the condition
        if n_agents % n_nearing or any((n_agents * w) % n_nearing for w in widths):
            raise ValueError(f"full observation: {n_agents} agents x feature widths {widths} do not split into n_nearing_agents_observed = {n_nearing} chunks "
                             "(the reference's reshape(batch, n_nearing_agents, -1) raises)")
        return own + n_agents * sum(widths) + pad
    other = (5 if obs_flags & OBS_NO_VERTICES else 8) + 2 + s + (0 if obs_flags & OBS_NO_DIST_AGENTS else 1) + r * 2 * n_short_term
    return own + n_nearing * other + pad


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_PKG_DIR, "csrc", "libsigmaenv.so")

_SIGS = {
    "obs_dim": (C.c_int, [C.c_int32]),
    "obs_dim_ex": (C.c_int, [C.c_int32, C.c_int32]),
    "obs_dim_full": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "n_short_term": (C.c_int, []),
    "create": (C.c_int, [C.POINTER(Config), C.POINTER(Map), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "destroy": (None, [C.c_void_p]),
    "last_error": (C.c_char_p, [C.c_void_p]),
    "reset": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "observe": (C.c_int, [C.c_void_p]),
    "auto_reset": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]),
    "get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "sync": (C.c_int, [C.c_void_p]),
    "set_lanelets": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "opponent_fill": (C.c_int, [C.c_void_p, C.c_void_p]),
    "set_scenario_lists": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cbf_attach": (C.c_int, [C.c_void_p, C.POINTER(CbfConfig), C.c_void_p, C.c_void_p, C.c_int32]),
    "cbf_rewards": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cbf_inject_centers": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cbf_qp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cbf_regroup": (C.c_int, [C.c_void_p]),
    "cbf_get_groups": (C.c_int, [C.c_void_p, C.c_void_p]),
}
_PRODUCT_ONLY = {
    "build_id": (C.c_char_p, []),
    "step_time_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "kernel_time_ms": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "set_slab": (C.c_int, [C.c_void_p, C.c_void_p]),
    "set_rollout_slab_stride": (C.c_int, [C.c_void_p, C.c_int64]),
    "trig_selftest": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "step_autoreset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]),
    "step_autoreset_n": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_int32, C.c_int32]),
    "step_autoreset_many": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32]),
    "actor_create": (C.c_int, [C.c_int32] + [C.c_void_p] * 10 + [C.POINTER(C.c_void_p)]),
    "actor_destroy": (None, [C.c_void_p]),
    "actor_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32]),
    "mlp32_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mlp32_destroy": (None, [C.c_void_p]),
    "mlp32_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "mlp32_set_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "mlp32_get_mode": (C.c_int, [C.c_void_p]),
    "actor_forward_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                                    C.c_int32]),
    "rollout_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64,
                              C.c_int32, C.c_int32, C.c_int32]),
    "rollout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_int32,
                          C.c_int32, C.c_int32]),
}


class Library:
    """A loaded C-ABI library; ``fn.<name>`` are the typed entry points."""

    def __init__(self, path: str, prefix: str = "sigmaenv_", extra: dict | None = None):
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} not found. The HIP extension is required (no CPU fallback): build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C sigmarl_amd/csrc`."
            )
        self.path = path
        self.prefix = prefix
        self.cdll = C.CDLL(path)
        sigs = dict(_SIGS)
        sigs.update(extra or {})
        self.names = list(sigs)
        # SIGMAENV_LIB_OLD_ABI=1 (tools/ab_libs.sh only): an OLDER build of the HIP library as the A/B baseline may lack the newest entry points
        lenient = os.environ.get("SIGMAENV_LIB_OLD_ABI") == "1" and "SIGMAENV_LIB" in os.environ
        for name, (res, args) in sigs.items():
            if lenient and not hasattr(self.cdll, prefix + name):
                continue
            f = getattr(self.cdll, prefix + name)  # AttributeError if the symbol is missing
            f.restype = res
            f.argtypes = args
            setattr(self, name, f)


_product_libs = {}


def variant_path(n_short_term: int = N_SHORT_TERM) -> str:
    """The HIP library built for ``n_points_short_term`` (a build constant, include/sigmaenv.h): libsigmaenv.so for 3, libsigmaenv_ns<k>.so otherwise."""
    return DEFAULT_LIB if n_short_term == N_SHORT_TERM else os.path.join(_PKG_DIR, "csrc", f"libsigmaenv_ns{int(n_short_term)}.so")


def source_build_id() -> str:
    """What ``sigmaenv_build_id()`` of a library built from THIS tree returns: the first 16 hex digits of the SHA-256 over the files of the Makefile's SRC list, in
    that order (the Makefile is one of them)."""
    import hashlib
    import re

    csrc = os.path.join(_PKG_DIR, "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    files = re.search(r"^SRC = (.*)$", mk, re.M).group(1).split()
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def check_build_id(lib: "Library") -> None:
    """A library built from other sources than the tree's is refused (the .so files are not rebuilt on the GPU box).  SIGMAENV_ALLOW_STALE=1 turns the error into a
    warning (A/B runs against a kept older build)."""
    have, want = (lib.build_id().decode() if hasattr(lib, "build_id") else "none (an older build)"), source_build_id()
    if have != want:
        msg = (f"{lib.path} was built from other sources than this tree (build id {have}, tree {want}): run `python -c 'import __graft_entry__ as g; g.build()'` "
               f"or `make -C sigmarl_amd/csrc`")
        if os.environ.get("SIGMAENV_ALLOW_STALE") == "1":
            import warnings

            warnings.warn(msg)
        else:
            raise RuntimeError(msg)


def load_library(path: str | None = None, n_short_term: int = N_SHORT_TERM) -> Library:
    """Loads ``libsigmaenv.so`` (or the build for another ``n_points_short_term``).  torch is imported first on purpose: PyTorch-ROCm ships its own
    ``libamdhip64``; loading ours before torch's would put two HIP runtimes in the process (observed: hipGetDeviceCount fails in the second one)."""
    import torch  # noqa: F401  (must precede the CDLL below)

    n_short_term = int(n_short_term or N_SHORT_TERM)
    if path is None:
        lib = _product_libs.get(n_short_term)
        if lib is None:
            # SIGMAENV_LIB: alternative build of the same HIP library (A/B experiments); still the HIP path, never a fallback
            p = os.environ.get("SIGMAENV_LIB", DEFAULT_LIB) if n_short_term == N_SHORT_TERM else variant_path(n_short_term)
            if not os.path.exists(p) and n_short_term != N_SHORT_TERM:
                raise RuntimeError(f"{p} not found: n_points_short_term={n_short_term} needs its own build of the HIP library "
                                   f"(`make -C sigmarl_amd/csrc NS={n_short_term}`; no fallback)")
            lib = Library(p, "sigmaenv_", _PRODUCT_ONLY)
            if lib.n_short_term() != n_short_term:
                raise RuntimeError(f"{p} is built for n_points_short_term={lib.n_short_term()}, not {n_short_term}")
            check_build_id(lib)
            _product_libs[n_short_term] = lib
        return lib
    return Library(path, "sigmaenv_", _PRODUCT_ONLY)


def exported_symbols() -> list:
    """Every function ``include/sigmaenv.h`` declares (checked by the CPU test-suite against the built .so)."""
    return ["sigmaenv_" + n for n in list(_SIGS) + list(_PRODUCT_ONLY)]
