"""``SigmaEnv``: device-buffer front end of the HIP environment step (thin Python over the C-ABI).

PyTorch is used for plumbing only: device selection, the HIP stream, and zero-copy ``torch.Tensor`` views of the
library-owned output buffers (``sigmaenv_get``).  All arithmetic of the path happens in ``csrc/sigmaenv.hip``.
There is no CPU fallback: constructing a ``SigmaEnv`` without the built extension or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi
from .maps import MapTable, load_map
from .params import Parameters, make_config

_TORCH_DTYPES = {"f32": (torch.float32, "<f4", 4), "i32": (torch.int32, "<i4", 4), "u8": (torch.uint8, "|u1", 1)}


def _buf_layout(B, N, K, D, NS=capi.N_SHORT_TERM):
    return {
        capi.BUF_STATE: ("f32", (B, N, 8)), capi.BUF_PREV_POS: ("f32", (B, N, 2)), capi.BUF_VERTICES: ("f32", (B, N, 5, 2)),
        capi.BUF_PATH: ("i32", (B, N, 4)), capi.BUF_SHORT_TERM: ("f32", (B, N, NS, 2)), capi.BUF_DIST_REF: ("f32", (B, N)),
        capi.BUF_DIST_LEFT: ("f32", (B, N, 5)), capi.BUF_DIST_RIGHT: ("f32", (B, N, 5)), capi.BUF_DIST_BOUND: ("f32", (B, N)),
        capi.BUF_CLOSEST: ("i32", (B, N, 3)), capi.BUF_DIST_AGENTS: ("f32", (B, N, N)), capi.BUF_COL_AGENTS: ("u8", (B, N, N)),
        capi.BUF_COL_FLAGS: ("u8", (B, N, 4)), capi.BUF_REWARD: ("f32", (B, N)), capi.BUF_REWARD_INFO: ("f32", (12, B, N)),
        capi.BUF_OBS: ("f32", (B, N, D)), capi.BUF_NEARING: ("i32", (B, N, K)), capi.BUF_DONE: ("u8", (B,)),
        capi.BUF_TIMER: ("i32", (B, 4)), capi.BUF_ACTION: ("f32", (B, N, 2)),
        capi.BUF_CBF_NOMINAL: ("f32", (B, N, 2)),
    }


# ---- zero-copy torch view of a raw device pointer ---------------------------------------------------------------
class _CudaArray:
    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2, "strides": None}
        self._owner = owner


class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int32), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]
_noop_deleter = _DELETER(lambda p: None)
_keepalive = []


def _view_dlpack(ptr, shape, kind, device_index):
    code, bits = {"f32": (2, 32), "i32": (0, 32), "u8": (1, 8)}[kind]
    shp = (C.c_int64 * len(shape))(*shape)
    mt = _DLManagedTensor()
    mt.dl_tensor.data = int(ptr)
    mt.dl_tensor.device = _DLDevice(10, device_index)  # kDLROCM
    mt.dl_tensor.ndim = len(shape)
    mt.dl_tensor.dtype = _DLDataType(code, bits, 1)
    mt.dl_tensor.shape = shp
    mt.dl_tensor.strides = None
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _noop_deleter
    _keepalive.append((mt, shp))
    C.pythonapi.PyCapsule_New.restype = C.py_object
    C.pythonapi.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    cap = C.pythonapi.PyCapsule_New(C.addressof(mt), b"dltensor", None)
    return torch.utils.dlpack.from_dlpack(cap)


def device_view(ptr, shape, kind, device_index, owner):
    """torch.Tensor aliasing ``ptr`` (no copy).  Empty shapes get an ordinary empty tensor."""
    tdtype, typestr, _ = _TORCH_DTYPES[kind]
    if int(np.prod(shape)) == 0:
        return torch.empty(shape, dtype=tdtype, device=f"cuda:{device_index}")
    try:
        t = torch.as_tensor(_CudaArray(ptr, shape, typestr, owner), device=f"cuda:{device_index}")
        if t.data_ptr() == int(ptr):
            return t
    except Exception:
        pass
    t = _view_dlpack(ptr, shape, kind, device_index)
    assert t.data_ptr() == int(ptr), "zero-copy view of the device buffer failed"
    return t


class SigmaEnv:
    """One env shard on one GPU.  ``step(actions)`` is one fused HIP launch over agents x envs."""

    def __init__(self, parameters: Parameters | None = None, n_envs: int | None = None, device=None, *, cfg: capi.Config | None = None,
                 map_table: MapTable | None = None, make_world_scenario_type: str = "cpm_entire", lib_path: str | None = None,
                 envs_per_group: int = 0, env_index_base: int | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("sigmarl_amd.SigmaEnv needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        if cfg is None:
            if parameters is None:
                raise ValueError("pass `parameters` (+ n_envs) or a ready `cfg` + `map_table`")
            map_table = map_table or load_map(parameters.scenario_type)
            cfg = make_config(parameters, map_table, int(n_envs if n_envs is not None else parameters.num_vmas_envs), make_world_scenario_type)
        self.n_short_term = int(getattr(cfg, "n_points_short_term", 0) or capi.N_SHORT_TERM)
        self.lib = capi.load_library(lib_path, self.n_short_term)  # (n_points_short_term is a build constant: one library per value)
        if env_index_base is not None:  # this shard's first env in the whole batch (shard.shard_range): the random draws of env e do not depend on the sharding
            cfg.env_index_base = int(env_index_base)
        if envs_per_group:  # full tiles even for a small shard (several handles stepped concurrently on different streams)
            cfg.envs_per_group = int(envs_per_group)
        self.parameters = parameters
        self.cfg = cfg
        self.map = map_table
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.B, self.N, self.K = cfg.n_envs, cfg.n_agents, cfg.n_nearing
        self.D = capi.obs_dim(self.K, int(getattr(self.cfg, "obs_flags", 0)), self.n_short_term, self.N)
        self._map_struct = map_table.as_struct()
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream(self.device)
            h = C.c_void_p()
            rc = self.lib.create(C.byref(cfg), C.byref(self._map_struct), self.device.index, C.c_void_p(self.stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_create failed with code {rc}")
        self.h = h
        self._views = {}
        layout = _buf_layout(self.B, self.N, self.K, self.D, self.n_short_term)
        for which, (kind, shape) in layout.items():
            p = C.c_void_p()
            nb = C.c_size_t()
            self._chk(self.lib.get(self.h, which, C.byref(p), C.byref(nb)), "get")
            assert int(np.prod(shape)) * _TORCH_DTYPES[kind][2] == nb.value, (which, shape, nb.value)
            self._views[which] = device_view(p.value, shape, kind, self.device.index, self)
        self._reset_counter = 0
        self._scenario_lists = False
        if getattr(map_table, "scenario_type", "") == "cpm_mixed":  # per-env sub-scenario path lists (world_state_rt_sim.py:313-358)
            probs = list(getattr(parameters, "cpm_scenario_probabilities", None) or [1.0, 0.0, 0.0]) if parameters is not None else [1.0, 0.0, 0.0]
            self.set_scenario_lists(probs)
        # bird view + is_apply_mask: the lanelet-relation mask needs the map's lanelet tables (none on the CPM map: the mask is empty there)
        if (int(getattr(cfg, "obs_flags", 0)) & capi.OBS_BIRD_VIEW) and cfg.is_apply_mask and map_table.lanelet_tables() is not None:
            centers, neigh = map_table.lanelet_tables()
            self._lanelets = (centers, neigh)
            self._chk(self.lib.set_lanelets(self.h, int(centers.shape[0]), int(centers.shape[1]), centers.ctypes.data_as(C.c_void_p), neigh.ctypes.data_as(C.c_void_p)),
                      "set_lanelets")

    def set_scenario_lists(self, probabilities):
        """``cpm_mixed``: the device-side resets draw every finished env's sub-scenario (intersection / merge-in / merge-out: the map's path lists 1..3)
        with these probabilities and keep it for the env's per-agent resets (``sigmaenv_set_scenario_lists``)."""
        n = len(probabilities)
        first = np.asarray([self.map.list_first[k + 1] for k in range(n)], np.int32)
        count = np.asarray([self.map.list_count[k + 1] for k in range(n)], np.int32)
        pr = np.asarray(probabilities, np.float32)
        self._chk(self.lib.set_scenario_lists(self.h, n, first.ctypes.data_as(C.c_void_p), count.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p)),
                  "set_scenario_lists")
        self._scenario_lists = True

    def default_paths(self):
        """(path_first, path_count) of a device-side reset when the caller names none: the map's whole path list, or the sub-scenario lists (cpm_mixed)."""
        if self._scenario_lists:
            return 0, capi.SCENARIO_LISTS
        return self.map.list_first[0], self.map.list_count[0]

    # ---- plumbing ---------------------------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.last_error(self.h)
            raise RuntimeError(f"sigmaenv_{what} failed with code {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.destroy(self.h)
            self.h = None
            self._views = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def buffer(self, which) -> torch.Tensor:
        """Zero-copy view of a library-owned device buffer (see ``sigmaenv_buf_t``)."""
        return self._views[which]

    # ---- the C-ABI calls ----------------------------------------------------------------------------------------
    def reset(self, env_idx, agent_idx, path_ids, state8, full_env: bool):
        env_idx = np.ascontiguousarray(env_idx, np.int32)
        agent_idx = np.ascontiguousarray(agent_idx, np.int32)
        path_ids = np.ascontiguousarray(path_ids, np.int32).reshape(-1, 4)
        state8 = np.ascontiguousarray(state8, np.float32).reshape(-1, 8)
        n = len(env_idx)
        assert len(agent_idx) == n and len(path_ids) == n and len(state8) == n
        with torch.cuda.device(self.device):
            self._chk(self.lib.reset(self.h, n, env_idx.ctypes.data_as(C.c_void_p), agent_idx.ctypes.data_as(C.c_void_p),
                                     path_ids.ctypes.data_as(C.c_void_p), state8.ctypes.data_as(C.c_void_p), int(bool(full_env))), "reset")

    def step(self, actions: torch.Tensor):
        """actions: float32 CUDA tensor [B, N, 2] (v_cmd, delta_cmd); enqueues one fused step on the env's stream."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()):
            raise TypeError("actions must be a contiguous float32 CUDA tensor")
        if tuple(actions.shape) != (self.B, self.N, 2):
            raise ValueError(f"actions must have shape {(self.B, self.N, 2)}, got {tuple(actions.shape)}")
        self._chk(self.lib.step(self.h, C.c_void_p(actions.data_ptr())), "step")

    def observe(self):
        self._chk(self.lib.observe(self.h), "observe")

    def opponent_fill(self, actions: torch.Tensor):
        """Opponent modelling (helper_training.py:1117-1137): writes the tentative ``actions`` [B, N, 2] of every agent's observed neighbours into
        the placeholder columns at the end of its observation row (``is_using_opponent_modeling``)."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()):
            raise TypeError("actions must be a contiguous float32 CUDA tensor")
        if tuple(actions.shape) != (self.B, self.N, 2):
            raise ValueError(f"actions must have shape {(self.B, self.N, 2)}, got {tuple(actions.shape)}")
        self._chk(self.lib.opponent_fill(self.h, C.c_void_p(actions.data_ptr())), "opponent_fill")

    # ---- QP-free CBF margin reward (sigmarl/cbf_qp.py:2534-2804) ------------------------------------------------------
    def cbf_attach(self, cbf_cfg=None, seg_left=None, seg_right=None):
        """Uploads the pseudo-distance segment tables of the env's map and the CBF constants.  Defaults: ``make_cbf_config`` of
        the env's parameters and ``load_segment_tables`` of its map."""
        from . import cbf

        if cbf_cfg is None:
            cbf_cfg = cbf.make_cbf_config(self.parameters)
        if seg_left is None or seg_right is None:
            seg_left, seg_right = cbf.load_segment_tables(self.map)
        seg_left = np.ascontiguousarray(seg_left, np.float32)
        seg_right = np.ascontiguousarray(seg_right, np.float32)
        assert seg_left.shape == seg_right.shape and seg_left.shape[0] == self.map.n_paths and seg_left.shape[2] == cbf.SEG_FIELDS
        self.cbf_cfg = cbf_cfg
        with torch.cuda.device(self.device):
            self._chk(self.lib.cbf_attach(self.h, C.byref(cbf_cfg), seg_left.ctypes.data_as(C.c_void_p), seg_right.ctypes.data_as(C.c_void_p),
                                          int(seg_left.shape[1])), "cbf_attach")

    def cbf_margin_count(self) -> int:
        if getattr(self, "cbf_cfg", None) is None:
            raise RuntimeError("sigmaenv cbf: cbf_attach has not been called")
        Cc = int(self.cbf_cfg.n_circles)
        return 2 * self.B * self.N * Cc + self.B * self.N * self.N * Cc * Cc

    def cbf_rewards(self, actions: torch.Tensor, margins: torch.Tensor | None = None):
        """``CBFQP.update_qp`` (``is_solve_qp=False``) for every env: writes rew_near_left_lane / rew_near_right_lane /
        rew_near_other_agents of ``BUF_REWARD_INFO``; the next ``step`` adds them when ``rew_method`` contains "cbf".
        ``margins``: optional float64 CUDA tensor with ``cbf_margin_count()`` elements (lane_left, lane_right, pair)."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()):
            raise TypeError("actions must be a contiguous float32 CUDA tensor")
        if tuple(actions.shape) != (self.B, self.N, 2):
            raise ValueError(f"actions must have shape {(self.B, self.N, 2)}, got {tuple(actions.shape)}")
        mp = None
        if margins is not None:
            if not (margins.is_cuda and margins.dtype == torch.float64 and margins.is_contiguous() and margins.numel() == self.cbf_margin_count()):
                raise TypeError("margins must be a contiguous float64 CUDA tensor with cbf_margin_count() elements")
            mp = C.c_void_p(margins.data_ptr())
        self._warn_cbf_noise()
        self._chk(self.lib.cbf_rewards(self.h, C.c_void_p(actions.data_ptr()), mp), "cbf_rewards")

    def cbf_qp(self, actions: torch.Tensor, actions_safe: torch.Tensor | None = None, u_opt: torch.Tensor | None = None, info: torch.Tensor | None = None):
        """The centralized CBF-QP safety filter of every env (``CBFQP.update_centralized_cbf_qp``, ``sigmarl/cbf_qp.py:1019-1400``):
        returns ``actions_safe`` float32 [B, N, 2] = ``u_to_rl_action`` of the minimiser; ``u_opt`` (float64 [B, N, 2]) and ``info``
        (int32 [B, 2]: Newton iterations, converged) are filled when given."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()):
            raise TypeError("actions must be a contiguous float32 CUDA tensor")
        if tuple(actions.shape) != (self.B, self.N, 2):
            raise ValueError(f"actions must have shape {(self.B, self.N, 2)}, got {tuple(actions.shape)}")
        if actions_safe is None:
            actions_safe = torch.empty_like(actions)
        for t, dt in ((actions_safe, torch.float32), (u_opt, torch.float64), (info, torch.int32)):
            if t is not None and not (t.is_cuda and t.dtype == dt and t.is_contiguous()):
                raise TypeError("cbf_qp outputs must be contiguous CUDA tensors (float32 / float64 / int32)")
        self._warn_cbf_noise()
        self._chk(self.lib.cbf_qp(self.h, C.c_void_p(actions.data_ptr()), C.c_void_p(actions_safe.data_ptr()),
                                  C.c_void_p(u_opt.data_ptr()) if u_opt is not None else None,
                                  C.c_void_p(info.data_ptr()) if info is not None else None), "cbf_qp")
        return actions_safe

    def cbf_groups(self) -> np.ndarray:
        """[B, N] int32: the group of every vehicle (grouped CBF-QPs, ``Parameters.is_grouping_agents``), formed by the first ``cbf_qp`` call
        and kept (``use_fixed_groups``, ``sigmarl/cbf_qp.py:1897-1909``)."""
        g = np.zeros((self.B, self.N), np.int32)
        self._chk(self.lib.cbf_get_groups(self.h, g.ctypes.data_as(C.c_void_p)), "cbf_get_groups")
        return g

    def cbf_regroup(self):
        """Forget the groups: the next ``cbf_qp`` call forms them again (what constructing new ``CBFQP`` objects does in the reference)."""
        self._chk(self.lib.cbf_regroup(self.h), "cbf_regroup")

    def step_autoreset(self, actions: torch.Tensor, seed: int = 0, counter: int | None = None, path_first: int | None = None,
                       path_count: int | None = None):
        """``step`` + ``auto_reset`` in one launch (same end state); the terminal observation goes to the slab only."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()):
            raise TypeError("actions must be a contiguous float32 CUDA tensor")
        if tuple(actions.shape) != (self.B, self.N, 2):
            raise ValueError(f"actions must have shape {(self.B, self.N, 2)}, got {tuple(actions.shape)}")
        if counter is None:
            counter = self._reset_counter
            self._reset_counter += 1
        if path_first is None:
            path_first, path_count = self.default_paths()
        self._chk(self.lib.step_autoreset(self.h, C.c_void_p(actions.data_ptr()), int(seed), int(counter), int(path_first), int(path_count)),
                  "step_autoreset")

    def step_autoreset_n(self, actions: torch.Tensor, slab: torch.Tensor | None = None, seed: int = 0, counter0: int | None = None,
                         path_first: int | None = None, path_count: int | None = None):
        """``actions.shape[0]`` fused steps in ONE launch (``sigmaenv_step_autoreset_n``): ``actions`` float32 CUDA ``[T, B, N, 2]``, optional record
        ``slab`` ``[T, B, N*(D+1)+1]``; same end state, record and reset draws as T calls of ``step_autoreset`` with counters ``counter0 + t``."""
        if not (isinstance(actions, torch.Tensor) and actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()):
            raise TypeError("actions must be a contiguous float32 CUDA tensor")
        if actions.dim() != 4 or tuple(actions.shape[1:]) != (self.B, self.N, 2) or actions.shape[0] < 1:
            raise ValueError(f"actions must have shape (T, {self.B}, {self.N}, 2), got {tuple(actions.shape)}")
        T = int(actions.shape[0])
        W = self.N * (self.D + 1) + 1
        if slab is not None:
            if not (slab.is_cuda and slab.dtype == torch.float32 and slab.is_contiguous() and tuple(slab.shape) == (T, self.B, W)):
                raise ValueError(f"slab must be a contiguous float32 CUDA tensor of shape {(T, self.B, W)}")
        if counter0 is None:
            counter0 = self._reset_counter
            self._reset_counter += T
        if path_first is None:
            path_first, path_count = self.default_paths()
        self.step_autoreset_n_ptr(actions.data_ptr(), T, self.B * self.N * 2, slab.data_ptr() if slab is not None else 0, self.B * W, seed, counter0,
                                  path_first, path_count)

    def step_autoreset_n_ptr(self, actions_ptr: int, n_steps: int, action_stride: int, slab_ptr: int, slab_stride: int, seed: int, counter0: int,
                             path_first: int, path_count: int):
        """Pointer-level form (strides in floats): lets a caller step several env shards into one ``[T, B_total, W]`` chunk buffer."""
        self._chk(self.lib.step_autoreset_n(self.h, C.c_void_p(actions_ptr), int(n_steps), int(action_stride), C.c_void_p(slab_ptr) if slab_ptr else None,
                                            int(slab_stride), int(seed), int(counter0), int(path_first), int(path_count)), "step_autoreset_n")

    def auto_reset(self, seed: int = 0, counter: int | None = None, path_first: int | None = None, path_count: int | None = None):
        if counter is None:
            counter = self._reset_counter
            self._reset_counter += 1
        if path_first is None:
            path_first, path_count = self.default_paths()
        self._chk(self.lib.auto_reset(self.h, int(seed), int(counter), int(path_first), int(path_count)), "auto_reset")

    def _warn_cbf_noise(self):
        """With ``is_obs_noise`` the reference perturbs the policy's action before it enters the "rl" nominal controller of the CBF module
        (``rl_i + rand_like(rl_i) * obs_noise_level``, cbf_qp.py:1060-1061, :2608-2609); the device path takes the action as given."""
        if getattr(self.parameters, "is_obs_noise", False) and not getattr(self, "_cbf_noise_warned", False):
            import warnings

            self._cbf_noise_warned = True
            warnings.warn("sigmarl_amd: is_obs_noise=True, but the CBF module on the device uses the policy's action without the reference's "
                          "action noise (cbf_qp.py:2608-2609); add it to the action tensor before the call to reproduce it", stacklevel=3)

    def set_slab(self, slab: torch.Tensor | None):
        """Route the per-step rollout record ([B, N*(D+1)+1] fp32: obs | reward | done) into ``slab`` (None disables it)."""
        if slab is None:
            self._chk(self.lib.set_slab(self.h, None), "set_slab")
            return
        if not (slab.is_cuda and slab.dtype == torch.float32 and slab.is_contiguous() and tuple(slab.shape) == (self.B, self.N * (self.D + 1) + 1)):
            raise ValueError(f"slab must be a contiguous float32 CUDA tensor of shape {(self.B, self.N * (self.D + 1) + 1)}")
        self._chk(self.lib.set_slab(self.h, C.c_void_p(slab.data_ptr())), "set_slab")

    def set_rollout_slab_stride(self, stride_floats: int = 0):
        """Floats between the record blocks of consecutive steps of ``Actor.rollout`` (0: this handle's own ``[T, B, W]`` layout).  An env shard that records into
        a ``[T, B_total, W]`` buffer of the whole batch passes ``B_total * W`` here and the address of its first row as the rollout's slab pointer."""
        self._chk(self.lib.set_rollout_slab_stride(self.h, int(stride_floats)), "set_rollout_slab_stride")

    # pointer-level variants for rollout loops that precompute their device addresses (no per-call tensor checks / views)
    def set_slab_ptr(self, ptr: int):
        self._chk(self.lib.set_slab(self.h, C.c_void_p(ptr)), "set_slab")

    def step_autoreset_ptr(self, actions_ptr: int, seed: int, counter: int, path_first: int, path_count: int):
        self._chk(self.lib.step_autoreset(self.h, C.c_void_p(actions_ptr), seed, counter, path_first, path_count), "step_autoreset")

    def sync(self):
        self._chk(self.lib.sync(self.h), "sync")

    def kernel_time_ms(self, kernel_id: int):
        """(average ms, launches) of the HIP-event brackets of kernel ``capi.KERNEL_*`` since the last call; the first call arms the bracketing."""
        avg = C.c_double()
        n = C.c_int32()
        self._chk(self.lib.kernel_time_ms(self.h, int(kernel_id), C.byref(avg), C.byref(n)), "kernel_time_ms")
        return avg.value, n.value

    def step_time_ms(self):
        avg = C.c_double()
        n = C.c_int32()
        self._chk(self.lib.step_time_ms(self.h, C.byref(avg), C.byref(n)), "step_time_ms")
        return avg.value, n.value

    # ---- convenience --------------------------------------------------------------------------------------------
    @property
    def obs(self):
        return self._views[capi.BUF_OBS]

    @property
    def reward(self):
        return self._views[capi.BUF_REWARD]

    @property
    def done(self):
        return self._views[capi.BUF_DONE]

    @property
    def state(self):
        return self._views[capi.BUF_STATE]

    def reset_injected(self, predefined_ref_path_idx, init_state):
        """Initial reset of every env from ONE injected state (``Parameters.predefined_ref_path_idx`` / ``init_state``: path per agent and
        [x, y, yaw] per agent; speed, steering, velocity and side-slip zero -- world_state_rt_sim.py:99-126), the same in every env."""
        B, N = self.B, self.N
        idx = np.asarray(predefined_ref_path_idx, np.int32).reshape(N)
        st = np.asarray(init_state, np.float32).reshape(N, -1)[:, 0:3]
        ids = np.zeros((B, N, 4), np.int32)
        ids[..., 0] = np.asarray([self.map.global_path(0, int(p)) for p in idx], np.int32)[None, :]
        ids[..., 2] = idx[None, :]
        state8 = np.zeros((B, N, 8), np.float32)
        state8[..., 0:3] = st[None, :, :]
        self.reset(np.repeat(np.arange(B, dtype=np.int32), N), np.tile(np.arange(N, dtype=np.int32), B), ids.reshape(-1, 4), state8.reshape(-1, 8), True)
        self.observe()

    def reset_random(self, seed: int = 0):
        """Initial reset of every env through the device-side sampler (marks all envs done first)."""
        self._views[capi.BUF_DONE].fill_(1)
        self.auto_reset(seed=seed)


from .cbf import split_cbf_margins  # noqa: E402


class NumpyAdapter:
    """Host-array facade over ``SigmaEnv`` with the interface of ``tests/oracle_binding.OracleEnv`` (used by the parity tests)."""

    def __init__(self, env: SigmaEnv):
        self.env = env
        self.B, self.N, self.K, self.D = env.B, env.N, env.K, env.D

    def reset(self, env_idx, agent_idx, path_ids, state8, full_env):
        self.env.reset(env_idx, agent_idx, path_ids, state8, full_env)

    def step(self, actions):
        a = torch.as_tensor(np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)).to(self.env.device)
        self.env.step(a)
        self.env.sync()

    def observe(self):
        self.env.observe()

    def set_scenario_lists(self, probabilities):
        self.env.set_scenario_lists(probabilities)

    def opponent_fill(self, actions):
        a = torch.as_tensor(np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)).to(self.env.device)
        self.env.opponent_fill(a)
        self.env.sync()

    def cbf_attach(self, cbf_cfg, seg_left, seg_right):
        self.env.cbf_attach(cbf_cfg, seg_left, seg_right)

    def cbf_inject_centers(self, centers):
        """Test hook (sigmaenv_cbf_inject_centers): circle centres [B,N,C,2] float32 replace the computed ones; None clears."""
        self._centers = None if centers is None else torch.as_tensor(np.ascontiguousarray(centers, np.float32)).to(self.env.device).contiguous()
        self.env._chk(self.env.lib.cbf_inject_centers(self.env.h, C.c_void_p(self._centers.data_ptr()) if self._centers is not None else None), "cbf_inject_centers")

    def cbf_rewards(self, actions, want_margins=True):
        a = torch.as_tensor(np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)).to(self.env.device)
        m = torch.full((self.env.cbf_margin_count(),), float("nan"), dtype=torch.float64, device=self.env.device) if want_margins else None
        self.env.cbf_rewards(a, m)
        self.env.sync()
        return None if m is None else split_cbf_margins(m.cpu().numpy(), self.B, self.N, int(self.env.cbf_cfg.n_circles))

    def cbf_qp(self, actions):
        a = torch.as_tensor(np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)).to(self.env.device)
        u = torch.zeros((self.B, self.N, 2), dtype=torch.float64, device=self.env.device)
        info = torch.zeros((self.B, 2), dtype=torch.int32, device=self.env.device)
        safe = self.env.cbf_qp(a, None, u, info)
        self.env.sync()
        return safe.cpu().numpy(), u.cpu().numpy(), info.cpu().numpy()

    def cbf_groups(self):
        return self.env.cbf_groups()

    def cbf_regroup(self):
        self.env.cbf_regroup()

    def auto_reset(self, seed, counter, path_first, path_count):
        self.env.auto_reset(seed, counter, path_first, path_count)

    def step_autoreset(self, actions, seed, counter, path_first, path_count):
        a = torch.as_tensor(np.ascontiguousarray(actions, np.float32).reshape(self.B, self.N, 2)).to(self.env.device)
        self.env.step_autoreset(a, seed, counter, path_first, path_count)
        self.env.sync()

    def get(self, which, copy=True):
        self.env.sync()
        return self.env.buffer(which).cpu().numpy()

    def close(self):
        self.env.close()
