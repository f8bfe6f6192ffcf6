"""The policy (actor) forward pass and whole rollouts on the device.

``Actor`` wraps ``sigmaenv_actor_*`` / ``sigmaenv_rollout`` of the C-ABI: the actor network of
``sigmarl/modules/decision_making_module.py:34-82`` (torchrl ``MultiAgentMLP(depth=3, num_cells=256, activation=Tanh, share_params=True)``
+ ``NormalParamExtractor`` + ``TanhNormal``).  Two precisions:
  ``precision="fp32"`` (default)  the reference's fp32 network on the matrix cores (``sigmaenv_actor_forward_f32``); ``mode="split"`` (default) forms every fp32 product
                                  from three exact fp16 products, ``mode="exact"`` runs fp32 fma chains (2.7 x slower); both within 1e-5 of torch.nn
  ``precision="bf16"``            the fast inference variant: bf16 weights / activations, fp32 accumulation, one fused MFMA kernel; also what
                                  ``sigmaenv_rollout`` (policy + step without the host in the loop) runs
``Critic`` is the MAPPO critic of ``sigmarl/modules/optimization_module.py:16-32`` (centralised, shared parameters) in fp32 (same modes).
Weights come from any ``torch.nn.Sequential`` of four ``Linear`` layers (the parameter layout torchrl's shared-parameter MLP has).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi
from .env import SigmaEnv


def make_mlp(obs_dim: int = 32, hidden: int = 256, n_out: int = 4) -> torch.nn.Sequential:
    """The reference's actor architecture in plain torch.nn (what ``MultiAgentMLP(share_params=True)`` applies to every agent row)."""
    return torch.nn.Sequential(torch.nn.Linear(obs_dim, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, hidden), torch.nn.Tanh(),
                               torch.nn.Linear(hidden, hidden), torch.nn.Tanh(), torch.nn.Linear(hidden, n_out))


class Mlp32:
    """``sigmaenv_mlp32_*``: a Tanh MLP with hidden width 256 in fp32 on the matrix cores; ``forward(env, x[rows, in_dim]) -> [rows, out_dim]``.
    ``mode``: ``"split"`` (default: every fp32 operand as two fp16 numbers, three exact-product MFMAs per fp32 product, 3/16 of the fp32 matrix time; falls back
    to exact when a weight is outside +-255) or ``"exact"`` (``v_mfma_f32_32x32x2_f32`` fma chains).  Both are held to torch.nn within 1e-5."""

    def __init__(self, mlp: torch.nn.Module, lib: capi.Library | None = None, mode: str = "split"):
        if mode not in ("split", "exact"):
            raise ValueError("mode must be 'split' or 'exact'")
        self.mode = mode
        # A handle belongs to the library that made it, and an env handle to the build of ITS n_points_short_term (libsigmaenv_ns<k>.so): every call
        # that takes ``env.h`` goes through ``env.lib`` with a network handle created by that same library (one per library, made on first use).
        self.lib = lib or capi.load_library()
        self._handles = {}
        lin = [m for m in mlp.modules() if isinstance(m, torch.nn.Linear)]
        if not (2 <= len(lin) <= 4) or any(m.out_features != 256 for m in lin[:-1]) or lin[-1].out_features > 32:
            raise ValueError("expected 2-4 Linear layers with hidden width 256 and at most 32 outputs")
        self.in_dim, self.out_dim = lin[0].in_features, lin[-1].out_features
        dims = np.asarray([self.in_dim] + [m.out_features for m in lin], np.int32)
        ws = [np.ascontiguousarray(m.weight.detach().cpu().numpy(), np.float32) for m in lin]
        bs = [np.ascontiguousarray(m.bias.detach().cpu().numpy(), np.float32) for m in lin]
        self._keep = (dims, ws, bs)
        self.h = self.handle(self.lib)

    def handle(self, lib: capi.Library):
        """The network's handle in ``lib`` (created on first use)."""
        ent = self._handles.get(lib.path)
        if ent is None:
            dims, ws, bs = self._keep
            PA = C.c_void_p * len(ws)
            wp, bp = PA(*[w.ctypes.data for w in ws]), PA(*[b.ctypes.data for b in bs])
            h = C.c_void_p()
            rc = lib.mlp32_create(len(ws), dims.ctypes.data_as(C.c_void_p), wp, bp, C.byref(h))
            if rc != 0:
                raise RuntimeError(f"sigmaenv_mlp32_create failed with code {rc}")
            if self.mode == "exact":
                lib.mlp32_set_mode(h, capi.MLP32_EXACT)
            ent = self._handles[lib.path] = (lib, h)
        return ent[1]

    def set_mode(self, mode: str) -> str:
        """Switch every handle to ``"split"`` / ``"exact"``; returns the mode in force (a network outside the split form's range stays exact)."""
        if mode not in ("split", "exact"):
            raise ValueError("mode must be 'split' or 'exact'")
        self.mode = mode
        got = mode
        for lib, h in self._handles.values():
            lib.mlp32_set_mode(h, capi.MLP32_SPLIT if mode == "split" else capi.MLP32_EXACT)
            got = "split" if lib.mlp32_get_mode(h) == capi.MLP32_SPLIT else "exact"
        return got

    def close(self):
        for lib, h in getattr(self, "_handles", {}).values():
            lib.mlp32_destroy(h)
        self._handles = {}
        self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def forward(self, env: SigmaEnv, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] == self.in_dim):
            raise TypeError(f"input must be a contiguous float32 CUDA tensor [..., {self.in_dim}]")
        rows = x.numel() // self.in_dim
        if out is None:
            out = torch.empty((*x.shape[:-1], self.out_dim), dtype=torch.float32, device=x.device)
        rc = env.lib.mlp32_forward(env.h, self.handle(env.lib), C.c_void_p(x.data_ptr()), rows, C.c_void_p(out.data_ptr()))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_mlp32_forward failed with code {rc}: {env.lib.last_error(env.h).decode()}")
        return out


class Critic(Mlp32):
    """The MAPPO critic (optimization_module.py:16-32): ``MultiAgentMLP(centralised=True, share_params=True, depth=3, num_cells=256, Tanh)`` --
    one shared MLP on the concatenated observations of all agents of an env; its output is the state value of every agent of that env."""

    def values(self, env: SigmaEnv, obs: torch.Tensor | None = None) -> torch.Tensor:
        """[B, N, 1] state values of ``env.obs`` (or ``obs [B, N, D]``)."""
        o = env.obs if obs is None else obs
        if self.in_dim != env.N * env.D:
            raise ValueError(f"critic input width {self.in_dim} != n_agents * obs_dim = {env.N * env.D}")
        v = self.forward(env, o.reshape(env.B, env.N * env.D))
        return v.reshape(env.B, 1, 1).expand(env.B, env.N, 1)


class Actor:
    def __init__(self, mlp: torch.nn.Module, low, high, lib: capi.Library | None = None, precision: str = "fp32", mode: str = "split"):
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' (the reference's arithmetic) or 'bf16' (fast inference variant)")
        self.precision = precision
        self.lib = lib or capi.load_library()
        self._mlp32 = Mlp32(mlp, self.lib, mode=mode)  # the exact network (also kept by the bf16 variant: a caller may ask for either per call)
        import weakref

        self._scratch_by_env = weakref.WeakKeyDictionary()  # SigmaEnv -> {kind: tensor}
        lin = [m for m in mlp.modules() if isinstance(m, torch.nn.Linear)]
        if len(lin) != 4 or lin[1].in_features != 256 or lin[2].out_features != 256 or lin[3].out_features != 4:
            raise ValueError("expected Linear(D,256), Linear(256,256), Linear(256,256), Linear(256,4)")
        self.obs_dim = lin[0].in_features
        arrs = []
        for m in lin:
            arrs.append(np.ascontiguousarray(m.weight.detach().cpu().numpy(), np.float32))
            arrs.append(np.ascontiguousarray(m.bias.detach().cpu().numpy(), np.float32))
        self._keep = arrs + [np.ascontiguousarray(low, np.float32), np.ascontiguousarray(high, np.float32)]
        # the bf16 kernel handle exists only for the widths its MFMA tiling takes (sigmaenv_actor_create: obs_dim in {8, 16, 24, 32}); other
        # observation switches (e.g. is_obs_steering: 35) run the fp32 network, which takes any width
        self._bf16 = {}  # library path -> (library, handle): see Mlp32
        if precision == "bf16":
            self._bf16_handle(self.lib)

    def _bf16_handle(self, lib: capi.Library):
        ent = self._bf16.get(lib.path)
        if ent is None:
            if self.obs_dim not in (8, 16, 24, 32):
                raise ValueError(f"the bf16 actor kernel takes obs_dim 8, 16, 24 or 32, not {self.obs_dim}: use precision='fp32' for this observation layout")
            h = C.c_void_p()
            rc = lib.actor_create(self.obs_dim, *[a.ctypes.data_as(C.c_void_p) for a in self._keep], C.byref(h))
            if rc != 0:
                raise RuntimeError(f"sigmaenv_actor_create failed with code {rc}")
            ent = self._bf16[lib.path] = (lib, h)
        return ent[1]

    def close(self):
        for lib, h in getattr(self, "_bf16", {}).values():
            lib.actor_destroy(h)
        self._bf16 = {}
        if getattr(self, "_mlp32", None) is not None:
            self._mlp32.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # Scratch buffers are keyed PER ENV OBJECT: one Actor may drive several env shards on their own streams (bench.py --policy --streams 2), and a buffer shared by two
    # shards of equal size would be overwritten by the other shard's launches.  The key is the SigmaEnv itself, held weakly (the raw handle
    # address can be handed to a NEW env after close() -- same B, larger N: an undersized buffer, written past its end -- and the entries of closed envs never went
    # away); the FULL shape is compared as well.
    def _scratch(self, kind: str, env: SigmaEnv, shape, zero: bool) -> torch.Tensor:
        per_env = self._scratch_by_env.setdefault(env, {})
        t = per_env.get(kind)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != env.device:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=torch.float32, device=env.device)
            per_env[kind] = t
        return t

    def _scratch_actions(self, env: SigmaEnv) -> torch.Tensor:
        return self._scratch("a", env, (env.B, env.N, 2), True)

    def _scratch_out4(self, env: SigmaEnv) -> torch.Tensor:
        return self._scratch("o", env, (env.B * env.N, 4), False)

    def forward(self, env: SigmaEnv, actions: torch.Tensor, log_prob: torch.Tensor | None = None, loc_scale: torch.Tensor | None = None,
                obs: torch.Tensor | None = None, seed: int = 0, counter: int = 0, deterministic: bool = False, precision: str | None = None):
        """actions[B,N,2] := policy(env.obs or ``obs``); enqueued on the env's stream."""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        if (precision or self.precision) == "fp32":
            lo, hi = self._keep[-2], self._keep[-1]
            rc = env.lib.actor_forward_f32(env.h, self._mlp32.handle(env.lib), p(obs), p(self._scratch_out4(env)), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p),
                                            p(actions), p(log_prob), p(loc_scale), int(seed), int(counter), int(bool(deterministic)))
            if rc != 0:
                raise RuntimeError(f"sigmaenv_actor_forward_f32 failed with code {rc}: {env.lib.last_error(env.h).decode()}")
            return actions
        rc = env.lib.actor_forward(env.h, self._bf16_handle(env.lib), p(obs), p(actions), p(log_prob), p(loc_scale), int(seed), int(counter), int(bool(deterministic)))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_actor_forward failed with code {rc}: {env.lib.last_error(env.h).decode()}")
        return actions

    def rollout(self, env: SigmaEnv, n_steps: int, slab: torch.Tensor | None = None, log_prob: torch.Tensor | None = None,
                actions: torch.Tensor | None = None, seed: int = 0, counter0: int = 0, path_first: int | None = None, path_count: int | None = None,
                deterministic: bool = False, precision: str | None = None, slab_ptr: int | None = None):
        """``n_steps`` x (policy -> fused step + record + resets) enqueued back to back; optional records ``slab [T,B,W]``,
        ``log_prob [T,B,N]``, ``actions [T,B,N,2]`` (CUDA float32, contiguous).  ``slab_ptr``: the record target as a raw device address instead of ``slab`` (an env
        shard's first row inside a ``[T, B_total, W]`` buffer of the whole batch, with ``env.set_rollout_slab_stride(B_total * W)``).  ``precision`` (default: the actor's): "fp32" = the reference's
        arithmetic (``sigmaenv_rollout_f32``), "bf16" = the fast inference variant (``sigmaenv_rollout``)."""
        if path_first is None:
            path_first, path_count = env.default_paths()
        scratch = self._scratch_actions(env)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        p_slab = C.c_void_p(int(slab_ptr)) if slab_ptr else p(slab)
        if (precision or self.precision) == "fp32":
            lo, hi = self._keep[-2], self._keep[-1]
            rc = env.lib.rollout_f32(env.h, self._mlp32.handle(env.lib), lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p), p(self._scratch_out4(env)), int(n_steps),
                                      p(scratch), p_slab, p(log_prob), p(actions), int(seed), int(counter0), int(path_first), int(path_count),
                                      int(bool(deterministic)))
            if rc != 0:
                raise RuntimeError(f"sigmaenv_rollout_f32 failed with code {rc}: {env.lib.last_error(env.h).decode()}")
            return
        rc = env.lib.rollout(env.h, self._bf16_handle(env.lib), int(n_steps), p(scratch), p_slab, p(log_prob), p(actions), int(seed), int(counter0), int(path_first),
                              int(path_count), int(bool(deterministic)))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_rollout failed with code {rc}: {env.lib.last_error(env.h).decode()}")
