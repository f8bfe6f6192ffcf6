"""The policy (actor) forward pass and whole rollouts on the device.

``Actor`` wraps ``sigmaenv_actor_*`` / ``sigmaenv_rollout`` of the C-ABI: the actor network of
``sigmarl/modules/decision_making_module.py:34-82`` (torchrl ``MultiAgentMLP(depth=3, num_cells=256, activation=Tanh, share_params=True)``
+ ``NormalParamExtractor`` + ``TanhNormal``) as one MFMA kernel (bf16 weights / activations, fp32 accumulation).  Weights come from any
``torch.nn.Sequential`` of four ``Linear`` layers (the parameter layout torchrl's shared-parameter MLP has), or from plain arrays.
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


class Actor:
    def __init__(self, mlp: torch.nn.Module, low, high, lib: capi.Library | None = None):
        self.lib = lib or capi.load_library()
        lin = [m for m in mlp.modules() if isinstance(m, torch.nn.Linear)]
        if len(lin) != 4 or lin[1].in_features != 256 or lin[2].out_features != 256 or lin[3].out_features != 4:
            raise ValueError("expected Linear(D,256), Linear(256,256), Linear(256,256), Linear(256,4)")
        self.obs_dim = lin[0].in_features
        arrs = []
        for m in lin:
            arrs.append(np.ascontiguousarray(m.weight.detach().cpu().numpy(), np.float32))
            arrs.append(np.ascontiguousarray(m.bias.detach().cpu().numpy(), np.float32))
        self._keep = arrs + [np.ascontiguousarray(low, np.float32), np.ascontiguousarray(high, np.float32)]
        h = C.c_void_p()
        rc = self.lib.actor_create(self.obs_dim, *[a.ctypes.data_as(C.c_void_p) for a in self._keep], C.byref(h))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_actor_create failed with code {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.actor_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def forward(self, env: SigmaEnv, actions: torch.Tensor, log_prob: torch.Tensor | None = None, loc_scale: torch.Tensor | None = None,
                obs: torch.Tensor | None = None, seed: int = 0, counter: int = 0, deterministic: bool = False):
        """actions[B,N,2] := policy(env.obs or ``obs``); enqueued on the env's stream."""
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = self.lib.actor_forward(env.h, self.h, p(obs), p(actions), p(log_prob), p(loc_scale), int(seed), int(counter), int(bool(deterministic)))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_actor_forward failed with code {rc}: {env.lib.last_error(env.h).decode()}")
        return actions

    def rollout(self, env: SigmaEnv, n_steps: int, slab: torch.Tensor | None = None, log_prob: torch.Tensor | None = None,
                actions: torch.Tensor | None = None, seed: int = 0, counter0: int = 0, path_first: int | None = None, path_count: int | None = None,
                deterministic: bool = False):
        """``n_steps`` x (policy -> fused step + record + resets) enqueued back to back; optional records ``slab [T,B,W]``,
        ``log_prob [T,B,N]``, ``actions [T,B,N,2]`` (CUDA float32, contiguous)."""
        if path_first is None:
            path_first, path_count = env.map.list_first[0], env.map.list_count[0]
        if not hasattr(self, "_scratch") or self._scratch.shape[0] != env.B or self._scratch.device != env.device:
            self._scratch = torch.zeros((env.B, env.N, 2), dtype=torch.float32, device=env.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        rc = self.lib.rollout(env.h, self.h, int(n_steps), p(self._scratch), p(slab), p(log_prob), p(actions), int(seed), int(counter0), int(path_first),
                              int(path_count), int(bool(deterministic)))
        if rc != 0:
            raise RuntimeError(f"sigmaenv_rollout failed with code {rc}: {env.lib.last_error(env.h).decode()}")
