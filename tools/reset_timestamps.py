#!/usr/bin/env python
"""Diagnostic: per-phase shader-clock cycles of the auto-reset kernel's active workgroups (needs SIGMAENV_TIMESTAMPS=2)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIGMAENV_TIMESTAMPS"] = "2"
os.environ.setdefault("SIGMAENV_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sigmarl_amd", "csrc", "libsigmaenv_prof.so"))  # profile build (make -C sigmarl_amd/csrc prof)
import numpy as np, torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B, N = int(os.environ.get("B", 4096)), int(os.environ.get("N", 16))
env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=B, device="cuda:0")
env.reset_random(seed=1)
acts = torch.rand((B, N, 2), device="cuda") * torch.tensor([1.0, 0.5], device="cuda") - torch.tensor([0.0, 0.25], device="cuda")
for t in range(20):
    env.step(acts); env.auto_reset(seed=1)
env.sync()
f = env.lib.cdll.sigmaenv_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
ts = np.zeros((B, 16), np.uint64)
n = f(env.h, ts.ctypes.data_as(C.c_void_p), B)
ts = ts[:n].astype(np.int64)
act = ts[:, 6] > 0
print("workgroups", n, "active", int(act.sum()))
t0 = ts[:, 0].min()
print("start spread (all wgs)", ts[:, 0].max() - t0, " last end", ts[act, 6].max() - t0)
a = ts[act]
for a0, a1, nm in ((0, 1, "flags + state load"), (1, 2, "sampler + placement"), (2, 5, "pair distances + env tail"), (5, 6, "observation")):
    d = a[:, a1] - a[:, a0]
    print(f"{nm:28s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p90 {np.percentile(d, 90):9.0f}")
print("active wg total mean", (a[:, 6] - a[:, 0]).mean())
