#!/usr/bin/env python
"""Diagnostic: per-phase shader-clock cycles of the fused step kernel (needs SIGMAENV_TIMESTAMPS=1)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SIGMAENV_TIMESTAMPS"] = "1"
import numpy as np, torch
from sigmarl_amd import capi
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
ts8 = ts[:n].astype(np.int64)
print("A split: loads+bicycle %.0f  vertices+stores %.0f  mask stage 2 %.0f" % ((ts8[:, 6] - ts8[:, 0]).mean(), (ts8[:, 7] - ts8[:, 6]).mean(), (ts8[:, 1] - ts8[:, 7]).mean()))
print("C split: reward terms %.0f  short-term path %.0f  stores+done %.0f" % ((ts8[:, 14] - ts8[:, 3]).mean(), (ts8[:, 15] - ts8[:, 14]).mean(), (ts8[:, 4] - ts8[:, 15]).mean()))
o = ts8[:, 8:14]
print("D split: topk %.0f  sync %.0f  pass1 %.0f  pass2+3 %.0f  sync %.0f  stores %.0f" % tuple([(ts8[:, 8] - ts8[:, 4]).mean()] + [(o[:, k + 1] - o[:, k]).mean() for k in range(5)]))
ts = ts[:n, :6].astype(np.int64)
d = np.diff(ts, axis=1)
names = ["A dynamics", "B1 pairs", "B2 scan", "C reward", "D obs"]
print("groups", n, "kernel span (cycles)", ts[:, 5].max() - ts[:, 0].min())
for k, nm in enumerate(names):
    print(f"{nm:12s} mean {d[:, k].mean():9.0f}  p10 {np.percentile(d[:, k], 10):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f}")
print("block total mean", (ts[:, 5] - ts[:, 0]).mean(), "start spread", ts[:, 0].max() - ts[:, 0].min())
