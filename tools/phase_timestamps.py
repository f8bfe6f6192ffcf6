#!/usr/bin/env python
"""Diagnostic: per-phase shader-clock cycles of the wave-per-tile step kernel.  Needs the profile build
(make -C sigmarl_amd/csrc prof; SIGMAENV_LIB is set here) -- the product library carries no stamps."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIGMAENV_TIMESTAMPS"] = "1"
os.environ.setdefault("SIGMAENV_LIB", os.path.join(ROOT, "sigmarl_amd", "csrc", "libsigmaenv_prof.so"))
import numpy as np, torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B, N = int(os.environ.get("B", 4096)), int(os.environ.get("N", 16))
scen = os.environ.get("SCENARIO", "cpm_entire")
import json
extra = json.loads(os.environ.get("PARAMS", "{}"))  # e.g. PARAMS='{"is_ego_view": false}': the non-default observation rows
env = SigmaEnv(Parameters(**dict(dict(n_agents=N, scenario_type=scen, is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), **extra)), n_envs=B, device="cuda:0")
env.reset_random(seed=1)
acts = torch.rand((B, N, 2), device="cuda") * torch.tensor([1.0, 0.5], device="cuda") - torch.tensor([0.0, 0.25], device="cuda")
pf, pc = env.map.list_first[0], env.map.list_count[0]
for t in range(20):
    env.step_autoreset(acts, seed=1, counter=t, path_first=pf, path_count=pc)
env.sync()
f = env.lib.cdll.sigmaenv_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
ts = np.zeros((B, 16), np.uint64)
n = f(env.h, ts.ctypes.data_as(C.c_void_p), B)
sub = ts[:n, [1, 10, 11, 2]].astype(np.int64)  # scan: start, end of S1 (candidate masks), end of S2 (work-list rounds), end (S3)
stats = ts[:n, 8:10].astype(np.int64)
lvl = ts[:n, 12:16].astype(np.int64)
ts = ts[:n, :8].astype(np.int64)
ts = ts[ts[:, 0] > 0]
d = np.diff(ts, axis=1)
names = ["A dynamics", "S scan", "E edges", "B1 pairs", "C reward", "D obs+slab", "R resets"]
print("tiles", len(ts), "(shader-clock cycles per tile and phase; the clocks of different XCDs are not comparable, only differences within a tile)")
for k, nm in enumerate(names):
    print(f"{nm:12s} mean {d[:, k].mean():9.0f}  p10 {np.percentile(d[:, k], 10):9.0f}  p90 {np.percentile(d[:, k], 90):9.0f}")
print("tile total mean", (ts[:, 7] - ts[:, 0]).mean())
if stats[:, 1].sum():  # accumulated over the launches above
    per = stats[:, 0] / np.maximum(stats[:, 1], 1)
    print("scan work list: items per tile-step mean %.1f p10 %.0f p50 %.0f p90 %.0f max %.0f; rounds per step %.2f" % (per.mean(), np.percentile(per, 10), np.percentile(per, 50), np.percentile(per, 90), per.max(), stats[:, 1].sum() / (20.0 * len(stats))))
sub = sub[(sub[:, 0] > 0) & (sub[:, 1] > 0)]
if len(sub):
    print("scan split: S1 candidate masks %.0f, S2 work-list rounds %.0f, S3 results %.0f" % ((sub[:, 1] - sub[:, 0]).mean(), (sub[:, 2] - sub[:, 1]).mean(), (sub[:, 3] - sub[:, 2]).mean()))
if lvl.sum():
    steps = np.maximum(stats[:, 1], 1)
    print("scan neighbour levels: tasks beyond the near level per tile-step %.2f, tile-steps with such a task %.3f, tasks at the tight level per tile-step %.1f" % ((lvl[:, 0] / steps).mean(), (lvl[:, 1] / steps).mean(), (lvl[:, 2] / steps).mean()))
    print("  of those: the first agent's (stale corner queries) %.2f per tile-step" % ((lvl[:, 3] / steps).mean()))
