#!/usr/bin/env python
"""Diagnostic: shader-clock stamps of the fp32 MLP kernel's phases per workgroup (profile build: make -C sigmarl_amd/csrc prof)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIGMAENV_TIMESTAMPS"] = "1"
os.environ.setdefault("SIGMAENV_LIB", os.path.join(ROOT, "sigmarl_amd", "csrc", "libsigmaenv_prof.so"))
import numpy as np, torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
from sigmarl_amd.actor import Actor, make_mlp
B, N = 4096, 16
env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=B, device="cuda:0")
env.reset_random(seed=1)
torch.manual_seed(0)
actor = Actor(make_mlp(env.D), low=[-1.0, -0.6109], high=[1.0, 0.6109], precision="fp32")
act = torch.zeros((B, N, 2), device="cuda")
for t in range(3):
    actor.forward(env, act, seed=1, counter=t)
env.sync()
f = env.lib.cdll.sigmaenv_mlp32_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
G = B * N // 64
ts = np.zeros((G, 16), np.uint64)
mh = actor._mlp32.handle(env.lib)
n = f(mh, ts.ctypes.data_as(C.c_void_p), G)
ts = ts[:n].astype(np.int64)
names = ["input staging", "L0 multiply+tanh", "L0 wait (barrier 1)", "L0 store+barrier 2", "L1 multiply+tanh", "L1 wait", "L1 store+barrier", "L2 multiply+tanh", "L2 wait", "L2 store+barrier",
         "out multiply", "out reduce+store"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13]
d = np.diff(ts[:, idx], axis=1)
print("workgroups", n, "(wavefront 0 of every workgroup; shader-clock cycles)")
for k, nm in enumerate(names):
    print(f"{nm:22s} mean {d[:, k].mean():8.0f}  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}")
tot = ts[:, 13] - ts[:, 0]
print("workgroup total mean %.0f p90 %.0f; first start to last end over all workgroups %.0f" % (tot.mean(), np.percentile(tot, 90), ts[:, 13].max() - ts[:, 0].min()))
hw = ts[:, 14]; xcc = ts[:, 15] & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
key = xcc * 4096 + se * 256 + sh * 16 + cu
import collections
groups = collections.defaultdict(list)
for b, k in enumerate(key): groups[int(k)].append(b)
print("distinct (xcc, se, sh, cu):", len(groups), "; workgroups per CU:", sorted(collections.Counter(len(v) for v in groups.values()).items()))
first = sorted(groups.items())[:6]
for k, v in first: print("  cu key", k, "blocks", v, "starts", [int(ts[b, 0] - ts[v[0], 0]) for b in v])
