#!/usr/bin/env python
"""Timing / accuracy of the on-device actor and of a whole rollout (policy + fused step) at the bench size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sigmarl_amd.actor import Actor, make_mlp
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters

B, N = int(os.environ.get("B", 4096)), 16
env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="distance", is_apply_mask=False,
                          is_obs_noise=False, max_steps=128), n_envs=B, device="cuda:0")
env.reset_random(seed=1)
torch.manual_seed(0)
mlp = make_mlp(env.D)
MODE = os.environ.get("MODE", "split")  # arithmetic of the fp32 network: split (fp16 hi + lo, default) | exact (fp32 MFMA)
actor = Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], mode=MODE)
print("mode:", MODE)
act = torch.zeros((B, N, 2), device="cuda")
lp = torch.zeros((B, N), device="cuda")
ls = torch.zeros((B, N, 4), device="cuda")
actor.forward(env, act, lp, ls)
env.sync()
with torch.no_grad():
    ref = mlp(env.obs.reshape(-1, env.D).cpu()).numpy()
got = ls.reshape(-1, 4).cpu().numpy()
print("loc vs torch fp32: max %.3e mean %.3e" % (np.abs(got[:, :2] - ref[:, :2]).max(), np.abs(got[:, :2] - ref[:, :2]).mean()))
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
stream = torch.cuda.current_stream()
torch.cuda.synchronize()
s.record()
for t in range(200):
    actor.forward(env, act, lp, None, seed=1, counter=t)
e.record(); torch.cuda.synchronize()
print("actor forward: %.2f us per call (%d rows)" % (s.elapsed_time(e) / 200 * 1e3, B * N))
from sigmarl_amd import capi
env.kernel_time_ms(capi.KERNEL_MLP32)  # arm the HIP-event brackets, then time the fp32 MLP kernel alone
for t in range(64):
    actor.forward(env, act, lp, None, seed=1, counter=t)
ms, n = env.kernel_time_ms(capi.KERNEL_MLP32)
flops = 2.0 * B * N * (32 * 256 + 2 * 256 * 256 + 256 * 4)
print("fp32 MLP kernel alone: %.2f us per launch over %d brackets = %.1f TFLOP/s of fp32-equivalent products = %.1f %% of the 157.3 TFLOP/s fp32 matrix peak"
      % (ms * 1e3, n, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 157.3e12 * 100))
if MODE == "split":
    print("  split mode issues 3 fp16 matrix products per fp32 product: %.1f TFLOP/s on the fp16 pipe = %.1f %% of its 2.5 PFLOP/s" % (3 * flops / (ms * 1e-3) / 1e12, 3 * flops / (ms * 1e-3) / 2.5e15 * 100))
T = 256
for name, kw in (("rollout (policy + step)", {}),):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    actor.rollout(env, T, seed=1, counter0=1000)
    env.sync(); el = time.perf_counter() - t0
    print("%s: %.4f ms per step, %.4g agent-env-steps/s" % (name, el / T * 1e3, B * N * T / el))
