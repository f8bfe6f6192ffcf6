#!/usr/bin/env python
"""Timing of the CBF margin-reward launch at the bench size, beside the CPU oracle on a bounded sample of the same states."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from sigmarl_amd import capi, cbf
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters, make_config

B, N = int(os.environ.get("B", 4096)), int(os.environ.get("N", 16))
p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="cbf", is_solve_qp=False,
               is_using_cbf_training=True, is_apply_mask=False, is_obs_noise=False, max_steps=128)
env = SigmaEnv(p, n_envs=B, device="cuda:0")
env.reset_random(seed=1)
env.cbf_attach()
g = torch.Generator(device="cuda").manual_seed(0)
act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.2 - 0.1, torch.rand(B, N, generator=g, device="cuda") * 0.8 - 0.4], dim=-1).contiguous()
for _ in range(20):  # move the vehicles off their start poses
    env.cbf_rewards(act)
    env.step_autoreset(act, seed=1)
env.sync()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
R = 50
torch.cuda.synchronize(); s.record()
for _ in range(R):
    env.cbf_rewards(act)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / R
print("cbf_rewards: %.4f ms per launch (%d envs x %d agents): %.4g agent-env margins/s" % (ms, B, N, B * N / ms * 1e3))
s.record()
for _ in range(R):
    env.cbf_rewards(act)
    env.step_autoreset(act, seed=1)
e.record(); torch.cuda.synchronize()
ms2 = s.elapsed_time(e) / R
print("cbf_rewards + fused step: %.4f ms per step: %.4g agent-env-steps/s" % (ms2, B * N / ms2 * 1e3))
ri = env.buffer(capi.BUF_REWARD_INFO)
print("violated channels (left, right, pair): %d %d %d of %d" % ((ri[5] < 0).sum().item(), (ri[6] < 0).sum().item(), (ri[4] < 0).sum().item(), B * N))
if os.environ.get("CPU", "1") != "0":
    import oracle_binding as ob
    Bs = int(os.environ.get("CPU_ENVS", 512))
    cfg = make_config(p, env.map, Bs)
    ora = ob.OracleEnv(cfg, env.map)
    sl, sr = cbf.load_segment_tables(env.map)
    ora.cbf_attach(cbf.make_cbf_config(p), sl, sr)
    st = env.buffer(capi.BUF_STATE)[:Bs].cpu().numpy()
    pa = env.buffer(capi.BUF_PATH)[:Bs].cpu().numpy()
    ora.reset(np.repeat(np.arange(Bs), N), np.tile(np.arange(N), Bs), pa.reshape(-1, 4), st.reshape(-1, 8), 1)
    a = act[:Bs].cpu().numpy()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 5.0:
        ora.cbf_rewards(a, want_margins=False); reps += 1
    el = (time.perf_counter() - t0) / reps
    print("CPU oracle (OpenMP, %d threads): %.2f ms per %d envs: %.4g agent-env margins/s" % (os.cpu_count(), el * 1e3, Bs, Bs * N / el))
    env.cbf_rewards(act); env.sync()
    d = np.abs(env.buffer(capi.BUF_REWARD_INFO)[4:7, :Bs].cpu().numpy() - ora.get(capi.BUF_REWARD_INFO)[4:7])
    print("HIP vs oracle reward channels on the sample: max |diff| %.3g" % d.max())
    if os.environ.get("QP", "1") != "0":
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < 5.0:
            ora.cbf_qp(a); reps += 1
        el = (time.perf_counter() - t0) / reps
        print("CPU oracle QP (OpenMP, %d threads): %.2f ms per %d envs: %.4g env QPs/s" % (os.cpu_count(), el * 1e3, Bs, Bs / el))
if os.environ.get("QP", "1") != "0":
    u = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
    safe = torch.zeros((B, N, 2), device="cuda")
    env.cbf_qp(act, safe, u, info); env.sync()
    torch.cuda.synchronize(); s.record()
    for _ in range(10):
        env.cbf_qp(act, safe, u, info)
    e.record(); torch.cuda.synchronize()
    msq = s.elapsed_time(e) / 10
    it = info[:, 0].float()
    print("cbf_qp: %.3f ms per launch (%d envs x %d agents): %.4g env QPs/s; Newton iterations mean %.1f max %d; converged %d of %d; actions changed in %d envs"
          % (msq, B, N, B / msq * 1e3, it.mean().item(), int(it.max().item()), int(info[:, 1].sum().item()), B,
             int(((safe - act.clamp(min=torch.tensor([-0.5, -0.5411], device="cuda"), max=torch.tensor([1.0, 0.5411], device="cuda"))).abs().amax(dim=(1, 2)) > 1e-5).sum().item())))
