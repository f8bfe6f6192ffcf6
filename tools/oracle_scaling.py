import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, oracle_binding as ob
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config
B, N = 4096, 16
mp = load_map("cpm_entire")
p = Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, rew_method="distance", dt=0.05)
env = ob.OracleEnv(make_config(p, mp, B), mp)
env.get(capi.BUF_DONE, copy=False)[:] = 1
env.auto_reset(0, 0, 0, 40)
rng = np.random.default_rng(0)
act = np.stack([rng.uniform(0, 1, (B, N)), rng.uniform(-0.25, 0.25, (B, N))], -1).astype(np.float32)
env.step(act); env.auto_reset(0, 1, 0, 40)
t0 = time.perf_counter(); k = 0
while time.perf_counter() - t0 < 3.0:
    env.step(act); t1 = time.perf_counter(); env.auto_reset(0, k + 2, 0, 40); k += 1
el = time.perf_counter() - t0
print("threads", os.environ.get("OMP_NUM_THREADS"), "steps", k, "agent-env-steps/s %.3e" % (B * N * k / el))
