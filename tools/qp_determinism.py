import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B,N=2048,16
p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_apply_mask=False, is_obs_noise=False)
env = SigmaEnv(p, n_envs=B, device="cuda:0"); env.reset_random(seed=1); env.cbf_attach()
g = torch.Generator(device="cuda").manual_seed(0)
act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.2 - 0.1, torch.rand(B, N, generator=g, device="cuda") * 0.8 - 0.4], dim=-1).contiguous()
for _ in range(10): env.step_autoreset(act, seed=1)
us=[]
for r in range(6):
    u = torch.zeros((B,N,2), dtype=torch.float64, device="cuda"); env.cbf_qp(act, None, u, None); env.sync(); us.append(u.clone())
print("bitwise identical across 6 launches:", all(torch.equal(us[0], x) for x in us[1:]), "max diff", max(float((us[0]-x).abs().max()) for x in us[1:]))
