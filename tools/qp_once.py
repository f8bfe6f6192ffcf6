"""A few centralized CBF-QP launches on a 16 x 2048 CPM batch (the product library): the workload of the PMC passes of tools/pmc_qp.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B, N = int(os.environ.get("B", 2048)), int(os.environ.get("N", 16))
p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="cbf", is_solve_qp=os.environ.get("QP", "1") == "1", is_using_cbf_training=True,
               is_apply_mask=False, is_obs_noise=False, max_steps=128)
env = SigmaEnv(p, n_envs=B, device="cuda:0"); env.reset_random(seed=1); env.cbf_attach()
g = torch.Generator(device="cuda").manual_seed(0)
act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.2 - 0.1, torch.rand(B, N, generator=g, device="cuda") * 0.8 - 0.4], dim=-1).contiguous()
for _ in range(20):
    env.step_autoreset(act, seed=1)
u = torch.zeros((B, N, 2), dtype=torch.float64, device="cuda"); info = torch.zeros((B, 2), dtype=torch.int32, device="cuda")
for _ in range(int(os.environ.get("QP_CALLS", 8))):
    if p.is_solve_qp: env.cbf_qp(act, None, u, info)
    else: env.cbf_rewards(act)
    env.sync()
print("iterations mean %.2f" % info[:, 0].float().mean().item())
