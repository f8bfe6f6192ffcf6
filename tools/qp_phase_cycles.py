import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SIGMAENV_CBF_DEBUG_SKIP", "128")
os.environ.setdefault("SIGMAENV_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sigmarl_amd", "csrc", "libsigmaenv_prof.so"))  # profile build (make -C sigmarl_amd/csrc prof)
import torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B,N=4096,16
p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_apply_mask=False, is_obs_noise=False, max_steps=128)
env = SigmaEnv(p, n_envs=B, device="cuda:0"); env.reset_random(seed=1); env.cbf_attach()
g = torch.Generator(device="cuda").manual_seed(0)
act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.2 - 0.1, torch.rand(B, N, generator=g, device="cuda") * 0.8 - 0.4], dim=-1).contiguous()
for _ in range(20):
    env.step_autoreset(act, seed=1)
u = torch.zeros((B,N,2), dtype=torch.float64, device="cuda"); info = torch.zeros((B,2), dtype=torch.int32, device="cuda")
env.cbf_qp(act, None, u, info); env.sync()
d = u.reshape(B,-1)[:, :7].cpu()
it = info[:,0].float().cpu()
print("cycles (x100MHz shader clock?) mean total %.0f eval %.0f chol %.0f ls %.0f ; iters mean %.2f" % (d[:,0].mean(), d[:,1].mean(), d[:,2].mean(), d[:,3].mean(), it.mean()))
print("per iteration: eval %.0f chol %.0f ls %.0f ; outside loop %.0f" % ((d[:,1]/it).mean(), (d[:,2]/it).mean(), (d[:,3]/it).mean(), (d[:,0]-d[:,1]-d[:,2]-d[:,3]).mean()))
print("before the Newton loop: load %.0f stencil phase %.0f lane + candidate rows %.0f" % (d[:,4].mean(), d[:,5].mean(), d[:,6].mean()))
