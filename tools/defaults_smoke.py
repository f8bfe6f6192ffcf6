"""The reference's DEFAULT Parameters() through the VMAS surface (is_apply_mask, observation noise, mtv distances): a six-step smoke."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sigmarl_amd.params import Parameters
from sigmarl_amd.scenario import make_scenario
p = Parameters()
print({k: getattr(p, k) for k in ("scenario_type", "n_agents", "is_apply_mask", "is_obs_noise", "is_use_mtv_distance", "rew_method", "dt")})
sc = make_scenario(p)
w = sc.env_make_world(16, "cuda:0", n_agents=p.n_agents)
sc.env_reset_world_at(None)
for t in range(6):
    for a in w.agents:
        a.action.u = torch.rand((16, 2), device="cuda") * torch.tensor([1.0, 0.5], device="cuda")
    w.step()
    r = [sc.reward(a) for a in w.agents]; o = [sc.observation(a) for a in w.agents]; i = [sc.info(a) for a in w.agents]; d = sc.done()
    for e in torch.nonzero(d).flatten().tolist(): sc.env_reset_world_at(e)
print("ok", o[0].shape, float(torch.stack(r).mean()))
