import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SIGMAENV_CBF_DEBUG_SKIP", "128")
os.environ.setdefault("SIGMAENV_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sigmarl_amd", "csrc", "libsigmaenv_prof.so"))  # profile build (make -C sigmarl_amd/csrc prof)
import torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B,N=int(os.environ.get("B", 2048)),16
p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=False, rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_apply_mask=False, is_obs_noise=False, max_steps=128)
env = SigmaEnv(p, n_envs=B, device="cuda:0"); env.reset_random(seed=1); env.cbf_attach()
g = torch.Generator(device="cuda").manual_seed(0)
act = torch.stack([torch.rand(B, N, generator=g, device="cuda") * 1.2 - 0.1, torch.rand(B, N, generator=g, device="cuda") * 0.8 - 0.4], dim=-1).contiguous()
for _ in range(20):
    env.step_autoreset(act, seed=1)
u = torch.zeros((B,N,2), dtype=torch.float64, device="cuda"); info = torch.zeros((B,2), dtype=torch.int32, device="cuda")
for _ in range(int(os.environ.get("QP_CALLS", 1))):  # (QP_CALLS > 1: repeated solves of the same problem)
    env.cbf_qp(act, None, u, info); env.sync()
d = u.reshape(B,-1)[:, :22].cpu()
it = info[:,0].float().cpu()
print("cycles (x100MHz shader clock?) mean total %.0f eval %.0f chol %.0f ls %.0f ; iters mean %.2f" % (d[:,0].mean(), d[:,1].mean(), d[:,2].mean(), d[:,3].mean(), it.mean()))
print("per iteration: eval %.0f chol %.0f ls %.0f ; outside loop %.0f" % ((d[:,1]/it).mean(), (d[:,2]/it).mean(), (d[:,3]/it).mean(), (d[:,0]-d[:,1]-d[:,2]-d[:,3]).mean()))
print("before the Newton loop: load %.0f stencil phase %.0f lane + candidate rows %.0f" % (d[:,4].mean(), d[:,5].mean(), d[:,6].mean()))
print("candidate pair rows per env: mean %.1f p50 %.0f p90 %.0f max %.0f (of %d pair rows); lane rows %d" % (d[:,7].mean(), d[:,7].median(), d[:,7].quantile(0.9), d[:,7].max(), N*(N-1)//2*9, N*3*2))
z = d[:,7] == 0
print("pairs whose rows were evaluated (of %d): mean %.1f p50 %.0f p90 %.0f max %.0f" % (N*(N-1)//2, d[:,21].mean(), d[:,21].median(), d[:,21].quantile(0.9), d[:,21].max()))
print("stencil phase of the Newton wavefront: set-up %.0f, stage 1 %.0f, stage 2 %.0f, bound %.0f, later passes %.0f, conversion + barrier %.0f" % tuple(d[:,k].mean() for k in range(15, 21)))
print("between the stencil phase and the Newton phase: lane rows %.0f, candidate flags %.0f, compaction %.0f; register path %.0f (without / with candidates %.0f / %.0f), output phase %.0f" % (d[:,10].mean(), d[:,11].mean(), d[:,12].mean(), d[:,1].mean(), d[z,1].mean(), d[~z,1].mean(), d[:,13].mean()))
print("envs without candidate pair rows: %.1f %%; Newton phase cycles (loop + outputs): those %.0f, the rest %.0f; iterations %.2f / %.2f" % (100*z.float().mean(), d[z,0].mean(), d[~z,0].mean(), it[z].mean(), it[~z].mean()))
t0 = d[:,8].min(); st = (d[:,8]-t0)*0.01; en = (d[:,9]-t0)*0.01
print("wall clock (us, 100 MHz): workgroup start p50 %.1f p99 %.1f max %.1f; duration p10 %.1f p50 %.1f p90 %.1f max %.1f; last end %.1f" % (st.median(), st.quantile(0.99), st.max(), (en-st).quantile(0.1), (en-st).median(), (en-st).quantile(0.9), (en-st).max(), en.max()))
first = st < 5.0
print("first-round workgroups (start < 5 us): %d, duration p50 %.1f; later ones: duration p50 %.1f; durations of envs without / with candidates p50 %.1f / %.1f" % (int(first.sum()), (en-st)[first].median(), (en-st)[~first].median(), (en-st)[z].median(), (en-st)[~z].median()))
order = torch.argsort(en - st, descending=True)[:8]
for k in order.tolist():
    print("  env %4d: duration %.1f us (start %.1f), candidates %d, iterations %d, cycles total %.0f eval %.0f chol %.0f ls %.0f, before the loop %.0f" % (k, float(en[k]-st[k]), float(st[k]), int(d[k,7]), int(it[k]), d[k,0], d[k,1], d[k,2], d[k,3], d[k,4]+d[k,5]+d[k,6]))
print("iterations histogram:", torch.bincount(it.long()).tolist())
print("candidates histogram:", torch.bincount(d[:,7].long()).tolist())
reg = d[:,1] > 0
for nm, msk in (("without candidates", reg & z), ("with candidates", reg & ~z)):
    if msk.any():
        print("register path, envs %s: evaluations %.2f per solve (%.2f per iteration), cycles per evaluation %.0f, per direction %.0f (iterations %.2f); solve %.0f" % (nm, d[msk,14].mean(), (d[msk,14]/it[msk]).mean(), (d[msk,2]/d[msk,14]).mean(), (d[msk,3]/it[msk]).mean(), it[msk].mean(), d[msk,1].mean()))
k = int(order[0]); print("slowest env: evaluations %d, cycles in evaluations %.0f, in directions %.0f, iterations %d" % (int(d[k,14]), d[k,2], d[k,3], int(it[k])))
