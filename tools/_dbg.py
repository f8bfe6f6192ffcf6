import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_parity as tp
from sigmarl_amd import capi, cbf
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config
mp = load_map("cpm_entire")
import os
for (N, m) in [tuple(int(v) for v in os.environ.get('NM', '9,6').split(','))]:
    p = Parameters(n_agents=N, scenario_type="cpm_entire", dt=0.05, rew_method="cbf_sparse", is_solve_qp=True, is_using_cbf_training=True, is_obs_noise=False,
                   is_apply_mask=False, nom_controller_type="rl", adaptive_lambda=True, n_circles_approximate_vehicle=3, is_use_mtv_distance=False, max_steps=20,
                   is_grouping_agents=True, max_group_size=m, observation_range=0.5)
    cfg = make_config(p, mp, 33)
    dev = tp._hip_env(cfg, mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    dev.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
    dev.env.buffer(capi.BUF_DONE).fill_(1)
    dev.auto_reset(5, 0, int(mp.list_first[0]), int(mp.list_count[0]))
    act = np.random.default_rng(0).uniform(-0.3, 1.2, (33, N, 2)).astype(np.float32)
    print("N", N, "m", m, "solving", flush=True)
    out = dev.cbf_qp(act)
    print("  ok iters", out[2][:, 0].max(), "conv", out[2][:, 1].all(), dev.cbf_groups()[0], flush=True)
    dev.close()
