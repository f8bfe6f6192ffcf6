#!/usr/bin/env python
"""Diagnostic: wall-clock timeline of ONE single-step launch of the step kernel (100 MHz s_memrealtime stamps: comparable across XCDs) and where its
wavefronts sat (HW_ID / XCC_ID).  Needs `make -C sigmarl_amd/csrc prof_rt`.  Answers: how long is the dispatch ramp, when do the tiles reach phase R, how long is
the tail behind the last tile without a reset, and how the resets pile up per SIMD."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIGMAENV_TIMESTAMPS"] = "1"
os.environ.setdefault("SIGMAENV_LIB", os.path.join(ROOT, "sigmarl_amd", "csrc", "libsigmaenv_prof_rt.so"))
import numpy as np, torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B, N = int(os.environ.get("B", 4096)), int(os.environ.get("N", 16))
env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=B, device="cuda:0")
env.reset_random(seed=1)
acts = torch.rand((B, N, 2), device="cuda") * torch.tensor([1.0, 0.5], device="cuda") - torch.tensor([0.0, 0.25], device="cuda")
pf, pc = env.map.list_first[0], env.map.list_count[0]
f = env.lib.cdll.sigmaenv_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
for t in range(400):  # (sustained clocks)
    env.step_autoreset(acts, seed=1, counter=t, path_first=pf, path_count=pc)
env.sync()
rows = []
for rep in range(int(os.environ.get("REPS", 5))):
    env.step_autoreset(acts, seed=1, counter=400 + rep, path_first=pf, path_count=pc)
    env.sync()
    ts = np.zeros((B, 16), np.uint64)
    n = f(env.h, ts.ctypes.data_as(C.c_void_p), B)
    ts = ts[:n].astype(np.int64)
    rt0, rtR, rt1, hw = ts[:, 12], ts[:, 13], ts[:, 14], ts[:, 15]
    cyc = ts[:, :8]
    t0 = rt0.min()
    r_cycles = cyc[:, 7] - cyc[:, 6]
    is_reset = r_cycles > 3000
    # placement: HW_ID bits (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0] of the upper word
    hwid, xcc = hw & 0xFFFFFFFF, (hw >> 32) & 0xF
    simd_key = (xcc << 12) | (((hwid >> 13) & 7) << 9) | (((hwid >> 12) & 1) << 8) | (((hwid >> 8) & 0xF) << 4) | ((hwid >> 4) & 3)
    keys, inv, cnt = np.unique(simd_key, return_inverse=True, return_counts=True)
    resets_per_simd = np.bincount(inv, weights=is_reset.astype(float), minlength=len(keys))
    end_per_simd = np.zeros(len(keys)); np.maximum.at(end_per_simd, inv, (rt1 - t0).astype(float))
    tick = 0.01  # microseconds per tick
    us = lambda a: np.asarray(a) * tick
    out = {
        "tiles": int(n), "resets": int(is_reset.sum()), "simds_seen": int(len(keys)), "tiles_per_simd_max": int(cnt.max()), "xcc_by_block_mod8_consistent": bool((np.bincount((np.arange(n) // 4 % 8) * 16 + xcc, minlength=128).reshape(8, 16) > 0).sum() == 8),
        "start_us p50/p99/max": [round(float(x), 2) for x in us(np.percentile(rt0 - t0, [50, 99, 100]))],
        "reach_R_us p1/p50/p99/max": [round(float(x), 2) for x in us(np.percentile(rtR - t0, [1, 50, 99, 100]))],
        "end_us non-reset p50/max": [round(float(x), 2) for x in us(np.percentile((rt1 - t0)[~is_reset], [50, 100]))],
        "end_us reset p50/p90/max": [round(float(x), 2) for x in us(np.percentile((rt1 - t0)[is_reset], [50, 90, 100]))] if is_reset.any() else None,
        "R_us reset tiles p50/p90/max": [round(float(x), 2) for x in us(np.percentile((rt1 - rtR)[is_reset], [50, 90, 100]))] if is_reset.any() else None,
        "step_body_us p50 (start -> R)": round(float(np.median(rtR - rt0)) * tick, 2),
        "launch_us (first start -> last end)": round(float((rt1.max() - t0)) * tick, 2),
        "end_us by resets on the SIMD (mean of the SIMD's last end)": {int(k): round(float(end_per_simd[resets_per_simd == k].mean()) * tick, 2) for k in np.unique(resets_per_simd)},
        "simds by resets": {int(k): int((resets_per_simd == k).sum()) for k in np.unique(resets_per_simd)},
    }
    rows.append(out)
import json
for r in rows:
    print(json.dumps(r))
