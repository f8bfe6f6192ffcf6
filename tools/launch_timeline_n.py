#!/usr/bin/env python
"""Diagnostic: wall-clock spread of the wavefronts of ONE T-step launch of the step kernel (100 MHz s_memrealtime stamps, comparable across XCDs): when each wavefront
starts the launch and when it ends its last step, per SIMD (HW_ID / XCC_ID).  Needs `make -C sigmarl_amd/csrc prof_rt`.  Answers: how much of a launch is the drain at its
end (SIMDs running fewer than four wavefronts because the others are done) -- the part a larger batch amortises over several rounds."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIGMAENV_TIMESTAMPS"] = "1"
os.environ.setdefault("SIGMAENV_LIB", os.path.join(ROOT, "sigmarl_amd", "csrc", "libsigmaenv_prof_rt.so"))
import numpy as np, torch
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters
B, N, T = int(os.environ.get("B", 4096)), int(os.environ.get("N", 16)), int(os.environ.get("T", 32))
env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=B, device="cuda:0")
env.reset_random(seed=1)
acts = (torch.rand((T, B, N, 2), device="cuda") * torch.tensor([1.0, 0.5], device="cuda") - torch.tensor([0.0, 0.25], device="cuda")).contiguous()
pf, pc = env.map.list_first[0], env.map.list_count[0]
f = env.lib.cdll.sigmaenv_debug_timestamps
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
for k in range(12):  # (sustained clocks)
    env.step_autoreset_n(acts, seed=1, counter0=k * T, path_first=pf, path_count=pc)
env.sync()
for rep in range(int(os.environ.get("REPS", 4))):
    env.step_autoreset_n(acts, seed=1, counter0=(12 + rep) * T, path_first=pf, path_count=pc)
    env.sync()
    ts = np.zeros((B, 16), np.uint64)
    n = f(env.h, ts.ctypes.data_as(C.c_void_p), B)
    ts = ts[:n].astype(np.int64)
    rt0, rt1, hw = ts[:, 12], ts[:, 14], ts[:, 15]
    t0 = rt0.min(); tick = 0.01
    hwid, xcc = hw & 0xFFFFFFFF, (hw >> 32) & 0xF
    simd_key = (xcc << 12) | (((hwid >> 13) & 7) << 9) | (((hwid >> 12) & 1) << 8) | (((hwid >> 8) & 0xF) << 4) | ((hwid >> 4) & 3)
    keys, inv, cnt = np.unique(simd_key, return_inverse=True, return_counts=True)
    end = (rt1 - t0) * tick; start = (rt0 - t0) * tick
    span = float(end.max())
    # per SIMD: wave-slot-time used vs available until the launch ends (4 slots per SIMD with this kernel's registers)
    busy = np.zeros(len(keys)); np.add.at(busy, inv, end - start)
    last = np.zeros(len(keys)); np.maximum.at(last, inv, end)
    first_end = np.full(len(keys), 1e9); np.minimum.at(first_end, inv, end)
    # time during which k wavefronts of the launch are alive (all SIMDs together)
    ev = np.sort(end); alive_time = {}
    out = {"tiles": int(n), "simds": int(len(keys)), "tiles_per_simd min/max": [int(cnt.min()), int(cnt.max())],
           "start_us p50/max": [round(float(np.median(start)), 1), round(float(start.max()), 1)],
           "wave duration us p1/p50/p99/max": [round(float(x), 1) for x in np.percentile(end - start, [1, 50, 99, 100])],
           "wave end us p1/p10/p50/p90/max": [round(float(x), 1) for x in np.percentile(end, [1, 10, 50, 90, 100])],
           "launch span us": round(span, 1),
           "mean wave-slot occupancy over the span (sum of wave lifetimes / (4 slots x SIMDs x span))": round(float((end - start).sum() / (4 * len(keys) * span)), 4),
           "SIMD: first wave done us p50": round(float(np.median(first_end)), 1), "SIMD: last wave done us p50/p90/max": [round(float(x), 1) for x in np.percentile(last, [50, 90, 100])],
           "by XCC: mean end us": {int(x): round(float(end[xcc == x].mean()), 1) for x in np.unique(xcc)},
           "by wave slot (HW_ID[3:0]): count, mean end us": {int(x): [int(((hwid & 0xF) == x).sum()), round(float(end[(hwid & 0xF) == x].mean()), 1)] for x in np.unique(hwid & 0xF)},
           "by wave index in its workgroup (tile mod 4): mean end us": {int(x): round(float(end[np.arange(n) % 4 == x].mean()), 1) for x in range(4)},
           "by rank of the tile index among the SIMD's four: mean end us": None}
    # rank of each wave among the waves of its SIMD by tile index (dispatch order)
    order = np.lexsort((np.arange(n), inv)); rank = np.zeros(n, np.int64); rank[order] = np.arange(n) % 1  # placeholder
    pos = np.zeros(n, np.int64); seen = {}
    for t_ in order.tolist():
        k_ = int(inv[t_]); pos[t_] = seen.get(k_, 0); seen[k_] = pos[t_] + 1
    out["by rank of the tile index among the SIMD's four: mean end us"] = {int(x): round(float(end[pos == x].mean()), 1) for x in range(int(pos.max()) + 1)}
    # where the spread between SIMDs comes from: the SIMD's last end by XCC / CU / SIMD
    cu_key = simd_key >> 2
    lastw = last[inv]
    import collections
    def spread(keyarr):
        ks, iv = np.unique(keyarr, return_inverse=True)
        m = np.zeros(len(ks)); c = np.zeros(len(ks)); np.add.at(m, iv, lastw); np.add.at(c, iv, 1); m /= c
        return float(m.std()), float(m.max() - m.min())
    out["std / range of the mean last-end us by XCC, by CU, by SIMD"] = [[round(x, 1) for x in spread(xcc)], [round(x, 1) for x in spread(cu_key)], [round(x, 1) for x in spread(simd_key)]]
    out["SIMD index within the CU: mean last-end us"] = {int(x): round(float(last[(keys & 3) == x].mean()), 1) for x in range(4)}
    print(json.dumps(out))
