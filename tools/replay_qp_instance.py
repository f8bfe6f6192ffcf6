#!/usr/bin/env python
"""Diagnostic: replays the instance tools/fuzz_cbf.py saved (gpurun_out/fuzz_cbf_fail.npz: every env of the failing call) through the HIP library named by SIGMAENV_LIB and,
with --oracle, the CPU oracle.  Usage: python tools/replay_qp_instance.py [file] [--oracle]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import test_gpu_parity as tp
from sigmarl_amd import capi, cbf
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config
args = [a for a in sys.argv[1:] if not a.startswith("--")]
d = np.load(args[0] if args else os.path.join(ROOT, "gpurun_out", "fuzz_cbf_fail.npz"), allow_pickle=True)
kw = eval(str(d["kw"])); B, N = d["state"].shape[:2]
mp = load_map(kw["scenario_type"]); p = Parameters(**kw)
makers = [("hip", tp._hip_env)] + ([("oracle", ob.OracleEnv)] if "--oracle" in sys.argv else [])
res = {}
for name, make in makers:
    env = make(make_config(p, mp, B), mp)
    seg_l, seg_r = cbf.load_segment_tables(mp)
    env.cbf_attach(cbf.make_cbf_config(p), seg_l, seg_r)
    env.reset(np.repeat(np.arange(B), N).astype(np.int32), np.tile(np.arange(N), B).astype(np.int32), d["path"].reshape(-1, 4), d["state"].reshape(-1, 8), 1)
    if name == "oracle":
        env.get(capi.BUF_SHORT_TERM, copy=False)[:] = d["short"]
    else:
        import torch
        env.env.buffer(capi.BUF_SHORT_TERM)[:] = torch.as_tensor(d["short"]).to(env.env.device)
    safe, u, info = env.cbf_qp(d["act"].astype(np.float32))[:3]
    res[name] = (u, info)
    print(name, "iterations", info[:, 0].tolist(), "converged", info[:, 1].tolist())
    env.close()
print("saved: hip", d["info_hip"][:, 0].tolist(), d["info_hip"][:, 1].tolist(), "oracle", d["info_ora"][:, 0].tolist(), d["info_ora"][:, 1].tolist())
if "oracle" in res:
    print("max |u_hip - u_oracle| per env:", np.abs(res["hip"][0] - res["oracle"][0]).max(axis=(1, 2)).tolist())
