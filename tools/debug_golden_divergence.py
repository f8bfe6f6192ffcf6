#!/usr/bin/env python
"""Diagnostic: replays one golden trajectory through the HIP env and the oracle side by side and prints the first (step, env, agent) where a float buffer of
the two differs by more than 1e-5.  Usage: tools/debug_golden_divergence.py <trajectory name>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob, traj_replay as tr
from sigmarl_amd import capi
from sigmarl_amd.env import NumpyAdapter, SigmaEnv

name = sys.argv[1]
z, meta = tr.load_fixture(name)
cfg, mp = tr.config_from_meta(meta)
dev, ora = NumpyAdapter(SigmaEnv(cfg=cfg, map_table=mp, device="cuda:0")), ob.OracleEnv(cfg, mp)
for e in (dev, ora):
    tr.apply_initial_reset(e, z, mp, meta)
bufs = {"state": capi.BUF_STATE, "dist_left": capi.BUF_DIST_LEFT, "dist_right": capi.BUF_DIST_RIGHT, "dist_ref": capi.BUF_DIST_REF, "closest": capi.BUF_CLOSEST, "path": capi.BUF_PATH}
def check(t, tag, envs=None):
    for nm, w in bufs.items():
        a, b = dev.get(w).astype(np.float64), ora.get(w).astype(np.float64)
        if envs is not None:  # (after events only the touched envs are comparable: the oracle's replay reproduces a reference quirk in the others)
            keep = np.zeros(a.shape[0], bool); keep[envs] = True
            a, b = np.where(keep.reshape((-1,) + (1,) * (a.ndim - 1)), a, 0), np.where(keep.reshape((-1,) + (1,) * (b.ndim - 1)), b, 0)
        d = np.abs(a - b)
        if d.max() > 1e-5:
            idx = np.unravel_index(np.argmax(d), d.shape)
            print("step", t, tag, nm, "max err", d.max(), "at", idx, "hip", a[idx], "oracle", b[idx])
            bi = idx[:2]
            print(" state", ora.get(capi.BUF_STATE)[bi], "\n path", ora.get(capi.BUF_PATH)[bi], "closest hip", dev.get(capi.BUF_CLOSEST)[bi], "oracle", ora.get(capi.BUF_CLOSEST)[bi])
            print(" dist_left hip", dev.get(capi.BUF_DIST_LEFT)[bi], "oracle", ora.get(capi.BUF_DIST_LEFT)[bi])
            print(" dist_right hip", dev.get(capi.BUF_DIST_RIGHT)[bi], "oracle", ora.get(capi.BUF_DIST_RIGHT)[bi])
            print(" vertices", ora.get(capi.BUF_VERTICES)[bi].ravel())
            np.savez(os.path.join(ROOT, "gpurun_out", "divergence.npz"), step=t, state=ora.get(capi.BUF_STATE), path=ora.get(capi.BUF_PATH), act=z["act"][t])
            ev = [k for k in range(len(z["ev_step"])) if z["ev_step"][k] == t]
            print(" events of this step (kind 0 = agent, env, agent, path_id, point_id):", [(int(z["ev_kind"][k]), int(z["ev_env"][k]), int(z["ev_agent"][k])) for k in ev])
            sys.exit(1)


for t in range(int(meta["T"])):
    for e in (dev, ora):
        e.step(z["act"][t])
    check(t, "after step")
    for e in (dev, ora):
        touched = tr.apply_events(e, z, mp, t)
        if touched:
            e.observe()
    if touched:
        check(t, "after events", touched)
print("no divergence")
