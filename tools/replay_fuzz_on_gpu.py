#!/usr/bin/env python
"""Replays the fresh reference rollouts that tools/fuzz_reference.py --keep tests/golden/_fuzz left behind (untracked scratch data, generated
in the build container) through the HIP path on the GPU box: the same check as the committed goldens, on trajectories nobody tuned anything on."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import test_gpu_parity as tp
import traj_replay as tr

d = os.path.join(ROOT, "tests", "golden", "_fuzz")
tr.GOLDEN_DIR = d
names = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(d, "traj_*.npz")))
worst, bad = {}, 0
for name in names:
    z, meta = tr.load_fixture(name)
    cfg, mp = tr.config_from_meta(meta)
    env = tp._hip_env(cfg, mp)
    rep = tr.replay(env, z, meta, mp, check_next=False)  # (post-step snapshots; the post-reset ones can carry the env-0 quirk, which only the oracle replays emulate)
    env.close()
    ok = rep.total_mismatch() == 0 and all(v <= 1e-5 for v in rep.max_abs.values())
    bad += int(not ok)
    for k, v in rep.max_abs.items():
        worst[k] = max(worst.get(k, 0.0), float(v))
    if not ok:
        print("FAIL", name, str(rep))
print(f"HIP replay of {len(names)} fresh reference rollouts: {len(names) - bad} exact on masks / indices / counters and within 1e-5; worst fp32 error per buffer:",
      {k: f"{v:.2e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]})
