#!/usr/bin/env python
"""Golden vectors for include/sigma_trig_f32.h (run in the build container; needs torch on the CPU).

Writes tests/golden/trig_f32.npz:
  atan2_y, atan2_x, atan2_out : torch.atan2 on PyTorch-CPU (SLEEF's Sleef_atan2f*_u10) -- the header restates it bit for bit
  x, sin, cos, tan, atan      : torch's own float32 sin / cos / tan / atan of x (MKL vector math on this build) -- NOT reproduced bit
                                for bit; the tests hold the contract's correctly rounded values within one ulp of them
Deterministic: numpy Generator(PCG64(20260928)); one command: python tests/golden/gen/gen_trig_golden.py
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trig_f32.npz")


def main():
    rng = np.random.default_rng(20260928)
    n = 1 << 13
    y = np.concatenate([rng.standard_normal(n), rng.uniform(-5, 5, n), rng.standard_normal(n) * 1e-4]).astype(np.float32)
    x = np.concatenate([rng.standard_normal(n), rng.uniform(-5, 5, n), rng.standard_normal(n)]).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-40, -1e-40, 3.0], np.float32)
    ey, ex = np.meshgrid(edge, edge, indexing="ij")
    y = np.concatenate([y, ey.ravel()])
    x = np.concatenate([x, ex.ravel()])
    out = torch.atan2(torch.from_numpy(y), torch.from_numpy(x)).numpy()
    a = np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-125, 125, n), rng.standard_normal(n) * 1e-2, rng.uniform(-0.7, 0.7, n)]).astype(np.float32)
    t = torch.from_numpy(a)
    np.savez_compressed(OUT, atan2_y=y, atan2_x=x, atan2_out=out, x=a, sin=torch.sin(t).numpy(), cos=torch.cos(t).numpy(),
                        tan=torch.tan(t).numpy(), atan=torch.atan(t).numpy())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
