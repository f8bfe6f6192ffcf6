#!/usr/bin/env python
"""Golden vectors for include/sigma_trig_f32.h (run in the build container; needs torch on the CPU).

Writes tests/golden/trig_f32.npz:
  x, sin, cos, tan, atan      : torch's own float32 sin / cos / tan / atan of x (a closed vector math library on this build) -- NOT
                                reproduced bit for bit; the tests hold the contract's correctly rounded values within one ulp of them
Deterministic: numpy Generator(PCG64(20260928)); one command: python tests/golden/gen/gen_trig_golden.py
"""
import os

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trig_f32.npz")


def main():
    rng = np.random.default_rng(20260928)
    n = 1 << 13
    a = np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-125, 125, n), rng.standard_normal(n) * 1e-2, rng.uniform(-0.7, 0.7, n)]).astype(np.float32)
    t = torch.from_numpy(a)
    np.savez_compressed(OUT, x=a, sin=torch.sin(t).numpy(), cos=torch.cos(t).numpy(),
                        tan=torch.tan(t).numpy(), atan=torch.atan(t).numpy())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
