#!/usr/bin/env python
"""Randomised check of the exact-fp32 MLP (sigmaenv_mlp32_*) against torch.nn in fp32 on the CPU: random depths (2-4 Linear layers), input
widths 1..600, output widths 1..32, row counts incl. ragged tiles.  Diagnostic tool for the GPU box:  python tools/fuzz_mlp32.py [--cases 60]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from sigmarl_amd.actor import Mlp32
from sigmarl_amd.env import SigmaEnv
from sigmarl_amd.params import Parameters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    torch.manual_seed(args.seed)
    env = SigmaEnv(Parameters(n_agents=2, scenario_type="cpm_entire", is_apply_mask=False, is_obs_noise=False), n_envs=2, device="cuda:0")
    worst = 0.0
    for k in range(args.cases):
        depth = int(rng.integers(2, 5))
        in_dim = int(rng.choice([1, 3, 10, 16, 17, 31, 32, 33, 43, 64, 100, 255, 256, 257, 512, 600]))
        out_dim = int(rng.integers(1, 33))
        rows = int(rng.choice([1, 5, 31, 32, 33, 64, 1000, 4097]))
        dims = [in_dim] + [256] * (depth - 1) + [out_dim]
        layers = []
        for a, b in zip(dims[:-1], dims[1:]):
            layers += [torch.nn.Linear(a, b), torch.nn.Tanh()]
        mlp = torch.nn.Sequential(*layers[:-1])
        scale = float(rng.choice([1.0, 1.7]))
        with torch.no_grad():
            for m in mlp:
                if isinstance(m, torch.nn.Linear):
                    m.weight.mul_(scale)
        net = Mlp32(mlp)
        x = (torch.rand((rows, in_dim)) * 2 - 1) * 1.5
        y = net.forward(env, x.cuda().contiguous())
        env.sync()
        with torch.no_grad():
            want = mlp(x)
        err = float((y.cpu() - want).abs().max() / max(1.0, float(want.abs().max())))
        assert err <= 2e-5, (k, dims, rows, err)
        worst = max(worst, err)
        net.close()
    print(f"fuzz_mlp32: {args.cases} random networks (2-4 layers, input widths 1..600, outputs 1..32, rows 1..4097): max relative error vs torch.nn fp32 {worst:.2e}")
    env.close()


if __name__ == "__main__":
    main()
