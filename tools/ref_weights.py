#!/usr/bin/env python
"""Prints include/sigmaenv_ref_weights.h's table: the bit patterns of torch.linspace(1, 0.2, n) / sum in float32 (road_traffic.py:536-543), n = 1 .. 8."""
import torch

for ns in range(1, 9):
    w = torch.linspace(1, 0.2, steps=ns, dtype=torch.float32)
    w /= w.sum()
    print(ns, "{" + ", ".join("0x%08Xu" % (int(x) & 0xFFFFFFFF) for x in w.view(torch.int32).tolist()) + "}")
