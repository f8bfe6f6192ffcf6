#!/usr/bin/env python
"""Where the time of ONE timed n-step launch goes on the host side (run on the GPU box): enqueue, wait, sync, vs the kernel's own duration by
HIP events.  Usage: python tools/launch_overhead.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
import bench

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sys.argv = sys.argv[:1]
ap = argparse.Namespace(agents=16, scenario="cpm_entire", distance="c2c", cbf=False, cbf_qp=False, cbf_group_size=0, streams=2, policy=False, policy_precision="fp32",
                        no_reset=False, separate_reset=False, no_gather=False, exchange="alltoall", chunk_steps=32, force_dist=False)
os.environ["SIGMAENV_TIMING_STRIDE"] = "1"
dev = torch.device("cuda", 0)
run = bench.GpuRun(ap, dev, 4096, 1, 0, T=T)
run.run_steps(0, 5)
run.finish_chunk()
torch.cuda.synchronize()
run.arm_timing()
for rep in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run_chunk(5 + rep * T, T)
    t1 = time.perf_counter()
    run.finish_chunk()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    ms, n = run.kernel_timing()
    print(f"rep {rep}: enqueue {1e6*(t1-t0):7.1f} us  finish_chunk {1e6*(t2-t1):6.1f} us  sync {1e6*(t3-t2):7.1f} us  total {1e6*(t3-t0):7.1f} us  kernel(events) {1e3*ms:7.1f} us  -> overhead {1e6*(t3-t0)-1e3*ms:6.1f} us")
    if rep == 2:
        time.sleep(0.5)  # idle gap: clocks drop
