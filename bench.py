#!/usr/bin/env python
"""bench.py -- env-steps/sec (agents x envs x steps) of the fused CAV environment step, CPM scenario, 16 agents.

Contract (see the round prompt): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it through
``python -m torch.distributed.run`` with one rank per GPU.  W untimed warm-up steps, then exactly K timed steps bracketed by a
barrier + torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input: ONE launch (sigmaenv_step_autoreset) that steps agents x envs,
writes the rollout record of the step (observation incl. the terminal one, reward, done -- the reference's step_and_maybe_reset
keeps both the terminal and the post-reset observation) into the rollout chunk buffer and re-places the envs that finished; for
N > 1 additionally one asynchronous RCCL gather per chunk of steps to the learner rank.  Inputs (actions) are resident in HBM
before the timed region starts.
Weak scaling: every GPU steps BASELINE config 2 (16 agents x 4096 envs); N = 8 is config 3 (32768 envs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_READ = 44  # state 32 + action 8 + path id 4           


def algorithmic_bytes_per_agent_step(n_agents: int) -> int:
    """44 B read + (251 + 5 N) B written per agent-env-step at obs_dim = 32 (375 B at N = 16)."""
    return ALGO_BYTES_READ + 251 + 5 * n_agents


def cpu_baseline(params_kw, n_envs, target_seconds):
    """The CPU oracle (bit-checked C restatement of the reference path, OpenMP over envs) timed on this box's host cores on a
    bounded sample of the same workload.  Checker code used as the measured baseline leg only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    from sigmarl_amd import capi
    from sigmarl_amd.maps import load_map
    from sigmarl_amd.params import Parameters, make_config

    p = Parameters(**params_kw)
    mp = load_map(p.scenario_type)
    cfg = make_config(p, mp, n_envs)
    env = ob.OracleEnv(cfg, mp)
    env.get(capi.BUF_DONE, copy=False)[:] = 1
    pf, pc = mp.list_first[0], mp.list_count[0]
    env.auto_reset(0, 0, pf, pc)
    rng = np.random.default_rng(0)
    N = cfg.n_agents
    acts = [np.stack([rng.uniform(0, 1, (n_envs, N)), rng.uniform(-0.25, 0.25, (n_envs, N))], axis=-1).astype(np.float32) for _ in range(8)]
    env.step(acts[0])  # warm-up (also spins the OpenMP team up)
    env.auto_reset(0, 1, pf, pc)
    t0 = time.perf_counter()
    k = 0
    while True:
        env.step(acts[k % 8])
        env.auto_reset(0, k + 2, pf, pc)
        k += 1
        el = time.perf_counter() - t0
        if el >= target_seconds or k >= 4096:
            break
    env.close()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    threads = int(os.environ.get("OMP_NUM_THREADS", cores))
    return {
        "value": N * n_envs * k / el, "unit": "agent-env-steps/s", "cores": threads, "kind": "port",
        "sample": f"C oracle (oracle/sigmaenv_oracle.c, OpenMP over envs), {N} agents x {n_envs} envs x {k} steps incl. resets, {el:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=16)
    ap.add_argument("--distance", choices=["c2c", "mtv"], default="c2c")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables it)")
    ap.add_argument("--no-reset", action="store_true", help="diagnostic: leave finished envs un-reset")
    ap.add_argument("--separate-reset", action="store_true", help="two launches per step (sigmaenv_step; sigmaenv_auto_reset) instead of the fused one")
    ap.add_argument("--no-gather", action="store_true", help="diagnostic: no rollout record (and no gather to the learner rank for N > 1)")
    ap.add_argument("--chunk-steps", type=int, default=32, help="steps per rollout chunk gathered to the learner rank (N > 1)")
    ap.add_argument("--force-dist", action="store_true", help="diagnostic: init RCCL and run the gather even with one rank")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: libraries (RCCL prints a version banner) get stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from sigmarl_amd import capi
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters
    from sigmarl_amd.shard import RolloutExchange

    B, N = args.envs_per_gpu, args.agents
    params_kw = dict(n_agents=N, scenario_type="cpm_entire", dt=0.05, is_use_mtv_distance=(args.distance == "mtv"), rew_method="distance",
                     is_apply_mask=False, is_obs_noise=False, max_steps=128, num_vmas_envs=B)
    env = SigmaEnv(Parameters(**params_kw), n_envs=B, device=device)
    seed = 1000 + rank
    env.reset_random(seed=seed)
    gen = torch.Generator(device=device).manual_seed(seed)
    n_act = 16
    acts = torch.empty((n_act, B, N, 2), dtype=torch.float32, device=device)
    acts[..., 0] = torch.rand((n_act, B, N), generator=gen, device=device)                 # v_cmd ~ U[0, 1]
    acts[..., 1] = torch.rand((n_act, B, N), generator=gen, device=device) * 0.5 - 0.25    # delta_cmd ~ U[-0.25, 0.25] rad
    gather = None
    gather_note = "disabled by --no-gather"
    if not args.no_gather:
        try:  # the rollout exchange must never take the benchmark down: fall back to "no gather" and say so in the JSON line
            gather = RolloutExchange(B, N, env.D, args.chunk_steps, device, force_collective=args.force_dist)
            env.set_slab(gather.slot())
            env.step(acts[0])
            gather.advance()
            gather.flush()
            gather.wait_all()
            torch.cuda.synchronize()
            env.auto_reset(seed=seed, counter=0, path_first=env.map.list_first[0], path_count=env.map.list_count[0])
            gather_note = (f"step kernel records (obs, reward, done) into a [{args.chunk_steps}, B, {N * (env.D + 1) + 1}] chunk buffer"
                           + ("; one async gather per chunk to rank 0, double buffered" if gather.collective else " (single GPU: no exchange)"))
        except Exception as exc:  # noqa: BLE001
            gather = None
            env.set_slab(None)
            gather_note = f"disabled: {type(exc).__name__}: {exc}"
            print(f"[bench] rollout exchange disabled: {exc}", file=sys.stderr)
    pf, pc = env.map.list_first[0], env.map.list_count[0]
    counter = [1]

    def one_step(t):
        if gather is not None:
            env.set_slab(gather.slot())
        if args.no_reset or args.separate_reset:
            env.step(acts[t % n_act])
            if not args.no_reset:
                env.auto_reset(seed=seed, counter=counter[0], path_first=pf, path_count=pc)
        else:  # one launch: the step, its record, then the device-side reset of the finished envs of the tile
            env.step_autoreset(acts[t % n_act], seed=seed, counter=counter[0], path_first=pf, path_count=pc)
        counter[0] += 1
        if gather is not None:
            gather.advance()

    for t in range(args.warmup):
        one_step(t)
    if gather is not None:
        gather.flush()
        gather.wait_all()
    env.step_time_ms()  # arms the HIP-event bracketing of the step launches (on the env's stream)
    resets_before = int(env.buffer(capi.BUF_TIMER)[:, 3].sum().item())  # episodes_reset counters

    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
    if gather is not None:
        gather.flush()
        gather.wait_all()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    kernel_ms, n_launch = env.step_time_ms()
    dones = int(env.buffer(capi.BUF_TIMER)[:, 3].sum().item()) - resets_before
    total_agent_steps = N * B * world * args.steps
    value = total_agent_steps / elapsed
    bytes_per = algorithmic_bytes_per_agent_step(N)
    fused = not (args.no_reset or args.separate_reset)
    # the fused launch also rewrites the whole record of every agent of a reset env: 320 + 5 N bytes
    reset_bytes = (320 + 5 * N) * N * (dones / max(1, args.steps)) if fused else 0.0
    slab_bytes = 4.0 * (N * (env.D + 1) + 1) * B if gather is not None else 0.0  # the rollout record row of every env
    achieved = (bytes_per * N * B + reset_bytes + slab_bytes) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
    traffic = None
    try:  # HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_passes.sh), corrected as the MI355X guide says
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            tr = json.load(f)
        if tr.get("n_agents") == N and tr.get("envs_per_gpu") == B and tr.get("distance") == args.distance:
            traffic = tr["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        pass
    out = {
        "metric": "env-steps/sec (agents x envs x steps), CPM scenario, 16 agents",
        "value": value, "unit": "agent-env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"cpm_entire map, {N} agents x {B} envs per GPU ({B * world} envs total), {args.distance} distance, rew_method=distance, "
                        f"dt=0.05, obs_dim={env.D}, fused step + device-side reset of finished envs "
                        + ("(one launch)" if fused else "(two launches)" if not args.no_reset else "(resets disabled)") + ((" + rollout record" + (" + gather" if gather.collective else "")) if gather else ""),
            "n_agents": N, "envs_per_gpu": B, "envs_total": B * world, "distance": args.distance,
            "resets_per_step_per_gpu": dones / max(1, args.steps), "rollout_gather": gather_note,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None,
            "traffic": traffic, "kernel": "sigmaenv_step_kernel", "kernel_avg_ms": kernel_ms, "kernel_launches": n_launch,
            "algorithmic_bytes_per_agent_env_step": bytes_per, "algorithmic_bytes_per_launch": bytes_per * N * B + reset_bytes + slab_bytes,
        },
    }
    if rank == 0 and args.cpu_seconds > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(params_kw, B, args.cpu_seconds)
    env.close()
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        os.write(1, (json.dumps(out) + "\n").encode())
    os.dup2(2, 1)  # RCCL prints its version banner at exit: keep it off stdout as well


if __name__ == "__main__":
    main()
