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
    ap.add_argument("--streams", type=int, default=2, help="env shards per GPU, each stepped by its own handle on its own HIP stream "
                    "(envs are independent: same total work per step, the shards' latency-bound phases overlap the others' scan)")
    ap.add_argument("--policy", action="store_true", help="widening (SURVEY 8f-3): the actor MLP runs on the device before every step "
                    "(sigmaenv_actor_forward, MFMA bf16) instead of replaying precomputed actions; reported in config.policy")
    ap.add_argument("--cbf", action="store_true", help="widening (SURVEY 8f-4): rew_method='cbf' with the QP-free CBF margin reward "
                    "(sigmaenv_cbf_rewards before every step); reported in config.cbf")
    ap.add_argument("--cbf-qp", action="store_true", help="widening (BASELINE config 5): rew_method='cbf' with the centralized CBF-QP safety "
                    "filter solved for every env before every step (sigmaenv_cbf_qp); reported in config.cbf")
    ap.add_argument("--exchange", choices=["alltoall", "gather"], default="alltoall",
                    help="N > 1: how the rollout buffer is concatenated.  alltoall: distributed over the ranks by time slices (every rank "
                         "receives 1/N of the steps of ALL envs -- a data-parallel learner; the record crosses xGMI once, over all links); "
                         "gather: everything to rank 0 (the 7 links into one GPU bound the rate)")
    ap.add_argument("--chunk-steps", type=int, default=32, help="steps per rollout chunk gathered to the learner rank (N > 1)")
    ap.add_argument("--force-dist", action="store_true", help="diagnostic: init RCCL and run the gather even with one rank")
    args = ap.parse_args()

    # the contract is ONE JSON line on stdout: libraries (RCCL prints a version banner) get stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("SIGMAENV_TIMING_STRIDE", "32")  # HIP-event bracket around every 32nd step launch (each bracket costs a few microseconds)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: enough hardware queues for the shard + RCCL streams
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
    if args.cbf:
        params_kw.update(rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True)
    if args.cbf_qp:
        params_kw.update(rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True)
        args.cbf = True  # same call sites below; the QP launch replaces the margin launch
    # env shards of this GPU: S handles of B / S envs, each on its own HIP stream (no cross-env dependency anywhere in the path)
    # (only when every shard still fills the GPU's CUs with whole tiles; small batches are launch-bound and stay in one piece)
    S = args.streams if (args.streams >= 1 and B % max(1, args.streams) == 0 and (B // max(1, args.streams)) * N >= 64 * int(os.environ.get("BENCH_MIN_TILES", "512"))) else 1
    Bs = B // S
    main_stream = torch.cuda.current_stream(device)
    # alternate priorities: the runtime maps streams to a few hardware queues, and two shard streams that land on the same queue
    # (observed once RCCL has created its own streams) would run their kernels back to back instead of side by side
    streams = [main_stream] if S == 1 else [torch.cuda.Stream(device, priority=(0 if os.environ.get("BENCH_PRIO") == "none" else -(k % 2))) for k in range(S)]
    seed = 1000 + rank
    envs = []
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            kw = dict(params_kw, num_vmas_envs=Bs)
            e = SigmaEnv(Parameters(**kw), n_envs=Bs, device=device)
            e.reset_random(seed=seed * 64 + k)
            if args.cbf:
                e.cbf_attach()
            envs.append(e)
    env = envs[0]
    gen = torch.Generator(device=device).manual_seed(seed)
    n_act = 16
    acts = torch.empty((n_act, B, N, 2), dtype=torch.float32, device=device)
    acts[..., 0] = torch.rand((n_act, B, N), generator=gen, device=device)                 # v_cmd ~ U[0, 1]
    acts[..., 1] = torch.rand((n_act, B, N), generator=gen, device=device) * 0.5 - 0.25    # delta_cmd ~ U[-0.25, 0.25] rad
    torch.cuda.synchronize()
    gather = None
    gather_note = "disabled by --no-gather"
    if not args.no_gather:
        try:  # the rollout exchange must never take the benchmark down: fall back to "no gather" and say so in the JSON line
            ex_mode = args.exchange if args.chunk_steps % max(1, world) == 0 else "gather"
            gather = RolloutExchange(B, N, env.D, args.chunk_steps, device, force_collective=args.force_dist, mode=ex_mode)
            slot0 = gather.slot()
            for k, e in enumerate(envs):
                e.set_slab(slot0[k * Bs:(k + 1) * Bs])
                e.step(acts[0][k * Bs:(k + 1) * Bs])
            torch.cuda.synchronize()
            gather.advance()
            gather.flush()
            gather.wait_all()
            torch.cuda.synchronize()
            for k, e in enumerate(envs):
                e.auto_reset(seed=seed * 64 + k, counter=0, path_first=env.map.list_first[0], path_count=env.map.list_count[0])
            gather_note = (f"step kernel records (obs, reward, done) into a [{args.chunk_steps}, B, {N * (env.D + 1) + 1}] chunk buffer"
                           + ((f"; one async all-to-all per chunk (rank r receives steps [r T/N, (r+1) T/N) of every rank's chunk), double buffered"
                               if gather.mode == "alltoall" else "; one async gather per chunk to rank 0, double buffered")
                              if gather.collective else " (single GPU: no exchange)"))
        except Exception as exc:  # noqa: BLE001
            gather = None
            for e in envs:
                e.set_slab(None)
            gather_note = f"disabled: {type(exc).__name__}: {exc}"
            print(f"[bench] rollout exchange disabled: {exc}", file=sys.stderr)
    pf, pc = env.map.list_first[0], env.map.list_count[0]
    counter = [1]
    gather_state = {"note": None}

    actors, act_bufs = [], []
    if args.policy:
        from sigmarl_amd.actor import Actor, make_mlp
        torch.manual_seed(0)
        mlp = make_mlp(env.D)
        for k, e in enumerate(envs):
            with torch.cuda.stream(streams[k]):
                actors.append(Actor(mlp, low=[-1.0, -0.6109], high=[1.0, 0.6109]))  # -/+ (max_speed, max_steering)
                act_bufs.append(torch.zeros((Bs, N, 2), dtype=torch.float32, device=device))
    safe_bufs = [torch.zeros((Bs, N, 2), dtype=torch.float32, device=device) for _ in range(S)] if args.cbf_qp else []
    W = N * (env.D + 1) + 1
    act_ptrs = [[acts[q].data_ptr() + k * Bs * N * 2 * 4 for k in range(S)] for q in range(n_act)]
    shard_seeds = [seed * 64 + k for k in range(S)]
    fused = not (args.no_reset or args.separate_reset)

    import ctypes as C
    HArr, PArr, SArr = C.c_void_p * S, C.c_void_p * S, C.c_uint64 * S
    h_arr = HArr(*[e.h for e in envs])
    seed_arr = SArr(*shard_seeds)
    act_arrs = [PArr(*ap_) for ap_ in act_ptrs]
    slab_arr = PArr()
    many = envs[0].lib.step_autoreset_many

    def one_step(t):
        base = 0
        if gather is not None:
            slot = gather.slot(streams if S > 1 else None)  # orders the shard streams behind the gather that still reads this buffer
            base = slot.data_ptr()
        ap = act_ptrs[t % n_act]
        cnt = counter[0]
        if fused and not args.policy and not args.cbf:  # ONE binding call: every shard's record target + fused step / record / reset launch
            for k in range(S):
                slab_arr[k] = (base + k * Bs * W * 4) if base else None
            rc = many(h_arr, S, act_arrs[t % n_act], slab_arr if base else None, seed_arr, cnt, pf, pc)
            if rc != 0:
                raise RuntimeError(f"sigmaenv_step_autoreset_many failed with code {rc}")
        elif base:
            for k, e in enumerate(envs):
                e.set_slab_ptr(base + k * Bs * W * 4)
        if fused and args.policy:  # policy on device, then the fused step on the actions it wrote
            for k, e in enumerate(envs):
                actors[k].forward(e, act_bufs[k], seed=shard_seeds[k], counter=cnt)
                if args.cbf_qp:
                    e.cbf_qp(act_bufs[k], safe_bufs[k])
                elif args.cbf:
                    e.cbf_rewards(act_bufs[k])
                e.step_autoreset_ptr(act_bufs[k].data_ptr(), shard_seeds[k], cnt, pf, pc)
        elif fused and args.cbf:  # margin rewards of the action about to be applied, then the fused step that consumes them
            a = acts[t % n_act]
            for k, e in enumerate(envs):
                if args.cbf_qp:
                    e.cbf_qp(a[k * Bs:(k + 1) * Bs], safe_bufs[k])
                else:
                    e.cbf_rewards(a[k * Bs:(k + 1) * Bs])
                e.step_autoreset_ptr(ap[k], shard_seeds[k], cnt, pf, pc)
        elif fused:
            pass  # done above
        else:
            a = acts[t % n_act]
            for k, e in enumerate(envs):
                e.step(a[k * Bs:(k + 1) * Bs])
                if not args.no_reset:
                    e.auto_reset(seed=shard_seeds[k], counter=cnt, path_first=pf, path_count=pc)
        counter[0] += 1
        if gather is not None:
            if S > 1 and gather.t == gather.T - 1:  # the chunk is complete: the gather (main stream) follows every shard's last write
                for st in streams:
                    main_stream.wait_stream(st)
            try:
                gather.advance()
            except Exception as exc:  # noqa: BLE001 -- a failing exchange must not take the benchmark down: keep recording, stop gathering
                print(f"[bench] rollout gather failed, continuing without the exchange: {exc}", file=sys.stderr)
                gather.collective = False
                gather.pending = [None, None]
                gather_state["note"] = f"gather failed at run time ({type(exc).__name__}); record kept, exchange disabled"
                gather.t = 0

    def finish_chunk():
        if gather is not None:
            if S > 1:
                for st in streams:
                    main_stream.wait_stream(st)
            gather.flush()
            gather.wait_all()

    for t in range(args.warmup):
        one_step(t)
    finish_chunk()
    torch.cuda.synchronize()
    for e in envs:
        e.step_time_ms()  # arms the HIP-event bracketing of the step launches (on the env's stream)
    resets_before = sum(int(e.buffer(capi.BUF_TIMER)[:, 3].sum().item()) for e in envs)  # episodes_reset counters

    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.steps):
        one_step(args.warmup + t)
    finish_chunk()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    timings = [e.step_time_ms() for e in envs]
    n_launch = sum(n for _, n in timings)
    kernel_ms = sum(ms * n for ms, n in timings) / max(1, n_launch)
    dones = sum(int(e.buffer(capi.BUF_TIMER)[:, 3].sum().item()) for e in envs) - resets_before
    total_agent_steps = N * B * world * args.steps
    value = total_agent_steps / elapsed
    bytes_per = algorithmic_bytes_per_agent_step(N)
    # the fused launch also rewrites the whole record of every agent of a reset env: 320 + 5 N bytes
    reset_bytes = (320 + 5 * N) * N * (dones / max(1, args.steps)) if fused else 0.0
    slab_bytes = 4.0 * (N * (env.D + 1) + 1) * B if gather is not None else 0.0  # the rollout record row of every env
    step_bytes = bytes_per * N * B + reset_bytes + slab_bytes  # algorithmic bytes of one step over all shards of this GPU
    if S == 1:
        achieved = step_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None
        achieved_basis = "algorithmic bytes per launch / average launch duration (HIP events)"
    else:  # S launches per step run concurrently: the rate the GPU sustains is bytes per step over the wall time per step
        achieved = step_bytes / (elapsed / args.steps) / 1e9
        achieved_basis = (f"{S} concurrent launches per step (one per env shard): algorithmic bytes of all shards per step / wall time per step; "
                          "kernel_avg_ms is the average duration of ONE shard's launch while it shares the GPU with the others")
    traffic = None
    try:  # HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_passes.sh), corrected as the MI355X guide says
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            tr = json.load(f)
        if tr.get("n_agents") == N and tr.get("envs_per_launch", tr.get("envs_per_gpu")) == Bs and tr.get("distance") == args.distance:
            traffic = tr["hbm_bytes_per_launch"] * S  # per step, like `achieved` (S launches of B / S envs)
    except Exception:  # noqa: BLE001
        pass
    out = {
        "metric": f"env-steps/sec (agents x envs x steps), CPM scenario, {N} agents",
        "value": value, "unit": "agent-env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"cpm_entire map, {N} agents x {B} envs per GPU ({B * world} envs total), {args.distance} distance, rew_method={params_kw['rew_method']}, "
                        f"dt=0.05, obs_dim={env.D}, fused step + device-side reset of finished envs "
                        + ("(one launch)" if fused else "(two launches)" if not args.no_reset else "(resets disabled)") + ((" + rollout record" + ((" + " + gather.mode) if gather.collective else "")) if gather else ""),
            "n_agents": N, "envs_per_gpu": B, "envs_total": B * world, "distance": args.distance, "env_shards_per_gpu": S,
            "policy": ("actor MLP 32-256-256-256-4 (bf16 MFMA, TanhNormal sample) on device before every step" if args.policy
                       else "none in the timed region (precomputed actions resident in HBM)"),
            **({"cbf": ("centralized CBF-QP safety filter of every env (sigmaenv_cbf_qp: 32 controls, 96 lane + 1080 pair constraints, projected "
                        "Newton in float64) solved before every step; the step penalises the deviation from the safe action" if args.cbf_qp else
                        "QP-free CBF margin reward (sigmaenv_cbf_rewards: 3 circles per vehicle, 9-point fp16 pseudo-distance stencils to both "
                        "boundaries, float64 margins) launched before every step")} if args.cbf else {}),
            "resets_per_step_per_gpu": dones / max(1, args.steps), "rollout_gather": gather_state["note"] or gather_note,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": (achieved / 8000.0) if achieved else None,
            "traffic": traffic, "kernel": "sigmaenv_step_kernel", "kernel_avg_ms": kernel_ms, "kernel_launches": n_launch,
            "algorithmic_bytes_per_agent_env_step": bytes_per, "algorithmic_bytes_per_launch": step_bytes / S, "launches_per_step": S, "achieved_basis": achieved_basis,
        },
    }
    if rank == 0 and args.cpu_seconds > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(params_kw, B, args.cpu_seconds)
    for e in envs:
        e.close()
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        os.write(1, (json.dumps(out) + "\n").encode())
    os.dup2(2, 1)  # RCCL prints its version banner at exit: keep it off stdout as well


if __name__ == "__main__":
    main()
