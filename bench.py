#!/usr/bin/env python
"""bench.py -- env-steps/sec (agents x envs x steps) of the fused CAV environment step.

Contract (see the round prompt): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it through
``python -m torch.distributed.run`` with one rank per GPU.  W untimed warm-up steps, then exactly K timed steps bracketed by a
barrier + torch.cuda.synchronize() on both sides, MAX over ranks, rank 0 prints ONE JSON line.

`value` is the contract measurement to the letter: W warm-up steps, then K timed steps, on the device as the setup left it.  An MI355X that idled for
~10 ms is back at its idle clocks and needs tens of milliseconds of load to reach its sustained ones -- a timed region of 20 steps (0.6 ms) sees 29 us per
step, a rollout loop that has been running for 50 ms sees 26.3 -- so bench.py ALSO reports what a running rollout loop sees: after `value` is taken it runs
--condition-ms (default 200) of the same step launches and then W + K steps again: `value_sustained` / `ms_per_step_sustained` at the top level
(`warmup_effective_steps_sustained` says how many steps preceded that timed region), details in config.sustained.  --condition-ms 0 skips it.  Rounds
1-3 reported the contract figure as `value`, round 4 the sustained one (with the contract figure in config.cold_start); from round 5 on `value` is the
contract figure again and stays comparable with rounds 1-3 and with the driver's clock.

A "step" is one pass of the hot path over one batch of synthetic input: agents x envs stepped once by the fused kernel, which also writes
the rollout record of the step (observation incl. the terminal one, reward, done -- the reference's step_and_maybe_reset keeps both the
terminal and the post-reset observation) into the rollout chunk buffer and re-places the envs that finished.  The steps are issued as the
reference's rollout loop issues them (helper_training.py:687-788) -- a chunk of T steps per call: ONE launch of sigmaenv_step_autoreset_n in
which every wavefront walks its env through the T steps (--chunk T; default: the largest divisor of --steps up to 32; --chunk 1 = one launch
per step, the round-2 form, timed as well and reported in config.per_step_launch).  For N > 1 additionally one asynchronous RCCL exchange of
the rollout chunk per launch (--exchange alltoall: by time slices to every rank, the default; --exchange gather: everything to rank 0).
Inputs (actions) are resident in HBM before the timed region starts.  Weak scaling: every GPU steps BASELINE config 2 (CPM map, 16 agents x
4096 envs); N = 8 is config 3 (32768 envs; rank r's envs are envs [4096 r, 4096 (r + 1)) of the batch -- `env_index_base` -- and draw what the
unsharded batch would draw).  --emulate-ranks R: config 3's workload on ONE GPU, rank by rank (config.emulated_ranks).

Other workloads of BASELINE.json through the same code path:
  --scenario on_ramp_1 --agents 32 --envs-per-gpu 8192      config 4 (injected start, see sigmarl_amd.maps.injected_start)
  --cbf-qp                                                  config 5 (centralized CBF-QP before every step; --cbf-group-size M: the grouped QPs)
  --sweep                                                   the metric's batch sweep 16 agents x {256 .. 32768} envs in `sweep`

roofline (SURVEY.md section 8d / BASELINE.md section 3): `achieved` = algorithmic bytes per agent-env-step (44 + 251 + 5 N) x agent-env-steps
per second of ONE GPU; `frac` = achieved / 8 TB/s.  The record and reset rewrites are real extra stores and are reported separately
(`achieved_incl_record`).  The step is VALU-bound, not HBM-bound: `valu_*` carry the issue-side picture from the committed PMC passes.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_READ = 44  # state 32 + action 8 + path id 4           
HBM_PEAK_GBPS = 8000.0
FP32_PEAK_TFLOPS = 157.3


def algorithmic_bytes_per_agent_step(n_agents: int, obs_dim: int = 32) -> int:
    """44 B read + (251 + 5 N) B written per agent-env-step at obs_dim = 32 (375 B at N = 16); another observation layout changes
    the observation's 4 x obs_dim bytes of that sum."""
    return ALGO_BYTES_READ + 251 + 4 * (obs_dim - 32) + 5 * n_agents


def make_params_kw(args, n_envs):
    kw = dict(n_agents=args.agents, scenario_type=args.scenario, dt=0.05, is_use_mtv_distance=(args.distance == "mtv"), rew_method="distance",
              is_apply_mask=False, is_obs_noise=False, max_steps=128, num_vmas_envs=n_envs)
    if args.cbf_qp:
        kw.update(rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True)
        if args.cbf_group_size > 0:  # the grouped CBF-QPs (Parameters.is_grouping_agents, cbf_qp.py:1562-2281) instead of the centralized one
            kw.update(is_grouping_agents=True, max_group_size=args.cbf_group_size, adaptive_lambda=True)
    elif args.cbf:
        kw.update(rew_method="cbf", is_solve_qp=False, is_using_cbf_training=True)
    if getattr(args, "defaults", False):  # the reference's own Parameters defaults for this path (helper_common.py:66-79): mtv distance, mask, sensor noise
        kw.update(is_use_mtv_distance=True, is_apply_mask=True, is_obs_noise=True, obs_noise_level=0.05)
    for item in getattr(args, "param", None) or []:  # --param key=value: any other Parameters field (observation switches ...)
        k, _, v = item.partition("=")
        kw[k.strip()] = json.loads(v) if v.strip().lower() not in ("true", "false") else (v.strip().lower() == "true")
    return kw


def params_note(args):
    kw = make_params_kw(args, 1)
    extra = {k: v for k, v in kw.items() if k in ("is_apply_mask", "is_obs_noise") and v} | {k: kw[k] for k in ((it.partition("=")[0].strip()) for it in (args.param or []))}
    return (", " + ", ".join(f"{k}={v}" for k, v in sorted(extra.items()))) if extra else ""


def needs_injected_start(mp, n_agents):
    """More agents than the reference's own reset can place on this map: start from the injected state."""
    return n_agents > 2 * mp.default_n_agents and mp.scenario_type != "cpm_entire"


def usable_cores():
    """Host cores this process may actually run on: the affinity mask clipped by the cgroup CPU quota (the GPU box shows 256 hardware
    threads under a 16-CPU quota; an OpenMP team sized by the former is throttled to a ninth of the 16-thread rate)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = parts[0], float(parts[1])
            else:
                quota, period = parts[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    return cores


def cpu_baseline(args, n_envs, target_seconds):
    """The CPU oracle (bit-checked C restatement of the reference path, OpenMP over envs) timed on this box's host cores on a
    bounded sample of the same workload.  Checker code used as the measured baseline leg only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    from sigmarl_amd import capi
    from sigmarl_amd.maps import injected_start, load_map
    from sigmarl_amd.params import Parameters, make_config

    threads = int(os.environ.get("OMP_NUM_THREADS", usable_cores()))
    try:  # size the oracle's OpenMP team before its first parallel region
        import ctypes

        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        pass
    p = Parameters(**make_params_kw(args, n_envs))
    mp = load_map(p.scenario_type)
    cfg = make_config(p, mp, n_envs)
    env = ob.OracleEnv(cfg, mp)
    pf, pc = mp.list_first[0], mp.list_count[0]
    N = cfg.n_agents
    if needs_injected_start(mp, N):
        idx, st = injected_start(mp, N)
        ids = np.zeros((n_envs, N, 4), np.int32)
        ids[..., 0] = np.asarray([mp.global_path(0, q) for q in idx], np.int32)[None, :]
        ids[..., 2] = np.asarray(idx, np.int32)[None, :]
        st8 = np.zeros((n_envs, N, 8), np.float32)
        st8[..., 0:3] = np.asarray(st, np.float32)[None, :, :]
        env.reset(np.repeat(np.arange(n_envs, dtype=np.int32), N), np.tile(np.arange(N, dtype=np.int32), n_envs), ids.reshape(-1, 4), st8.reshape(-1, 8), 1)
    else:
        env.get(capi.BUF_DONE, copy=False)[:] = 1
        env.auto_reset(0, 0, pf, pc)
    rng = np.random.default_rng(0)
    acts = [np.stack([rng.uniform(0, 1, (n_envs, N)), rng.uniform(-0.25, 0.25, (n_envs, N))], axis=-1).astype(np.float32) for _ in range(8)]
    pre = lambda a: None  # noqa: E731
    what = ""
    if args.cbf_qp or args.cbf:  # the same launch order as the GPU leg: CBF module on the action about to be applied, then the step
        from sigmarl_amd import cbf as cbfmod

        seg_l, seg_r = cbfmod.load_segment_tables(mp)
        env.cbf_attach(cbfmod.make_cbf_config(p), seg_l, seg_r)
        pre = (lambda a: env.cbf_qp(a)) if args.cbf_qp else (lambda a: env.cbf_rewards(a, want_margins=False))
        what = " + centralized CBF-QP before every step" if args.cbf_qp else " + CBF margin rewards before every step"
    pre(acts[0])
    env.step(acts[0])  # warm-up (also spins the OpenMP team up)
    env.auto_reset(0, 1, pf, pc)
    t0 = time.perf_counter()
    k = 0
    while True:
        pre(acts[k % 8])
        env.step(acts[k % 8])
        env.auto_reset(0, k + 2, pf, pc)
        k += 1
        el = time.perf_counter() - t0
        if el >= target_seconds or k >= 4096:
            break
    env.close()
    return {
        "value": N * n_envs * k / el, "unit": "agent-env-steps/s", "cores": threads, "kind": "port",
        "sample": f"C oracle (oracle/sigmaenv_oracle.c, OpenMP over envs), {p.scenario_type}, {N} agents x {n_envs} envs x {k} steps incl. resets{what}, {el:.1f} s",
    }


_SHARD_STREAMS = {}


def shard_streams(torch, device, S):
    """The S shard streams of this process, created once.  The runtime maps streams to a few hardware queues as they are created; two shard
    streams that land on the same queue run their kernels back to back instead of side by side (observed with streams created later in
    the process, and once RCCL has created its own) -- alternate priorities keep the pair apart."""
    key = (str(device), S)
    if key not in _SHARD_STREAMS:
        _SHARD_STREAMS[key] = [torch.cuda.Stream(device, priority=-(k % 2)) for k in range(S)]
    return _SHARD_STREAMS[key]


def pick_chunk(steps: int, cap: int = 32) -> int:
    """Steps per launch: the largest divisor of `steps` that is at most `cap` (the timed region is then whole launches) -- unless `steps` has no divisor of at
    least cap / 2 (a prime step count ...): then `cap`, and the last launch of the region is a shorter one."""
    if steps <= 0:
        return 1
    best = max(d for d in range(1, min(cap, steps) + 1) if steps % d == 0)
    return best if 2 * best >= min(cap, steps) else min(cap, steps)


class GpuRun:
    """The envs of ONE GPU: S shards (handles) of B / S envs on S HIP streams, precomputed actions, the rollout record + exchange.
    T > 1: one handle, T steps per launch (sigmaenv_step_autoreset_n); T == 1: one launch per step and shard."""

    def __init__(self, args, device, B, world, rank, with_exchange=True, T=1, env_base=None):
        import torch
        from sigmarl_amd.env import SigmaEnv
        from sigmarl_amd.maps import injected_start
        from sigmarl_amd.params import Parameters
        from sigmarl_amd.shard import RolloutExchange

        self.torch, self.args, self.device, self.B, self.N = torch, args, device, B, args.agents
        N = self.N
        self.T = T = max(1, int(T))
        # env shards of this GPU (no cross-env dependency anywhere in the path); small batches stay in one piece
        # Shards pay (per-step launches only) while the whole batch is about one resident round of wavefronts (one env per wavefront, 16 per CU): the
        # tail of one shard's launch -- its reset-heavy wavefronts -- then overlaps the other shard's start.  Larger batches run several rounds per launch
        # and fill the tail by themselves (measured: 2 shards 13 % faster at 4096 envs, 20 % at 8192, slower from 16384 on).  A T-step launch has no
        # per-step tail to hide: one handle.
        min_tiles, max_envs = int(os.environ.get("BENCH_MIN_TILES", "1024")), int(os.environ.get("BENCH_SHARD_MAX_ENVS", "8192"))
        S = args.streams if (T == 1 and args.streams >= 1 and B % max(1, args.streams) == 0 and (B // max(1, args.streams)) >= min_tiles and B <= max_envs) else 1
        self.S, self.Bs = S, B // S
        Bs = self.Bs
        self.env_base = rank * B if env_base is None else int(env_base)  # this GPU's first env in the whole batch (shard.shard_range)
        self.main_stream = torch.cuda.current_stream(device)
        self.streams = [self.main_stream] if S == 1 else shard_streams(torch, device, S)
        self.seed = 1000  # one seed for the whole batch: the generator is keyed on (seed, counter, env index in the batch, agent, draw)
        self.envs = []
        kw = make_params_kw(args, Bs)
        for k in range(S):
            with torch.cuda.stream(self.streams[k]):
                e = SigmaEnv(Parameters(**kw), n_envs=Bs, device=device, env_index_base=self.env_base + k * Bs)
                if needs_injected_start(e.map, N):
                    e.reset_injected(*injected_start(e.map, N))
                    self.start = "injected (sigmarl_amd.maps.injected_start: the same state in every env, zero speed)"
                else:
                    e.reset_random(seed=self.seed)
                    self.start = "device-side sampler"
                if args.cbf or args.cbf_qp:
                    e.cbf_attach()
                self.envs.append(e)
        self.env = self.envs[0]
        self.D = self.env.D
        gen = torch.Generator(device=device).manual_seed(self.seed + 7919 * (self.env_base // max(1, B)))
        self.n_act = T * ((16 + T - 1) // T)  # whole chunks of T consecutive action blocks
        self.acts = torch.empty((self.n_act, B, N, 2), dtype=torch.float32, device=device)
        self.acts[..., 0] = torch.rand((self.n_act, B, N), generator=gen, device=device)                 # v_cmd ~ U[0, 1]
        self.acts[..., 1] = torch.rand((self.n_act, B, N), generator=gen, device=device) * 0.5 - 0.25    # delta_cmd ~ U[-0.25, 0.25] rad
        torch.cuda.synchronize()
        self.pf, self.pc = self.env.map.list_first[0], self.env.map.list_count[0]
        self.counter = 1
        self.gather = None
        self.gather_note = "disabled by --no-gather"
        self.gather_fail = None
        if with_exchange and not args.no_gather:
            try:  # the rollout exchange must never take the benchmark down: fall back to "no gather" and say so in the JSON line
                self.chunk_steps = T if T > 1 else args.chunk_steps
                self.gather = RolloutExchange(B, N, self.D, self.chunk_steps, device, force_collective=args.force_dist, mode=args.exchange)
                # one exchange before anything is timed (a failure here disables the exchange instead of the benchmark); with T > 1 it is a launch
                # like all the others (T steps), so that per-launch profiles average over equal launches
                if T > 1:
                    self.envs[0].step_autoreset_n_ptr(self.acts.data_ptr(), T, B * N * 2, self.gather.chunk().data_ptr(), B * (N * (self.D + 1) + 1), self.seed, 1 << 20,
                                                      self.pf, self.pc)
                    torch.cuda.synchronize()
                    self.gather.commit()
                else:
                    slot0 = self.gather.slot()
                    for k, e in enumerate(self.envs):
                        e.set_slab(slot0[k * Bs:(k + 1) * Bs])
                        e.step(self.acts[0][k * Bs:(k + 1) * Bs])
                    torch.cuda.synchronize()
                    self.gather.advance()
                    self.gather.flush(self.streams if S > 1 else None)
                self.gather.wait_all()
                torch.cuda.synchronize()
                for k, e in enumerate(self.envs):
                    e.auto_reset(seed=self.seed, counter=0, path_first=self.pf, path_count=self.pc)
                self.gather_note = (f"step kernel records (obs, reward, done) into a [{self.chunk_steps}, B, {N * (self.D + 1) + 1}] chunk buffer"
                                    + (("; one async all-to-all per chunk (rank r receives steps [r T/N, (r+1) T/N) of every rank's chunk), double buffered"
                                        if self.gather.mode == "alltoall" else "; one async gather per chunk to rank 0, double buffered")
                                       if self.gather.collective else " (single GPU: no exchange)"))
            except Exception as exc:  # noqa: BLE001
                self.gather = None
                for e in self.envs:
                    e.set_slab(None)
                self.gather_note = f"disabled: {type(exc).__name__}: {exc}"
                print(f"[bench] rollout exchange disabled: {exc}", file=sys.stderr)
        self.actors, self.act_bufs = [], []
        if args.policy:
            from sigmarl_amd.actor import Actor, make_mlp
            torch.manual_seed(0)
            mlp = make_mlp(self.D)
            for k, e in enumerate(self.envs):
                with torch.cuda.stream(self.streams[k]):
                    self.actors.append(Actor(mlp, low=[-1.0, -0.6109], high=[1.0, 0.6109], precision=args.policy_precision, mode=args.policy_mode))  # -/+ (max_speed, max_steering)
                    self.act_bufs.append(torch.zeros((Bs, N, 2), dtype=torch.float32, device=device))
        self.safe_bufs = [torch.zeros((Bs, N, 2), dtype=torch.float32, device=device) for _ in range(S)] if args.cbf_qp else []
        self.W = N * (self.D + 1) + 1
        if args.policy and S > 1:  # the shards' rollouts record into one [T, B, W] chunk: step stride B W, shard q starts at row q Bs (sigmaenv_set_rollout_slab_stride)
            for e in self.envs:
                e.set_rollout_slab_stride(B * self.W)
        self.act_ptrs = [[self.acts[q].data_ptr() + k * Bs * N * 2 * 4 for k in range(S)] for q in range(self.n_act)]
        self.shard_seeds = [self.seed for k in range(S)]
        self.fused = not (args.no_reset or args.separate_reset)
        HArr, PArr, SArr = C.c_void_p * S, C.c_void_p * S, C.c_uint64 * S
        self.h_arr = HArr(*[e.h for e in self.envs])
        self.seed_arr = SArr(*self.shard_seeds)
        self.act_arrs = [PArr(*ap_) for ap_ in self.act_ptrs]
        self.slab_arr = PArr()
        self.many = self.envs[0].lib.step_autoreset_many

    def run_steps(self, t0, n):
        """n steps starting at step index t0: launches of up to T steps (T > 1), else one launch per step and shard."""
        if self.T == 1 and self.policy_chunked():
            done = 0
            while done < n:  # sigmaenv_rollout(_f32): up to chunk_steps x (actor, head, fused step + record + resets) enqueued by ONE binding call
                k = min(self.chunk_steps if self.gather is not None else 32, n - done)
                multi = self.streams if self.S > 1 else None
                slab = self.gather.chunk(multi) if self.gather is not None else None
                if self.S == 1:
                    self.actors[0].rollout(self.env, k, slab=slab, seed=self.seed, counter0=self.counter, path_first=self.pf, path_count=self.pc)
                else:  # every shard's chain of k steps on its own stream; shard q records rows [q Bs, (q + 1) Bs) of every step's block of the ONE [T, B, W] chunk
                    for q, e in enumerate(self.envs):
                        sp = (slab.data_ptr() + q * self.Bs * self.W * 4) if slab is not None else None
                        self.actors[q].rollout(e, k, slab_ptr=sp, seed=self.shard_seeds[q], counter0=self.counter, path_first=self.pf, path_count=self.pc)
                self.counter += k
                if self.gather is not None:
                    self.gather.commit(k, multi)
                done += k
            return
        if self.T == 1:
            for t in range(n):
                self.one_step(t0 + t)
            return
        done = 0
        while done < n:
            k = min(self.T, n - done)
            self.run_chunk(t0 + done, k)
            done += k

    def policy_chunked(self):
        """--policy without a CBF launch between policy and step, one env shard: the C-side rollout loop enqueues the steps (the host is out of the loop: per-step
        Python calls cost 0.10 - 0.16 ms per step on the GPU box's host cores, more than the 0.11 ms of GPU work)"""
        a = self.args
        return bool(a.policy) and self.fused and not (a.cbf or a.cbf_qp) and not a.policy_per_step_calls

    def run_chunk(self, t0, k):
        """ONE launch: k <= T fused steps of every env, the record rows of the k steps into the chunk buffer, then the chunk's exchange."""
        e, B, N, W = self.env, self.B, self.N, self.W
        a0 = ((t0 // self.T) * self.T) % self.n_act  # a whole window of T consecutive action blocks
        ap = self.acts.data_ptr() + a0 * B * N * 2 * 4
        gather = self.gather
        slab = gather.chunk().data_ptr() if gather is not None else 0
        e.step_autoreset_n_ptr(ap, k, B * N * 2, slab, B * W, self.seed, self.counter, self.pf, self.pc)
        self.counter += k
        if gather is not None:
            try:
                gather.commit(k)
            except Exception as exc:  # noqa: BLE001 -- a failing exchange must not take the benchmark down: keep recording, stop exchanging
                print(f"[bench] rollout exchange failed, continuing without it: {exc}", file=sys.stderr)
                gather.collective = False
                gather.pending = [None, None]
                self.gather_fail = f"exchange failed at run time ({type(exc).__name__}); record kept, exchange disabled"
                gather.t = 0

    def one_step(self, t):
        args, S, Bs, W = self.args, self.S, self.Bs, self.W
        base = 0
        gather = self.gather
        if gather is not None:
            slot = gather.slot(self.streams if S > 1 else None)  # orders the shard streams behind the exchange that still reads this buffer
            base = slot.data_ptr()
        ap = self.act_ptrs[t % self.n_act]
        cnt = self.counter
        plain = self.fused and not args.policy and not (args.cbf or args.cbf_qp)
        if plain:  # ONE binding call: every shard's record target + fused step / record / reset launch
            for k in range(S):
                self.slab_arr[k] = (base + k * Bs * W * 4) if base else None
            rc = self.many(self.h_arr, S, self.act_arrs[t % self.n_act], self.slab_arr if base else None, self.seed_arr, cnt, self.pf, self.pc)
            if rc != 0:
                raise RuntimeError(f"sigmaenv_step_autoreset_many failed with code {rc}")
        elif base:
            for k, e in enumerate(self.envs):
                e.set_slab_ptr(base + k * Bs * W * 4)
        if self.fused and args.policy:  # policy on device, then the fused step on the actions it wrote
            for k, e in enumerate(self.envs):
                self.actors[k].forward(e, self.act_bufs[k], seed=self.shard_seeds[k], counter=cnt)
                if args.cbf_qp:
                    e.cbf_qp(self.act_bufs[k], self.safe_bufs[k])
                elif args.cbf:
                    e.cbf_rewards(self.act_bufs[k])
                e.step_autoreset_ptr(self.act_bufs[k].data_ptr(), self.shard_seeds[k], cnt, self.pf, self.pc)
        elif self.fused and (args.cbf or args.cbf_qp):  # margin rewards / QP of the action about to be applied, then the fused step
            a = self.acts[t % self.n_act]
            for k, e in enumerate(self.envs):
                if args.cbf_qp:
                    e.cbf_qp(a[k * Bs:(k + 1) * Bs], self.safe_bufs[k])
                else:
                    e.cbf_rewards(a[k * Bs:(k + 1) * Bs])
                e.step_autoreset_ptr(ap[k], self.shard_seeds[k], cnt, self.pf, self.pc)
        elif not self.fused:
            a = self.acts[t % self.n_act]
            for k, e in enumerate(self.envs):
                e.step(a[k * Bs:(k + 1) * Bs])
                if not args.no_reset:
                    e.auto_reset(seed=self.shard_seeds[k], counter=cnt, path_first=self.pf, path_count=self.pc)
        self.counter += 1
        if gather is not None:
            try:
                gather.advance(self.streams if S > 1 else None)
            except Exception as exc:  # noqa: BLE001 -- a failing exchange must not take the benchmark down: keep recording, stop exchanging
                print(f"[bench] rollout exchange failed, continuing without it: {exc}", file=sys.stderr)
                gather.collective = False
                gather.pending = [None, None]
                self.gather_fail = f"exchange failed at run time ({type(exc).__name__}); record kept, exchange disabled"
                gather.t = 0

    def finish_chunk(self):
        if self.gather is not None:
            self.gather.flush(self.streams if self.S > 1 else None)
            self.gather.wait_all()

    def arm_timing(self):
        for e in self.envs:
            e.step_time_ms()

    def kernel_timing(self):
        """{kernel id: (average launch duration in ms by HIP events, bracketed launches)} over the env shards, for every timed kernel that ran"""
        from sigmarl_amd import capi
        out = {}
        for kid in range(len(capi.KERNEL_NAMES)):
            timings = [e.kernel_time_ms(kid) for e in self.envs]
            n_launch = sum(n for _, n in timings)
            if n_launch:
                out[kid] = (sum(ms * n for ms, n in timings) / n_launch, n_launch)
        return out

    def episodes_reset(self):
        from sigmarl_amd import capi
        return sum(int(e.buffer(capi.BUF_TIMER)[:, 3].sum().item()) for e in self.envs)

    def agent_requests(self):
        """per-agent reset requests raised by the last step, and its entry / exit crossings"""
        from sigmarl_amd import capi
        return (sum(int(e.buffer(capi.BUF_COL_FLAGS)[..., 3].sum().item()) for e in self.envs),
                sum(int(e.buffer(capi.BUF_COL_FLAGS)[..., 1:3].sum().item()) for e in self.envs))

    def close(self):
        for e in self.envs:
            e.close()


class SurfaceRun:
    """The drop-in surface: ``ScenarioRoadTraffic`` (the mirror of sigmarl/scenarios/road_traffic.py) under an ``Environment``-shaped driver that calls it in
    vmas' order -- per step: N action tensors set, ``world.step()`` (ONE fused launch), ``reward(a)`` / ``observation(a)`` / ``info(a)`` for every agent with
    every returned tensor cloned (39 info entries per agent), ``done()`` incl. the resets (on the device: ``device_side_resets``).  Same interface as GpuRun."""

    def __init__(self, args, device, B):
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from vmas_env_shim import EnvironmentShim
        from sigmarl_amd.params import Parameters
        from sigmarl_amd.scenario import ScenarioRoadTraffic

        self.torch, self.args, self.B, self.N, self.S, self.Bs, self.T = torch, args, B, args.agents, 1, B, 1
        sc = ScenarioRoadTraffic()
        sc.parameters = Parameters(**make_params_kw(args, B))
        sc.device_side_resets = True
        self.shim = EnvironmentShim(sc, num_envs=B, device=device, seed=0, n_agents=args.agents)
        self.env = sc.env
        self.envs = [sc.env]
        self.D = sc.env.D
        self.start = "device-side sampler (device_side_resets=True)"
        gen = torch.Generator(device=device).manual_seed(1000)
        self.acts = [[torch.stack([torch.rand(B, generator=gen, device=device), torch.rand(B, generator=gen, device=device) * 0.5 - 0.25], dim=-1)
                      for _ in range(self.N)] for _ in range(8)]
        self.gather, self.gather_note, self.gather_fail, self.fused = None, "none (the consumer collects the callbacks' tensors itself)", None, False
        self.n_tensors = 0

    def run_steps(self, t0, n):
        for t in range(n):
            obs, rews, dones, infos = self.shim.step(self.acts[(t0 + t) % 8])
            self.n_tensors = len(obs) + len(rews) + 1 + sum(len(i) for i in infos)

    def finish_chunk(self):
        pass

    arm_timing = GpuRun.arm_timing
    kernel_timing = GpuRun.kernel_timing
    episodes_reset = GpuRun.episodes_reset
    agent_requests = GpuRun.agent_requests
    close = GpuRun.close


def cbf_valu_roofline(kernel_name, N, Bs, kernel_ms, hbm_roofline):
    """The CBF kernels (centralized QP, margin rewards) are bound by VALU ISSUE, not by HBM or a matrix pipe ("frac 0.0018 of HBM is not a bound for a
    projected-Newton kernel"): roofline = the kernel's VALU wavefront-instructions per second and SIMD against what a SIMD-32 can issue -- one wave64 instruction per
    2 cycles (MI355X guide) at the shader clock the kernel ran at in its PMC pass.  Instruction counts, clock, fp64 share and wait fractions come from the committed
    pass of this workload (profiles/valu_dominant_latest.json / valu_cbf_margin_latest.json, tools/make_valu_json.py), the duration from THIS run's HIP events (one
    bracket around the launches of a call: for the QP the lean launch and the launch for the envs it left over)."""
    src = "valu_dominant_latest.json" if "qp" in kernel_name else "valu_cbf_margin_latest.json"
    try:
        with open(os.path.join(ROOT, "profiles", src)) as f:
            vj = json.load(f)
        if not (vj.get("kernel", "") in kernel_name and vj.get("n_agents") == N and vj.get("envs_per_launch") == Bs):
            return {}
    except Exception:  # noqa: BLE001
        return {}
    clk = vj.get("shader_clock_hz_measured") or vj["clock_hz"]
    rate = vj["valu_insts_per_launch"] / (kernel_ms * 1e-3) / vj["n_simd"]  # wavefront instructions per second and SIMD
    peak = clk / 2.0
    f64 = vj.get("f64_insts_per_launch") or 0.0
    return {
        "bound": "valu", "achieved": rate / 1e9, "peak": peak / 1e9, "unit": "G wave64-inst/s/SIMD", "frac": rate / peak,
        "frac_nominal_clock": rate / (vj["clock_hz"] / 2.0), "shader_clock_hz_measured": vj.get("shader_clock_hz_measured"),
        "valu_insts_per_launch": vj["valu_insts_per_launch"], "f64_inst_share": f64 / vj["valu_insts_per_launch"], "f64_flops_per_launch": vj.get("f64_flops_per_launch"),
        "f64_flop_frac": (vj.get("f64_flops_per_launch") or 0.0) / (kernel_ms * 1e-3) / 78.6e12,
        "mean_active_lanes_per_valu_inst": vj.get("mean_active_lanes_per_valu_inst"), "wait_any_frac": vj.get("wait_any_frac"), "wait_inst_any_frac": vj.get("wait_inst_any_frac"),
        "hbm_frac": hbm_roofline.get("frac"), "hbm_achieved_gbps": hbm_roofline.get("achieved"),
        "valu_source": vj.get("source"),
        "achieved_basis": "the DOMINANT kernel of this workload (largest share of GPU time by HIP events) is bound by VALU issue: its VALU wavefront-instructions per launch "
                          "(rocprofv3 --pmc SQ_INSTS_VALU of this workload, committed pass, not this run) / its average duration in THIS run / 1024 SIMDs, against one wave64 "
                          "instruction per 2 cycles at the shader clock of the pass (SQ_BUSY_CYCLES / 32 / dispatch duration).  What keeps it below that is latency, not "
                          "issue: wait_any_frac of the wave-cycles wait on memory / LDS counters (profiles/r06_qp_lds.txt); hbm_frac restates the old HBM figure",
    }


def baseline_lines(args, device, torch, dist, head_step_s):
    """config.lines: the other single-GPU configurations of BASELINE.json in the driver's own line (every line of BASELINE.md section 5 but the headline
    was the builder's own run) -- config 4 (on_ramp_1, 32 agents x 8192 envs), config 5 (the centralized CBF-QP before every step), the mtv distance, the reference's
    own defaults (mtv + mask + noise).  Each: a fresh run object, ~60 ms of its own launches (sustained clocks), W warm-up + K timed steps between synchronisations,
    HIP-event kernel averages, the dominant kernel's roofline with the bound that kernel has.  Timed AFTER the headline's regions: `value` is untouched."""
    import copy

    from sigmarl_amd import capi as _capi

    specs = [
        ("config4_on_ramp_32x8192", dict(scenario="on_ramp_1", agents=32, envs_per_gpu=8192), 64, 32,
         "BASELINE config 4: on_ramp_1, 32 agents x 8192 envs (injected start: every env restarts every step -- a reset benchmark by construction; its shape on CPM runs at ~1.6e9)"),
        ("config5_cbf_qp", dict(cbf_qp=True), 48, 16, "BASELINE config 5: CPM, 16 agents x 4096 envs, centralized CBF-QP safety filter solved before every step (two env shards on two streams)"),
        ("distance_mtv", dict(distance="mtv"), 128, 32, "config 2 with the mtv distance (Parameters.is_use_mtv_distance)"),
        ("reference_defaults", dict(defaults=True), 128, 32, "config 2 with the reference's own defaults for this path (helper_common.py:66-79): mtv distance, is_apply_mask, is_obs_noise"),
    ]
    lines = []
    for name, over, K, W, what in specs:
        t_line = time.perf_counter()
        try:
            a = copy.copy(args)
            for k, v in over.items():
                setattr(a, k, v)
            plain = not (a.policy or a.cbf or a.cbf_qp)
            T = pick_chunk(K) if plain else 1
            os.environ["SIGMAENV_TIMING_STRIDE"] = str(max(1, min(32, (K // T) // 8)))  # (read when a handle is created: the HIP-event brackets of every stride-th launch)
            run = GpuRun(a, device, a.envs_per_gpu, 1, 0, with_exchange=True, T=T)
            run.arm_timing()
            run.run_steps(0, W)
            run.finish_chunk()
            torch.cuda.synchronize()
            el0 = timed(run, K, W, False, dist, torch, device)
            condition_device(run, torch, 60.0, el0 / K, T)
            run.run_steps(0, W)
            run.finish_chunk()
            torch.cuda.synchronize()
            run.kernel_timing()
            el = timed(run, K, W, False, dist, torch, device)
            kt = run.kernel_timing()
            N, B, D, Bs, S = a.agents, a.envs_per_gpu, run.D, run.Bs, run.S
            dom = max(kt, key=lambda k: kt[k][0] * kt[k][1]) if kt else _capi.KERNEL_STEP
            dms, dn = kt.get(dom, (0.0, 0))
            value = N * B * K / el
            bytes_per = algorithmic_bytes_per_agent_step(N, D)
            hbm = {"bound": "hbm", "achieved": bytes_per * value / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": bytes_per * value / 1e9 / HBM_PEAK_GBPS,
                   "algorithmic_bytes_per_agent_env_step": bytes_per}
            roof = dict(hbm)
            if dom in (_capi.KERNEL_CBF_QP, _capi.KERNEL_CBF_MARGIN):
                roof.update(cbf_valu_roofline(_capi.KERNEL_NAMES[dom], N, Bs, dms, hbm))
            roof.update({"kernel": _capi.KERNEL_NAMES[dom], "kernel_avg_ms": dms, "kernel_launches": dn, "steps_per_launch": T,
                         "kernel_time_share": {_capi.KERNEL_NAMES[k]: {"avg_ms": v[0], "launches_bracketed": v[1]} for k, v in kt.items()}})
            lines.append({"name": name, "workload": what, "value": value, "value_contract_style": N * B * K / el0, "unit": "agent-env-steps/s", "ms_per_step": el / K * 1e3, "steps": K, "warmup": W,
                          "n_agents": N, "envs_per_gpu": B, "env_shards_per_gpu": S, "steps_per_launch": T, "obs_dim": D, "dominant_kernel": _capi.KERNEL_NAMES[dom], "roofline": roof,
                          "clocks": "sustained (60 ms of the line's own launches before its warm-up; value_contract_style = the first W + K steps of the fresh run object)",
                          "wall_s": None})
            run.close()
        except Exception as exc:  # noqa: BLE001 -- a side line must never take the headline down
            lines.append({"name": name, "workload": what, "error": f"{type(exc).__name__}: {exc}"})
        lines[-1]["wall_s"] = time.perf_counter() - t_line
    return lines


def condition_device(run, torch, ms, step_s, per=1):
    """Device conditioning (--condition-ms): `ms` milliseconds of the run's own step launches, as a STEP COUNT derived from `step_s` (a MAX-reduced time: the same
    count on every rank, so the ranks issue the same number of chunk exchanges).  Returns the steps run."""
    if ms <= 0:
        return 0
    per = max(1, per)
    n = int(math.ceil(ms * 1e-3 / max(step_s, 1e-9) / per)) * per
    done = 0
    while done < n:  # bounded queue depth: a host synchronisation every 8 launches (tens of microseconds: far below the ~10 ms of idling that let the clocks drop)
        k = min(8 * per, n - done)
        run.run_steps(done, k)
        run.finish_chunk()
        torch.cuda.synchronize()
        done += k
    return n


def timed(run, steps, start_t, use_dist, dist, torch, device):
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run_steps(start_t, steps)
    t1 = time.perf_counter()
    run.finish_chunk()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("BENCH_DEBUG_TIMING"):
        print(f"[bench] timed region: enqueue {1e6 * (t1 - t0):.1f} us, total {1e6 * elapsed:.1f} us", file=sys.stderr)
    el = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    return float(el.item())


def dry_run(args):
    """The N-rank run without its GPU work (see --dry-run): fails loudly on any inconsistency, prints the contract's JSON line from rank 0."""
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from sigmarl_amd import capi
    from sigmarl_amd.shard import RolloutExchange, shard_range, slab_width

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("gloo")
        assert dist.get_rank() == rank and dist.get_world_size() == world
    B, N = args.envs_per_gpu, args.agents
    plain = not (args.policy or args.cbf or args.cbf_qp or args.no_reset or args.separate_reset)
    T = 1 if not plain else (args.chunk if args.chunk >= 1 else pick_chunk(args.steps))
    b0, b1 = shard_range(B * world, rank, world)
    assert (b0, b1) == (rank * B, (rank + 1) * B), (b0, b1)  # weak scaling: rank r owns envs [r B, (r + 1) B) -- its env_index_base
    Bd, D = min(B, 8), capi.obs_dim(2)
    chunk_steps = T if T > 1 else args.chunk_steps
    ex = RolloutExchange(Bd, N, D, chunk_steps, "cpu", mode=args.exchange)
    W = slab_width(N, D)
    t0 = time.perf_counter()
    n_chunks = 0
    for c0 in range(0, args.steps, chunk_steps):
        k = min(chunk_steps, args.steps - c0)
        buf = ex.chunk()
        for t in range(k):
            buf[t].fill_(float(1000 * rank + c0 + t))  # what the step kernel would record: tagged by (rank, step)
        ex.commit(k)
        n_chunks += 1
    ex.wait_all()
    # check the last chunk's exchange
    kbuf, valid = ex.completed[-1]
    c0 = (n_chunks - 1) * chunk_steps
    if ex.collective and ex.mode == "alltoall":
        mine = ex.time_slice(kbuf)  # [my steps, world * Bd, W]
        lo, hi = ex.slices[rank], ex.slices[rank + 1]
        assert tuple(mine.shape) == (hi - lo, world * Bd, W), tuple(mine.shape)
        for q in range(lo, min(hi, valid)):
            for r in range(world):
                got = mine[q - lo, r * Bd:(r + 1) * Bd]
                assert bool((got == float(1000 * r + c0 + q)).all()), (rank, q, r)
    elif ex.collective and rank == ex.dst:
        parts = ex.gathered(kbuf)
        for r in range(world):
            for q in range(valid):
                assert bool((parts[r][q] == float(1000 * r + c0 + q)).all()), (rank, q, r)
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    out = {
        "metric": f"env-steps/sec (agents x envs x steps), {'CPM' if args.scenario.startswith('cpm') else args.scenario} scenario, {N} agents",
        "value": 0.0, "unit": "agent-env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
        "config": {"workload": f"DRY RUN (no GPU work): {args.scenario} map, {N} agents x {B} envs per GPU ({B * world} envs total)", "n_agents": N, "envs_per_gpu": B,
                   "envs_total": B * world, "steps_per_launch": T, "exchange": ex.mode if ex.collective else "none", "time_slices": ex.slices,
                   "env_index_base": b0, "local_rank": local_rank, "chunks": n_chunks},
    }
    if world > 1:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        os.write(1, (json.dumps(out) + "\n").encode())
    os.dup2(2, 1)


def ensure_world(args):
    """`--gpus N` must mean N ranks.  Started without a rendezvous (no WORLD_SIZE: `python bench.py --gpus 8`) the script re-executes itself under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` -- the driver's own command line -- instead of timing one
    rank and printing an `n_gpus: 1` line; started under a launcher whose world size differs from --gpus it exits non-zero with the command to use."""
    world = os.environ.get("WORLD_SIZE")
    if world is None:
        if args.gpus <= 1:
            return
        import socket

        with socket.socket() as sk:  # a free rendezvous port on the loopback interface
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] --gpus {args.gpus} without a rendezvous: re-launching as `{' '.join(cmd)}`", file=sys.stderr)
        sys.stderr.flush()
        os.execv(sys.executable, cmd)
    if int(world) != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 "
              f"--master-port <port> bench.py --gpus {args.gpus} ...` (or plain `python bench.py --gpus {args.gpus}`, which does that itself)", file=sys.stderr)
        sys.exit(2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--agents", type=int, default=16)
    ap.add_argument("--scenario", default="cpm_entire", help="map (sigmarl_amd/assets/maps); maps that cannot hold --agents through the "
                    "reference's own reset start from the injected state (BASELINE config 4: --scenario on_ramp_1 --agents 32 --envs-per-gpu 8192)")
    ap.add_argument("--distance", choices=["c2c", "mtv"], default="c2c")
    ap.add_argument("--defaults", action="store_true", help="the reference's own defaults for this path (helper_common.py:66-79): mtv distance, is_apply_mask, is_obs_noise")
    ap.add_argument("--param", action="append", default=[], metavar="KEY=VALUE", help="override a Parameters field (JSON value), e.g. --param is_ego_view=false "
                    "--param is_obs_steering=true: the non-default observation rows run inside the same fused launch")
    ap.add_argument("--surface", action="store_true", help="time the DROP-IN SURFACE instead of the C-ABI: ScenarioRoadTraffic driven the way vmas' Environment drives "
                    "a scenario (tests/vmas_env_shim.py: set actions per agent, world.step(), reward / observation / info per agent -- every returned tensor cloned --, "
                    "done(), device-side resets) -- what sigmarl/mappo_cavs.py:166-184 sees per env.step()")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables it)")
    ap.add_argument("--sweep", action="store_true", help="also time the metric's batch sweep (envs per GPU in --sweep-envs, same agents / map) and "
                    "report it in `sweep`; the headline stays --envs-per-gpu")
    ap.add_argument("--sweep-envs", default="256,512,1024,2048,4096,8192,16384,32768")
    ap.add_argument("--sweep-steps", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=0, help="steps per launch (sigmaenv_step_autoreset_n); 0 = the largest divisor of --steps up to 32; 1 = one launch "
                    "per step (sigmaenv_step_autoreset); modes with a launch between the steps (--policy, --cbf, --cbf-qp, --separate-reset) always use 1")
    ap.add_argument("--condition-ms", type=float, default=200.0, help="after `value` (the contract's W warm-up + K timed steps on the device as the setup left it) is taken: this "
                    "many milliseconds of the SAME step launches, then W + K steps again -- the GPU's sustained clocks, what a running rollout loop sees (an MI355X that idled "
                    "for 10 ms is back at its idle clocks, and a launch of 20 steps -- 0.6 ms -- is over before they have risen: 29 us per step against 26.5 sustained).  "
                    "Reported as value_sustained / config.sustained; 0 skips it")
    ap.add_argument("--no-one-stream", "--no-compare", dest="no_compare", action="store_true",
                    help="skip the additional per-step-launch measurement reported in config.per_step_launch")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="after the headline: BASELINE config 3's workload on this ONE GPU -- R shards of --envs-per-gpu envs "
                    "(rank r = envs [r B, (r + 1) B) of the batch), each timed like the headline with its own rollout exchange; reported in config.emulated_ranks")
    ap.add_argument("--no-lines", dest="lines", action="store_false", help="skip config.lines (--no-compare skips them too: profile runs must hold the headline's launches only): the other single-GPU BASELINE configurations (config 4, config 5, mtv, the "
                    "reference's defaults), each timed for a few seconds AFTER the headline's regions and appended to the headline's JSON line")
    ap.add_argument("--no-reset", action="store_true", help="diagnostic: leave finished envs un-reset")
    ap.add_argument("--separate-reset", action="store_true", help="two launches per step (sigmaenv_step; sigmaenv_auto_reset) instead of the fused one")
    ap.add_argument("--no-gather", action="store_true", help="diagnostic: no rollout record (and no exchange for N > 1)")
    ap.add_argument("--streams", type=int, default=None, help="env shards per GPU, each stepped by its own handle on its own HIP stream "
                    "(envs are independent: same total work per step; the tail of one shard's launch -- its reset-heavy wavefronts -- overlaps the other's).  Default 2; "
                    "with --policy 2 for the fp32 split-product actor (every shard's chain of (actor, head, step) launches is enqueued by sigmaenv_rollout_f32 on its own "
                    "stream and records into its rows of the one [T, B, W] chunk -- sigmaenv_set_rollout_slab_stride: 0.1008 against 0.1047 ms per step on one stream) and 1 "
                    "for the exact-fp32 and bf16 actors (bf16: 0.0717 on one stream against 0.0731 on two)")
    ap.add_argument("--policy", action="store_true", help="widening (SURVEY 8f-3): the actor MLP runs on the device before every step "
                    "(sigmaenv_actor_forward) instead of replaying precomputed actions; reported in config.policy")
    ap.add_argument("--policy-precision", choices=["fp32", "bf16"], default="fp32", help="--policy: the reference's fp32 arithmetic or the bf16 inference variant")
    ap.add_argument("--policy-per-step-calls", action="store_true", help="--policy: one Actor.forward + one step call per step from Python instead of the C-side rollout loop")
    ap.add_argument("--policy-mode", choices=["split", "exact"], default="split", help="--policy with fp32: how the fp32 products are formed on the matrix cores -- split "
                    "(default: every operand as hi + lo fp16, three exact-product v_mfma_f32_32x32x16_f16 per fp32 product, fp32 accumulation) or exact (v_mfma_f32_32x32x2_f32 fma "
                    "chains at the fp32 vector rate); both are held to torch.nn fp32 within 1e-5 (tests/test_gpu_actor.py)")
    ap.add_argument("--cbf", action="store_true", help="widening (SURVEY 8f-4): rew_method='cbf' with the QP-free CBF margin reward "
                    "(sigmaenv_cbf_rewards before every step); reported in config.cbf")
    ap.add_argument("--cbf-group-size", type=int, default=0, help="--cbf-qp: solve the grouped CBF-QPs with this max_group_size instead of the centralized QP")
    ap.add_argument("--cbf-qp", action="store_true", help="BASELINE config 5: rew_method='cbf' with the centralized CBF-QP safety filter solved for "
                    "every env before every step (sigmaenv_cbf_qp); reported in config.cbf")
    ap.add_argument("--exchange", choices=["alltoall", "gather"], default="alltoall",
                    help="N > 1: how the rollout buffer is concatenated.  alltoall: distributed over the ranks by time slices (every rank "
                         "receives 1/N of the steps of ALL envs -- a data-parallel learner; the record crosses xGMI once, over all links); "
                         "gather: everything to rank 0 (the 7 links into one GPU bound the rate)")
    ap.add_argument("--chunk-steps", type=int, default=32, help="steps per rollout chunk exchanged (N > 1)")
    ap.add_argument("--force-dist", action="store_true", help="diagnostic: init RCCL and run the exchange even with one rank")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: everything AROUND the step of an N-rank run -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* handling, "
                    "rendezvous, env ranges, the chunk exchange (gloo, host buffers tagged by rank and step, checked), barrier / MAX reduction, ONE JSON line from "
                    "rank 0 (`dry_run: true`, `value: 0`) -- so that `torch.distributed.run ... bench.py --gpus 8` can be rehearsed on a box without GPUs")
    args = ap.parse_args()
    if args.streams is None:
        args.streams = (2 if args.policy_precision == "fp32" and args.policy_mode == "split" else 1) if args.policy else 2
    ensure_world(args)
    if args.dry_run:
        return dry_run(args)

    # the contract is ONE JSON line on stdout: libraries (RCCL prints a version banner) get stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    plain = not (args.policy or args.cbf or args.cbf_qp or args.no_reset or args.separate_reset)
    T = 1 if not plain else (args.chunk if args.chunk >= 1 else pick_chunk(args.steps))
    # HIP-event bracket around a sample of the step launches (each bracket costs a few microseconds): at least 8 samples per shard when there are that many
    os.environ.setdefault("SIGMAENV_TIMING_STRIDE", str(max(1, min(32, (args.steps // T) // 8))))
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
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    B, N = args.envs_per_gpu, args.agents
    if args.surface:
        T = 1
        args.no_compare = True
    run = SurfaceRun(args, device, B) if args.surface else GpuRun(args, device, B, world, rank, T=T)
    dist_label = "mtv" if make_params_kw(args, B)["is_use_mtv_distance"] else "c2c"
    # nothing but the warm-up steps runs on the GPU right before the timed region (no reduction kernel, no device-to-host copy: both would let the
    # queue run empty and the first timed launch pay for it)
    run.arm_timing()  # arms the HIP-event bracketing of the step launches (on the env's stream)
    resets_before = run.episodes_reset()  # (a torch reduction + a device-to-host copy: BEFORE everything that is timed)
    steps_counted = args.warmup + args.steps
    # (1) `value`: the contract to the letter on the device as the setup left it -- W warm-up steps, K timed steps
    run.run_steps(0, args.warmup)
    run.finish_chunk()
    torch.cuda.synchronize()
    run.kernel_timing()  # drops the warm-up launches' brackets
    elapsed = timed(run, args.steps, args.warmup, use_dist, dist, torch, device)
    ktimes = run.kernel_timing()
    sustained = None
    if args.condition_ms > 0 and not args.surface:
        # (2) what a running rollout loop sees: the same launches for --condition-ms (a step count derived from the MAX-reduced time of (1): equal on every rank, so
        # that the ranks issue the same number of chunk exchanges), then W + K steps again
        n_cond = condition_device(run, torch, args.condition_ms, elapsed / args.steps, T)
        run.run_steps(0, args.warmup)
        run.finish_chunk()
        torch.cuda.synchronize()
        run.kernel_timing()
        el_s = timed(run, args.steps, args.warmup, use_dist, dist, torch, device)
        kt_s = run.kernel_timing()
        from sigmarl_amd import capi as _capi0
        k_ms, k_n = kt_s.get(_capi0.KERNEL_STEP, (0.0, 0))
        sustained = {"value": args.agents * B * world * args.steps / el_s, "ms_per_step": el_s / args.steps * 1e3,
                     "step_kernel_ms_per_launch": k_ms, "step_kernel_launches_bracketed": k_n, "conditioning_steps": n_cond,
                     "warmup_effective_steps": 2 * args.warmup + args.steps + n_cond,
                     "note": f"the same W warm-up + K timed steps, measured AFTER `value` and {args.condition_ms:g} ms ({n_cond} steps) of the same step launches: the GPU's "
                             "sustained clocks, what a rollout loop sees after its first milliseconds"}
        steps_counted += args.warmup + args.steps + n_cond
    from sigmarl_amd import capi as _capi
    # the kernel with the largest share of GPU time in the timed region names the roofline (every timed kernel is bracketed with the same stride)
    dom = max(ktimes, key=lambda k: ktimes[k][0] * ktimes[k][1]) if ktimes else _capi.KERNEL_STEP
    kernel_ms, n_launch = ktimes.get(_capi.KERNEL_STEP, (0.0, 0))
    dones = (run.episodes_reset() - resets_before) * args.steps / max(1, steps_counted)  # (finished episodes of all the steps run so far, pro rata)
    req_last, entry_exit_last = run.agent_requests()
    S, Bs, D = run.S, run.Bs, run.D
    total_agent_steps = N * B * world * args.steps
    value = total_agent_steps / elapsed
    value_per_gpu = value / world
    bytes_per = algorithmic_bytes_per_agent_step(N, D)
    achieved = bytes_per * value_per_gpu / 1e9  # GB/s of ONE GPU: SURVEY.md section 8(d)'s per-unit figure x the units it processes per second
    # what the launch stores on top of section 8(d)'s list: the rollout record row of every env, and the whole record of every agent of a
    # re-placed env once more (320 + 5 N bytes)
    reset_bytes = (320 + 5 * N) * N * (dones / max(1, args.steps)) if run.fused else 0.0
    slab_bytes = 4.0 * (N * (D + 1) + 1) * B if run.gather is not None else 0.0
    step_s = elapsed / args.steps
    achieved_incl = (bytes_per * N * B + reset_bytes + slab_bytes) / step_s / 1e9
    steps_per_launch = args.steps / float(-(-args.steps // T))  # (== T when --steps is whole launches, the default)
    per_launch_bytes = bytes_per * N * Bs * steps_per_launch
    traffic, traffic_source = None, None
    try:  # HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_passes.sh), corrected as the MI355X guide says
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            tr = json.load(f)
        if tr.get("n_agents") == N and tr.get("envs_per_launch") == Bs and tr.get("distance") == dist_label and tr.get("obs_dim", 32) == D and tr.get("scenario", "cpm_entire") == args.scenario and (T > 1) == (float(tr.get("steps_per_launch", 1)) > 1):
            # (a T-step launch writes the per-step outputs of its LAST step only: its traffic says nothing about one-launch-per-step runs, and vice versa)
            # the profile's launches may hold another number of steps: scale by the steps of ONE launch of this run
            traffic = tr["hbm_bytes_per_launch"] / float(tr.get("steps_per_launch", 1)) * steps_per_launch
            traffic_source = "profiles/traffic_latest.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (not this run), per launch"
    except Exception:  # noqa: BLE001
        pass
    valu = {}
    try:  # issue-side figures of the dominant kernel from the committed SQ passes (tools/make_valu_json.py)
        with open(os.path.join(ROOT, "profiles", "valu_latest.json")) as f:
            vj = json.load(f)
        if vj.get("n_agents") == N and vj.get("envs_per_launch") == Bs and vj.get("scenario", "cpm_entire") == args.scenario:
            # the per-launch counts of the profile (per step of its launches) at this run's step rate
            vs = float(vj.get("steps_per_launch", 1))
            launches_per_s = S / step_s
            # the kernel's VALU instruction RATE per SIMD against the rate of a pure all-VGPR v_fma_f32 stream at the same occupancy (profiles/valu_calibration.json: the
            # calibration VERDICT r4 asked for -- SQ_ACTIVE_INST_VALU turned out to count instructions, so round 4's "valu_issue_frac 0.94" was this rate x 4 nominal cycles)
            inst_rate = vj["valu_insts_per_launch"] / vs * launches_per_s / vj["n_simd"]
            try:
                with open(os.path.join(ROOT, "profiles", "valu_calibration.json")) as f:
                    cal = json.load(f)
            except Exception:  # noqa: BLE001
                cal = {}
            # against the MACHINE: a wave64 VALU instruction issues in 2 cycles on a SIMD-32 (MI355X guide: v_fma_f32 2 cyc), so a SIMD takes clock / 2
            # of them per second; the clock is the one the kernel ran at in the PMC pass (SQ_BUSY_CYCLES / 32 shader engines / the dispatch's duration), not the
            # nominal 2.4 GHz.  x mean active lanes / 64 = the share of lane-cycles that carry work.
            clk = vj.get("shader_clock_hz_measured") or vj["clock_hz"]
            valu = {
                "valu_issue_frac_arch": inst_rate * 2.0 / clk, "valu_issue_frac_arch_nominal_clock": inst_rate * 2.0 / vj["clock_hz"],
                "valu_lane_cycle_frac_arch": inst_rate * 2.0 / clk * vj.get("mean_active_lanes_per_valu_inst", 64.0) / 64.0,
                "shader_clock_hz_measured": vj.get("shader_clock_hz_measured"), "valu_cycles_per_inst_at_measured_clock": clk / inst_rate if inst_rate > 0 else None,
                "valu_inst_per_s_per_simd": inst_rate,
                "valu_rate_vs_vgpr_fma_stream": (inst_rate / cal["vgpr_fma_stream_inst_per_s_per_simd"]) if cal.get("vgpr_fma_stream_inst_per_s_per_simd") else None,
                "valu_nominal_cycles_per_inst": vj["clock_hz"] / inst_rate if inst_rate > 0 else None,
                "fp32_flop_frac": vj["fp32_flops_per_launch"] / vs * launches_per_s / (FP32_PEAK_TFLOPS * 1e12),
                "valu_insts_per_launch": vj["valu_insts_per_launch"] / vs * steps_per_launch, "valu_source": vj.get("source", "profiles/valu_latest.json") + " (not this run)",
            }
    except Exception:  # noqa: BLE001
        pass
    loop_kind = ("drop-in surface, one launch per step" if args.surface else
                 "closed loop: policy on device before every step" if args.policy else
                 "CBF-QP filter launch before every step" if args.cbf_qp else "CBF margin launch before every step" if args.cbf else
                 f"open loop, {T} step{'s' if T > 1 else ''} per launch")
    out = {
        "metric": f"env-steps/sec (agents x envs x steps), {'CPM' if args.scenario.startswith('cpm') else args.scenario} scenario, {N} agents ({loop_kind})",
        "value": value, "unit": "agent-env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # `value` is the contract measurement (W warm-up + K timed steps on the device as the setup left it: warmup_effective_steps == warmup).  What a rollout loop
        # that has been running for a while sees (the GPU's sustained clocks) is reported BESIDE it, never as `value`:
        "warmup_effective_steps": args.warmup,
        "value_semantics": ("value = the contract measurement (W warm-up + K timed steps on the device as the setup left it), as in rounds 1-3 and as round 4's "
                            "config.cold_start; value_sustained = the same steps after --condition-ms of the same launches, which is what round 4 printed as `value` "
                            "(compare BENCH_r04 value 2.46e9 with this line's value_sustained, and BENCH_r04 config.cold_start 2.195e9 with this line's value)"),
        **({"value_sustained": sustained["value"], "ms_per_step_sustained": sustained["ms_per_step"],
            "warmup_effective_steps_sustained": sustained["warmup_effective_steps"]} if sustained is not None else {}),
        "config": {
            "workload": f"{args.scenario} map, {N} agents x {B} envs per GPU ({B * world} envs total), {dist_label} distance, "
                        f"rew_method={make_params_kw(args, B)['rew_method']}{params_note(args)}, dt=0.05, obs_dim={D}, start: {run.start}, "
                        + ("VMAS plugin surface: ScenarioRoadTraffic under the Environment-shaped driver (per step: world.step() = one fused launch, then reward / "
                           f"observation / info of every agent cloned -- {run.n_tensors} tensors -- and done() with device-side resets) " if args.surface else
                           "fused step + device-side reset of finished envs ")
                        + ((f"({T} steps per launch)" if T > 1 else "(one launch per step)") if run.fused else "(two launches)" if not args.no_reset else "(resets disabled)")
                        + ((" + rollout record" + ((" + " + run.gather.mode) if run.gather.collective else "")) if run.gather else ""),
            "n_agents": N, "envs_per_gpu": B, "envs_total": B * world, "distance": dist_label, "scenario": args.scenario, "env_shards_per_gpu": S, "steps_per_launch": T,
            "policy": (f"actor MLP {D}-256-256-256-4 ({args.policy_precision}" + (f", {args.policy_mode} products" if args.policy_precision == "fp32" else "")
                       + ") on device before every step" if args.policy
                       else "none in the timed region (precomputed actions resident in HBM)"),
            **({"policy_enqueue": (f"sigmaenv_rollout{'_f32' if args.policy_precision == 'fp32' else ''}: up to {getattr(run, 'chunk_steps', 32)} x (actor, head, fused step + record + resets) per binding call"
                                   if getattr(run, "policy_chunked", lambda: False)() else "one Actor.forward + one step call per step and env shard from Python")} if args.policy else {}),
            **({"cbf": ("centralized CBF-QP safety filter of every env (sigmaenv_cbf_qp: 2 N controls, lane + pair constraints, projected "
                        "Newton in float64) solved before every step; the step penalises the deviation from the safe action"
                        + (f"; grouped QPs, max_group_size {args.cbf_group_size}" if args.cbf_group_size > 0 else "") if args.cbf_qp else
                        "QP-free CBF margin reward (sigmaenv_cbf_rewards: 3 circles per vehicle, 9-point fp16 pseudo-distance stencils to both "
                        "boundaries, float64 margins) launched before every step")} if (args.cbf or args.cbf_qp) else {}),
            "actions": ("one tensor per agent handed to env.step() (the surface's contract)" if args.surface else
                        f"open loop: precomputed actions resident in HBM, {T} step(s) per launch -- a closed-loop rollout (policy or CBF filter between the steps) "
                        "runs at the config.per_step_launch / --policy rates"),
            "resets_per_step_per_gpu": dones / max(1, args.steps),
            "agent_reset_requests_last_step": req_last, "entry_exit_crossings_last_step": entry_exit_last,
            "rollout_gather": run.gather_fail or run.gather_note,
            **({"sustained": sustained} if sustained is not None else {}),
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "sigmaenv_step_wave_kernel", "kernel_avg_ms": kernel_ms, "kernel_launches": n_launch, "launches_per_step": S / steps_per_launch, "steps_per_launch": steps_per_launch,
            "kernel_ms_per_step": kernel_ms / steps_per_launch,
            "algorithmic_bytes_per_agent_env_step": bytes_per, "algorithmic_bytes_per_launch": per_launch_bytes,
            "achieved_per_launch": (per_launch_bytes / (kernel_ms * 1e-3) / 1e9) if kernel_ms > 0 else None,
            "achieved_incl_record": achieved_incl,
            # what the kernel actually moves: the PMC-measured HBM bytes of one launch / its duration.  A T-step launch keeps the tile in LDS between its steps and
            # writes section 8(d)'s per-step outputs (distance rows, collision rows, closest indices ...) for the LAST step only, so this is BELOW `achieved`
            "achieved_measured": (traffic / (kernel_ms * 1e-3) / 1e9) if (traffic and kernel_ms > 0) else None,
            "frac_measured": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and kernel_ms > 0) else None,
            **({"frac_sustained": bytes_per * sustained["value"] / world / 1e9 / HBM_PEAK_GBPS} if sustained is not None else {}),
            "traffic_over_algorithmic": (traffic / per_launch_bytes) if traffic else None,
            "achieved_basis": "algorithmic bytes per agent-env-step (SURVEY.md 8d: 44 + 251 + 5 N) x agent-env-steps/s of one GPU; achieved_per_launch = the same "
                              "bytes of ONE launch / its average duration by HIP events (launches of different shards overlap); achieved_incl_record adds the "
                              "rollout record rows and the rewrites of re-placed envs",
            **valu,
        },
    }
    out["roofline"]["kernel_time_share"] = {_capi.KERNEL_NAMES[k]: {"avg_ms": v[0], "launches_bracketed": v[1]} for k, v in ktimes.items()}
    if dom != _capi.KERNEL_STEP:
        # The dominant kernel is not the step: name IT, with its own algorithmic traffic per launch (what it must read and write per vehicle) over its
        # own average duration.  All of these are compute-bound kernels (fp64 Newton solve / fp16-fp64 margin stencils / MFMA): the HBM fraction says so.
        dms, dn = ktimes[dom]
        per_vehicle = {_capi.KERNEL_CBF_QP: 44 + 16, _capi.KERNEL_CBF_MARGIN: 44 + 12, _capi.KERNEL_MLP32: 4 * D + 16, _capi.KERNEL_ACTOR_BF16: 4 * D + 8}[dom]
        dom_bytes = per_vehicle * N * Bs
        step_r = dict(out["roofline"])
        out["roofline"] = {
            "bound": "hbm", "achieved": dom_bytes / (dms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": dom_bytes / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "traffic": None, "kernel": _capi.KERNEL_NAMES[dom], "kernel_avg_ms": dms, "kernel_launches": dn, "launches_per_step": S,
            "algorithmic_bytes_per_agent_env_step": per_vehicle, "algorithmic_bytes_per_launch": dom_bytes,
            "achieved_basis": "the DOMINANT kernel of this workload (largest share of GPU time by HIP events): bytes it must move per vehicle (state 32 + action 8 + path id 4 "
                              "in; safe + nominal action / reward channels / action out) x vehicles per launch / its average launch duration.  It is compute-bound "
                              "(float64 projected-Newton solve, fp16 / float64 margin stencils, or MFMA), which is what the small HBM fraction states",
            "kernel_time_share": step_r["kernel_time_share"], "step_kernel": {k: step_r[k] for k in ("achieved", "frac", "kernel_avg_ms", "kernel_launches", "algorithmic_bytes_per_launch")},
        }
        if dom in (_capi.KERNEL_MLP32, _capi.KERNEL_ACTOR_BF16):
            # the network kernels are matrix work: their roofline is the matrix pipe they issue on.  fp32 "split" issues three fp16 MFMAs per fp32 product (2.5 PFLOP/s
            # pipe), "exact" one fp32 MFMA (157.3 TFLOP/s), the bf16 variant one bf16 MFMA (2.5 PFLOP/s)
            macs = N * Bs * (D * 256 + 2 * 256 * 256 + 256 * 4)
            split = dom == _capi.KERNEL_MLP32 and args.policy_mode == "split"
            issued = 2.0 * macs * (3 if split else 1)
            peak = FP32_PEAK_TFLOPS if (dom == _capi.KERNEL_MLP32 and not split) else 2500.0
            out["roofline"].update({
                "bound": "mfma", "achieved": issued / (dms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": issued / (dms * 1e-3) / 1e12 / peak,
                "fp32_equivalent_tflops": 2.0 * macs / (dms * 1e-3) / 1e12, "hbm_frac": dom_bytes / (dms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "achieved_basis": "the DOMINANT kernel of this workload is the actor network: matrix-pipe FLOP issued per launch (2 x MACs of 32-256-256-256-4 per row; x 3 in split "
                                  "mode: hi hi + hi lo + lo hi) / its average launch duration by HIP events, against the dense peak of the pipe it issues on "
                                  "(fp16 / bf16 2.5 PFLOP/s, fp32 157.3 TFLOP/s); fp32_equivalent_tflops counts every fp32 product once"})
        out["roofline"].update(cbf_valu_roofline(_capi.KERNEL_NAMES[dom], N, Bs, dms, out["roofline"]))
    run.close()
    if T > 1 and not args.no_compare and world == 1 and not use_dist:
        # the same workload with ONE launch per step (two env shards on two streams, the round-2 form): what the step loop inside the kernel buys
        r1 = GpuRun(args, device, B, world, rank, T=1)
        condition_device(r1, torch, args.condition_ms, step_s)
        r1.run_steps(0, min(args.warmup, 32))
        r1.finish_chunk()
        # (an auxiliary leg: the better of two repetitions -- on some boxes the first timed region of a freshly created pair of shard streams carries a one-time stall
        # of 30-60 ms, profiles/r05_bench.json against r05_lines_bench.json)
        el1 = min(timed(r1, args.steps, args.warmup, use_dist, dist, torch, device), timed(r1, args.steps, args.warmup + args.steps, use_dist, dist, torch, device))
        out["config"]["per_step_launch"] = {"ms_per_step": el1 / args.steps * 1e3, "value": total_agent_steps / el1, "env_shards_per_gpu": r1.S,
                                            "note": "same steps, one launch per step and env shard (sigmaenv_step_autoreset), timed after the headline regions at the GPU's "
                                                    "sustained clocks (compare with value_sustained, not with value); the better of two repetitions"}
        # the form that writes ALL of SURVEY 8(d)'s outputs every step (a T-step launch writes the per-step outputs of its last step only: traffic_over_algorithmic < 1)
        out["roofline"]["frac_full_outputs"] = bytes_per * out["config"]["per_step_launch"]["value"] / world / 1e9 / HBM_PEAK_GBPS
        out["roofline"]["frac_full_outputs_note"] = ("algorithmic bytes x config.per_step_launch.value / 8 TB/s: one launch per step materialises every per-step output; "
                                                     "`frac` credits the T-step launch with stores it elides (see traffic_over_algorithmic / frac_measured)")
        r1.close()
    if args.lines and not args.no_compare and world == 1 and not use_dist and plain and args.scenario == "cpm_entire" and N == 16 and not args.param and not args.defaults and args.distance == "c2c":
        out["config"]["lines"] = baseline_lines(args, device, torch, dist, step_s)
    if args.emulate_ranks > 1 and world == 1:
        # BASELINE config 3 (16 agents x 32768 envs over 8 GPUs) on the one GPU of this box: rank r's shard -- envs [r B, (r + 1) B) of the batch, the
        # same seed, its own rollout exchange (RCCL with world size 1 under --force-dist) -- timed like the headline, one rank after the other
        shards = []
        for r in range(args.emulate_ranks):
            rr = GpuRun(args, device, B, 1, 0, T=T, env_base=r * B)
            condition_device(rr, torch, args.condition_ms, step_s, T)
            rr.run_steps(0, args.warmup)
            rr.finish_chunk()
            el = timed(rr, args.steps, args.warmup, use_dist, dist, torch, device)
            shards.append(el / args.steps * 1e3)
            rr.close()
        out["config"]["emulated_ranks"] = {
            "ranks": args.emulate_ranks, "envs_total": B * args.emulate_ranks, "ms_per_step_per_rank": shards,
            "value_if_concurrent": N * B * args.emulate_ranks / (max(shards) * 1e-3),
            "note": "every rank's shard of the sharded batch stepped on THIS GPU, one after the other (same seed, env_index_base = r B); value_if_concurrent = "
                    "all ranks' agent-env-steps / the slowest rank's time -- what N GPUs deliver when the exchange hides behind the steps; NOT a measured multi-GPU number",
        }
    if args.sweep:
        sweep = []
        for Bx in [int(x) for x in args.sweep_envs.split(",") if x]:
            Tx = 1 if not plain else (args.chunk if args.chunk >= 1 else pick_chunk(args.sweep_steps))
            r = GpuRun(args, device, Bx, world, rank, T=Tx)
            condition_device(r, torch, args.condition_ms, step_s * Bx / B, Tx)
            r.run_steps(0, 16)
            r.finish_chunk()
            el = timed(r, args.sweep_steps, 16, use_dist, dist, torch, device)
            sweep.append({"envs_per_gpu": Bx, "value": N * Bx * world * args.sweep_steps / el, "ms_per_step": el / args.sweep_steps * 1e3,
                          "env_shards_per_gpu": r.S, "steps_per_launch": Tx, "roofline_frac": bytes_per * (N * Bx * args.sweep_steps / el) / 1e9 / HBM_PEAK_GBPS})
            r.close()
        out["sweep"] = sweep
    if rank == 0 and args.cpu_seconds > 0 and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, B, args.cpu_seconds)
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        os.write(1, (json.dumps(out) + "\n").encode())
    os.dup2(2, 1)  # RCCL prints its version banner at exit: keep it off stdout as well


if __name__ == "__main__":
    main()
