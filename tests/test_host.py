"""CPU tests (-m "not gpu"): host logic, the C-ABI library's exported symbols, env sharding + the rollout-slab exchange on gloo."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_binding as ob
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, check_supported, make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parameters_roundtrip_and_defaults(tmp_path):
    p = Parameters(n_agents=16, scenario_type="cpm_entire", rew_method="ttc_sparse")
    d = json.loads(json.dumps(p.to_dict()))
    q = Parameters.from_dict(d)
    assert q.to_dict() == p.to_dict()
    f = tmp_path / "cfg.json"
    f.write_text(json.dumps({"n_agents": 4, "dt": 0.1, "scenario_type": "cpm_mixed", "is_apply_mask": False, "max_steps": 128}))
    r = Parameters.from_json(str(f))
    assert r.dt == 0.1 and r.frames_per_batch == r.num_vmas_envs * 128 and r.model_name == "reward0.00"
    with pytest.raises(TypeError):
        Parameters(not_a_field=1)


def test_rew_flags_follow_the_reference_string_tests():
    f = capi.rew_flags_from_method
    assert f("distance") == capi.REW_DISTANCE
    assert f("sparse") == capi.REW_EXACT_SPARSE | capi.REW_HAS_SPARSE
    assert f("ttc_sparse") == capi.REW_TTC | capi.REW_HAS_SPARSE
    assert f("distance_sparse") == capi.REW_DISTANCE | capi.REW_HAS_SPARSE
    assert f("cbf") == capi.REW_CBF_QP and f("cbf", False) == capi.REW_CBF  # is_solve_qp (default True) selects the QP penalty
    assert f("cbf_sparse", False) == capi.REW_CBF | capi.REW_HAS_SPARSE


def test_unsupported_configurations_fail_loudly():
    ok = Parameters(is_apply_mask=False)
    check_supported(ok)
    for kw in (dict(is_apply_mask=False, is_partial_observation=False), dict(is_apply_mask=False, is_using_cbf_training=True, is_grouping_agents=True, is_solve_qp=False),
               dict(is_apply_mask=False, n_points_short_term=9), dict(is_apply_mask=False, n_points_short_term=0)):
        with pytest.raises(NotImplementedError):
            check_supported(Parameters(**kw))
    check_supported(Parameters(is_apply_mask=False, n_points_short_term=5))  # a build constant: its own library build (capi.variant_path)
    check_supported(Parameters(is_apply_mask=False, is_using_opponent_modeling=True))  # placeholder columns + sigmaenv_opponent_fill
    check_supported(Parameters(is_apply_mask=False, is_ego_view=False))  # bird view without the lanelet-relation mask
    check_supported(Parameters(is_apply_mask=True, is_ego_view=False))  # ... and with it (sigmaenv_set_lanelets)
    check_supported(Parameters(is_apply_mask=False, is_obs_steering=True, is_observe_vertices=False))  # observation switches
    check_supported(Parameters(is_apply_mask=True, scenario_type="cpm_entire"))  # the reference's default (distance mask)
    check_supported(Parameters(is_apply_mask=True, scenario_type="intersection_1"))
    check_supported(Parameters(is_apply_mask=False, is_using_cbf_testing=True))  # CBF-QP safety filter at test time
    check_supported(Parameters(is_apply_mask=False, is_using_cbf_training=True))  # centralized CBF-QP (is_solve_qp defaults to True)
    check_supported(Parameters(is_apply_mask=False, is_using_cbf_training=True, is_solve_qp=False, rew_method="cbf"))  # QP-free margin reward


def test_config_mirrors_init_params():
    mp = load_map("cpm_entire")
    c = make_config(Parameters(n_agents=16, scenario_type="cpm_entire", is_apply_mask=False, is_use_mtv_distance=False), mp, 4096)
    assert (c.n_envs, c.n_agents, c.n_nearing, c.has_entry_exit, c.distance_type) == (4096, 16, 2, 0, capi.DIST_C2C)
    assert abs(c.threshold_near_other_agents_high - 0.3) < 1e-7 and abs(c.penalty_collide_with_agents + 1.0) < 1e-7
    m = make_config(Parameters(n_agents=3, scenario_type="on_ramp_1", is_apply_mask=False, is_use_mtv_distance=True, n_nearing_agents_observed=5), load_map("on_ramp_1"), 2)
    assert m.n_nearing == 2 and m.has_entry_exit == 1 and abs(m.threshold_near_other_agents_high - 0.22) < 1e-7
    assert abs(m.lane_width - 0.15) < 1e-7  # make_world's scenario_type kwarg defaults to cpm_entire (road_traffic.py:116-123)
    assert capi.obs_dim(2) == 32
    # full observation: the reference reshapes all nine feature tensors of the N agents to [B, n_nearing, -1] (observation_provider_rt.py:790-816): N % n_nearing != 0 raises there
    # even when every INCLUDED width splits (6 agents, 4 chunks: vertices 48, velocities 12, distances 36)
    full = capi.OBS_FULL | capi.OBS_BIRD_VIEW
    assert capi.obs_dim(2, full, 3, 4) == 70
    for n, k in ((6, 4), (5, 2), (16, 3)):
        with pytest.raises(ValueError):
            capi.obs_dim(k, full, 3, n)


def test_map_tables():
    mp = load_map("cpm_entire")
    assert mp.n_paths == 72 and mp.list_count == {0: 40, 1: 24, 2: 4, 3: 4}
    assert int(mp.n_center[:40].min()) == 125 and int(mp.n_center[:40].max()) == 177 and bool(mp.is_loop[:40].all())
    assert mp.global_path(1, 3) == 43 and (mp.world_x_dim, mp.world_y_dim) == (4.5, 4.0)
    for name in ("intersection_1", "on_ramp_1", "roundabout_2"):
        t = load_map(name)
        assert not t.is_loop.any() and t.parser_lane_width == 0.25


def test_hip_library_exports_the_declared_abi():
    """The built .so loads (no GPU needed for that) and exports every function include/sigmaenv.h declares."""
    so = capi.DEFAULT_LIB
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.dirname(so)])
    import torch  # noqa: F401  (HIP runtime load order, see capi.load_library)

    lib = ctypes.CDLL(so)
    for sym in capi.exported_symbols():
        assert hasattr(lib, sym), sym
    hdr = open(os.path.join(ROOT, "include", "sigmaenv.h")).read()
    for sym in capi.exported_symbols():
        assert sym + "(" in hdr, f"{sym} missing from include/sigmaenv.h"
    import re

    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(sigmaenv_\w+)\(", hdr, flags=re.M))
    assert declared == set(capi.exported_symbols()), declared ^ set(capi.exported_symbols())  # and the other way round: nothing declared is unbound
    for ns in (2, 5):  # the n_points_short_term build variants export the same ABI and report their constant
        vp = capi.variant_path(ns)
        if os.path.exists(vp):
            v = ctypes.CDLL(vp)
            assert all(hasattr(v, sym) for sym in capi.exported_symbols())
            v.sigmaenv_n_short_term.restype = ctypes.c_int
            assert v.sigmaenv_n_short_term() == ns
    lib.sigmaenv_obs_dim.restype = ctypes.c_int
    assert lib.sigmaenv_obs_dim(2) == 32
    # every built library carries the id of the sources of this tree (the GPU box uses the prebuilt files: a stale one must fail loudly, capi.check_build_id)
    for path in [so] + [capi.variant_path(ns) for ns in (2, 5) if os.path.exists(capi.variant_path(ns))]:
        v = ctypes.CDLL(path)
        v.sigmaenv_build_id.restype = ctypes.c_char_p
        assert v.sigmaenv_build_id().decode() == capi.source_build_id(), f"{path} is stale: rebuild it (make -C sigmarl_amd/csrc [NS=k])"
    # create() without a device must fail cleanly, not crash
    cfg = make_config(Parameters(n_agents=2, scenario_type="cpm_entire", is_apply_mask=False), load_map("cpm_entire"), 1)
    m = load_map("cpm_entire").as_struct()
    h = ctypes.c_void_p()
    rc = lib.sigmaenv_create(ctypes.byref(cfg), ctypes.byref(m), 0, None, ctypes.byref(h))
    if not torch.cuda.is_available():
        assert rc == capi_err("ENODEV") and not h.value


def capi_err(name):
    return {"EINVAL": -22, "ENOMEM": -12, "EHIP": -5, "ENODEV": -19}[name]


def test_product_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sigmarl_amd.env import SigmaEnv

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SigmaEnv(Parameters(n_agents=2, scenario_type="cpm_entire", is_apply_mask=False), n_envs=1)


def test_oracle_auto_reset_invariants():
    """Device-style sampler spec (shared with the HIP kernel): feasible, on-path, deterministic per (seed, counter)."""
    mp = load_map("cpm_entire")
    cfg = make_config(Parameters(n_agents=16, scenario_type="cpm_entire", is_apply_mask=False, is_use_mtv_distance=False), mp, 64)
    a, b = ob.OracleEnv(cfg, mp), ob.OracleEnv(cfg, mp)
    for e in (a, b):
        e.get(capi.BUF_DONE, copy=False)[:] = 1
        e.auto_reset(3, 0, mp.list_first[0], mp.list_count[0])
    sa, sb = a.get(capi.BUF_STATE), b.get(capi.BUF_STATE)
    assert np.array_equal(sa, sb)
    pth = a.get(capi.BUF_PATH)
    assert (pth[..., 0] >= 0).all() and (pth[..., 0] < 40).all() and (pth[..., 3] >= 3).all()
    pos = sa[..., 0:2]
    for bb in range(64):
        assert np.array_equal(pos[bb], mp.center[pth[bb, :, 0], pth[bb, :, 3]])
    d = np.sqrt(((pos[:, :, None] - pos[:, None]) ** 2).sum(-1)) + np.eye(16)[None] * 10
    assert d.min() >= 1.5 * np.sqrt(0.22 ** 2 + 0.107 ** 2) - 1e-6  # reset_agent_min_distance, road_traffic.py:679-684
    assert (sa[..., 3] >= 0).all() and (sa[..., 3] < 1).all() and (a.get(capi.BUF_TIMER)[:, 0] == 0).all()
    assert not a.get(capi.BUF_DONE).any() and np.array_equal(a.get(capi.BUF_PREV_POS), pos)
    a.close(); b.close()


def test_shard_ranges_and_slab_roundtrip():
    import torch
    from sigmarl_amd.shard import pack_slab, shard_range, unpack_slab

    for total, world in ((32768, 8), (4097, 3), (5, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    obs, rew, done = torch.randn(6, 4, 32), torch.randn(6, 4), (torch.rand(6) > 0.5).to(torch.uint8)
    o2, r2, d2 = unpack_slab(pack_slab(obs, rew, done), 4, 32)
    assert torch.equal(o2, obs) and torch.equal(r2, rew) and torch.equal(d2, done.bool())


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_binding as ob
from sigmarl_amd import capi
from sigmarl_amd.maps import load_map
from sigmarl_amd.params import Parameters, make_config
from sigmarl_amd.shard import RolloutExchange, pack_slab, shard_range, unpack_slab
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
TOTAL, N, T = 24, 4, 3
mp = load_map("cpm_entire")
p = Parameters(n_agents=N, scenario_type="cpm_entire", is_apply_mask=False, is_use_mtv_distance=False)
b0, b1 = shard_range(TOTAL, rank, world)
# every rank steps ONLY its env shard (no data-path collective); the full batch on rank 0 is the cross-check
def make(lo, hi):
    e = ob.OracleEnv(make_config(p, mp, hi - lo, env_index_base=lo), mp)  # (is_obs_noise defaults to True: the noise draws follow the env's index in the whole batch)
    ids = np.zeros((hi - lo, N, 4), np.int32); st = np.zeros((hi - lo, N, 8), np.float32)
    for b in range(lo, hi):
        for i in range(N):
            gp, k = (7 * b + 3 * i) % 40, 5 + (11 * b + 9 * i) % 50
            ids[b - lo, i] = (gp, 0, gp, k); st[b - lo, i, 0:2] = mp.center[gp, k]; st[b - lo, i, 2] = mp.yaw[gp, k]
    e.reset(np.repeat(np.arange(hi - lo), N), np.tile(np.arange(N), hi - lo), ids.reshape(-1, 4), st.reshape(-1, 8), 1); e.observe()
    return e
env = make(b0, b1)
full = make(0, TOTAL) if rank == 0 else None
CH = 2  # steps per chunk: T = 3 steps -> one full chunk + one partial chunk flushed at the end
ex = RolloutExchange(b1 - b0, N, env.D, CH, "cpu", dst=0, mode="gather")
rng = np.random.default_rng(0)
ref = []
for t in range(T):
    act = np.stack([rng.uniform(0, 1, (TOTAL, N)), rng.uniform(-0.2, 0.2, (TOTAL, N))], -1).astype(np.float32)
    env.step(act[b0:b1])
    # (on the GPU the step kernel writes this row block itself: sigmaenv_set_slab)
    pack_slab(torch.from_numpy(env.get(capi.BUF_OBS)), torch.from_numpy(env.get(capi.BUF_REWARD)), torch.from_numpy(env.get(capi.BUF_DONE)), out=ex.slot())
    ex.advance()
    if rank == 0:
        full.step(act)
        ref.append((full.get(capi.BUF_OBS), full.get(capi.BUF_REWARD), full.get(capi.BUF_DONE).astype(bool)))
ex.flush(); ex.wait_all()
if rank == 0:
    assert ex.completed == [(0, CH), (1, T - CH)]  # (buffer, valid steps): the final partial chunk says how many of its rows are new
    t = 0
    for k, n_steps in ex.completed:
        chunk = torch.cat(ex.gathered(k), 1)  # [CH, TOTAL, W]: ranks own contiguous env ranges
        obs, rew, done = unpack_slab(chunk, N, env.D)
        for q in range(n_steps):
            assert np.array_equal(obs[q].numpy(), ref[t][0]) and np.array_equal(rew[q].numpy(), ref[t][1]) and np.array_equal(done[q].numpy(), ref[t][2])
            t += 1
    assert t == T
# the same rollout through the all-to-all exchange: rank r ends up with steps [r T/W, (r+1) T/W) of the chunk for ALL envs
env2 = make(b0, b1)
full2 = make(0, TOTAL)
ex2 = RolloutExchange(b1 - b0, N, env2.D, 2, "cpu")  # (the default mode)
assert ex2.mode == "alltoall"
rng = np.random.default_rng(0)
ref2 = []
for t in range(2):
    act = np.stack([rng.uniform(0, 1, (TOTAL, N)), rng.uniform(-0.2, 0.2, (TOTAL, N))], -1).astype(np.float32)
    env2.step(act[b0:b1])
    pack_slab(torch.from_numpy(env2.get(capi.BUF_OBS)), torch.from_numpy(env2.get(capi.BUF_REWARD)), torch.from_numpy(env2.get(capi.BUF_DONE)), out=ex2.slot())
    ex2.advance()
    full2.step(act)
    ref2.append((full2.get(capi.BUF_OBS), full2.get(capi.BUF_REWARD), full2.get(capi.BUF_DONE).astype(bool)))
ex2.wait_all()
mine = ex2.time_slice(ex2.completed[0][0])          # [1, TOTAL, W]: step `rank` of the chunk, every env of both ranks
obs, rew, done = unpack_slab(mine, N, env2.D)
assert tuple(obs.shape[:2]) == (1, TOTAL)
assert np.array_equal(obs[0].numpy(), ref2[rank][0]) and np.array_equal(rew[0].numpy(), ref2[rank][1]) and np.array_equal(done[0].numpy(), ref2[rank][2])
# a chunk of 5 steps over 2 ranks: uneven time slices (2 + 3 steps), filled as ONE n-step launch fills it (chunk / commit), and device-style
# resets keyed on the env's index in the WHOLE batch (env_index_base): the sharded run draws what the unsharded one draws
def make_random(lo, hi):
    e = ob.OracleEnv(make_config(p, mp, hi - lo, env_index_base=lo), mp)
    e.get(capi.BUF_DONE, copy=False)[:] = 1
    e.auto_reset(9, 0, mp.list_first[0], mp.list_count[0])
    return e
env3, full3 = make_random(b0, b1), make_random(0, TOTAL)
assert np.array_equal(env3.get(capi.BUF_STATE), full3.get(capi.BUF_STATE)[b0:b1]) and np.array_equal(env3.get(capi.BUF_PATH), full3.get(capi.BUF_PATH)[b0:b1])
T5 = 5
ex3 = RolloutExchange(b1 - b0, N, env3.D, T5, "cpu", mode="alltoall")
assert ex3.slices == [0, 2, 5]
rng = np.random.default_rng(1)
buf = ex3.chunk()
ref3 = []
for t in range(T5):
    act = np.stack([rng.uniform(-0.2, 1.3, (TOTAL, N)), rng.uniform(-0.7, 0.7, (TOTAL, N))], -1).astype(np.float32)
    env3.step(act[b0:b1]); full3.step(act)
    pack_slab(torch.from_numpy(env3.get(capi.BUF_OBS)), torch.from_numpy(env3.get(capi.BUF_REWARD)), torch.from_numpy(env3.get(capi.BUF_DONE)), out=buf[t])
    ref3.append((full3.get(capi.BUF_OBS), full3.get(capi.BUF_REWARD), full3.get(capi.BUF_DONE).astype(bool)))
    env3.auto_reset(9, 1 + t, mp.list_first[0], mp.list_count[0]); full3.auto_reset(9, 1 + t, mp.list_first[0], mp.list_count[0])
    for w in (capi.BUF_STATE, capi.BUF_PATH, capi.BUF_OBS, capi.BUF_TIMER):
        assert np.array_equal(env3.get(w), full3.get(w)[b0:b1]), ("sharded reset differs from the unsharded one", t, w)
assert sum(int(r[2].sum()) for r in ref3) > 0
ex3.commit()
ex3.wait_all()
mine = ex3.time_slice(ex3.completed[0][0])
lo, hi = ex3.slices[rank], ex3.slices[rank + 1]
obs, rew, done = unpack_slab(mine, N, env3.D)
assert tuple(obs.shape[:2]) == (hi - lo, TOTAL)
for q in range(lo, hi):
    assert np.array_equal(obs[q - lo].numpy(), ref3[q][0]) and np.array_equal(rew[q - lo].numpy(), ref3[q][1]) and np.array_equal(done[q - lo].numpy(), ref3[q][2])
try:  # shards of different size cannot share one exchange (ADVICE r1): loud error, no hang
    RolloutExchange(3 + rank, N, env.D, 2, "cpu", dst=0, mode="gather")
    raise SystemExit("unequal shards were accepted")
except ValueError:
    pass
dist.barrier()
if rank == 0: print("SHARD_OK")
dist.destroy_process_group()
'''


def test_two_rank_sharding_gloo(tmp_path):
    """world_size 2 on CPU (gloo): each rank steps its env shard, the rollout slab is gathered to rank 0 and equals the unsharded run."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "SHARD_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bench_dry_run_eight_ranks_gloo():
    """`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8 --steps 20 --warmup 5 --dry-run`: the driver's 8-GPU command line with everything but
    the GPU work -- env-variable handling, rendezvous on 127.0.0.1, env ranges, the chunk exchange with uneven time slices (20 steps over 8 ranks),
    barrier + MAX reduction, and exactly ONE JSON line on stdout (from rank 0)."""
    import json

    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29523",
                          os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-run"], env=env, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    c = d["config"]
    assert c["envs_total"] == 32768 and c["steps_per_launch"] == 20 and c["exchange"] == "alltoall" and c["time_slices"] == [0, 2, 5, 7, 10, 12, 15, 17, 20]


def test_bench_gpus_without_a_launcher_relaunches_itself_and_a_mismatch_is_an_error():
    """`python bench.py --gpus 2` with no rendezvous in the environment must not time ONE rank and print `n_gpus: 1`: it re-executes
    itself under torch.distributed.run with 2 ranks (checked through --dry-run: gloo, no GPU).  Under a launcher whose WORLD_SIZE differs from --gpus it
    exits non-zero without a JSON line."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"], env=env, capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["envs_total"] == 2 * rec["config"]["envs_per_gpu"] and rec["config"]["exchange"] == "alltoall"
    assert "re-launching" in out.stderr
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.strip().startswith("{")], bad.stdout[-1000:]
    assert "torch.distributed.run" in bad.stderr


def test_bench_device_conditioning_is_a_rank_independent_step_count():
    """bench.py --condition-ms: the conditioning before the W warm-up steps is a STEP COUNT derived from the MAX-reduced cold time (the same on every rank: the ranks
    must issue the same number of chunk exchanges), whole launches of T steps, at least the requested milliseconds, with a host synchronisation every 8 launches."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeRun:
        def __init__(self):
            self.calls, self.finished = [], 0

        def run_steps(self, t0, n):
            self.calls.append((t0, n))

        def finish_chunk(self):
            self.finished += 1

    class FakeTorch:
        class cuda:
            syncs = 0

            @staticmethod
            def synchronize():
                FakeTorch.cuda.syncs += 1

    for ms, step_s, per in [(200.0, 29e-6, 20), (200.0, 29e-6, 32), (50.0, 0.4e-3, 1), (1.0, 1.0, 4)]:
        r = FakeRun()
        FakeTorch.cuda.syncs = 0
        n = bench.condition_device(r, FakeTorch, ms, step_s, per)
        assert n % per == 0 and n * step_s >= ms * 1e-3 and (n - per) * step_s < ms * 1e-3
        assert sum(k for _, k in r.calls) == n and all(k <= 8 * per for _, k in r.calls)
        assert [t0 for t0, _ in r.calls] == [8 * per * i for i in range(len(r.calls))]  # consecutive step indices: the action windows of the timed runs
        assert r.finished == len(r.calls) == FakeTorch.cuda.syncs
    assert bench.condition_device(FakeRun(), FakeTorch, 0.0, 29e-6, 32) == 0


@pytest.mark.parametrize("tag,testing", [("train", False), ("test", True)])
def test_device_sampler_specification_matches_the_reference_distribution(tag, testing):
    """The counter-based reset sampler (oracle side of the specification the HIP kernels share) draws (path, point, speed) with the marginals of
    the reference's torch-RNG rejection sampler (world_state_rt_sim.py:215-311), in training and in testing mode (range growing with the tries),
    and never violates the minimum spacing: two-sample chi-square tests against tests/golden/reset_distribution.npz."""
    import oracle_binding as ob
    import reset_distribution_check as rdc
    from sigmarl_amd.maps import load_map
    from sigmarl_amd.params import Parameters, make_config

    mp = load_map("cpm_entire")
    p = Parameters(n_agents=16, scenario_type="cpm_entire", is_apply_mask=False, is_obs_noise=False, is_testing_mode=testing)
    env = ob.OracleEnv(make_config(p, mp, 512), mp)
    got = rdc.sample_histograms(env, mp, rounds=4)
    env.close()
    rdc.compare(tag, got)


def test_mixed_scenario_lists_match_the_reference_distribution():
    """cpm_mixed (world_state_rt_sim.py:313-358): the device-side reset specification (oracle side) draws every env's sub-scenario with
    cpm_scenario_probabilities and its agents' paths from that sub-scenario's list -- chi-square against the reference's own draws
    (tests/golden/reset_distribution.npz, mixed_*) and against the stated probabilities; a per-agent reset keeps the env's sub-scenario."""
    import oracle_binding as ob
    import reset_distribution_check as rdc
    from sigmarl_amd import capi
    from sigmarl_amd.maps import load_map
    from sigmarl_amd.params import Parameters, make_config

    mp = load_map("cpm_mixed")
    z = np.load(rdc.FIXTURE)
    probs = [float(v) for v in z["mixed_probabilities"]]
    assert [int(v) for v in z["mixed_list_counts"]] == [mp.list_count[k] for k in (1, 2, 3)]
    p = Parameters(n_agents=2, scenario_type="cpm_mixed", is_apply_mask=False, is_obs_noise=False, cpm_scenario_probabilities=probs)
    env = ob.OracleEnv(make_config(p, mp, 1024), mp)
    got = rdc.sample_mixed(env, mp, rounds=4, probabilities=probs)
    rdc.compare_mixed(got)
    # per-agent requests in unfinished envs: the re-placed agent stays on its env's list (:325-328)
    env.get(capi.BUF_DONE, copy=False)[:] = 0
    fl = env.get(capi.BUF_COL_FLAGS, copy=False)
    fl[..., 3] = 0
    fl[::2, 1, 3] = 1
    before = env.get(capi.BUF_PATH).copy()
    env.auto_reset(5, 100, 0, capi.SCENARIO_LISTS)
    after = env.get(capi.BUF_PATH)
    assert np.array_equal(after[..., 1], before[..., 1])                           # sub-scenario ids untouched
    first = np.asarray([mp.list_first[k] for k in range(4)])[after[..., 1]]
    count = np.asarray([mp.list_count[k] for k in range(4)])[after[..., 1]]
    assert ((after[..., 0] >= first) & (after[..., 0] < first + count)).all() and np.array_equal(after[..., 2], after[..., 0] - first)
    moved = (after[..., [0, 3]] != before[..., [0, 3]]).any(-1)
    assert moved[::2, 1].mean() > 0.9 and not moved[1::2].any() and not moved[::2, 0].any()
    # without lists the scenario-list mode is refused
    plain = ob.OracleEnv(make_config(Parameters(n_agents=4, scenario_type="cpm_entire", is_obs_noise=False), load_map("cpm_entire"), 4), load_map("cpm_entire"))
    assert plain.lib.auto_reset(plain.h, 0, 0, 0, capi.SCENARIO_LISTS) != 0
    plain.close()
    env.close()


def test_observation_noise_specification_matches_the_reference_scaling():
    """`obs + obs_noise_level * rand_like(obs)` (observation_provider_rt.py:613-618): uniform in [0, level) on every element, fresh at every step.  The
    oracle side of the shared specification (the HIP kernels draw the same numbers: tests/test_gpu_nstep.py): range, mean level / 2, variance
    level^2 / 12, no element repeated from one step to the next, a function of (random_seed, env index in the batch) only."""
    import numpy as np
    import oracle_binding as ob
    from sigmarl_amd import capi
    from sigmarl_amd.maps import load_map
    from sigmarl_amd.params import Parameters, make_config

    mp = load_map("cpm_entire")
    N, B, level = 8, 96, 0.05
    kw = dict(n_agents=N, scenario_type="cpm_entire", is_apply_mask=False, is_use_mtv_distance=False, max_steps=9)
    pf, pc = mp.list_first[0], mp.list_count[0]

    def run(noise, seed=0, base=0, lo=0, hi=B):
        p = Parameters(is_obs_noise=noise, obs_noise_level=level, random_seed=seed, **kw)
        e = ob.OracleEnv(make_config(p, mp, hi - lo, env_index_base=base + lo), mp)
        e.get(capi.BUF_DONE, copy=False)[:] = 1
        e.auto_reset(3, 0, pf, pc)
        rng = np.random.default_rng(0)
        out = [e.get(capi.BUF_OBS).copy()]
        for t in range(6):
            act = np.stack([rng.uniform(0, 1, (B, N)), rng.uniform(-0.25, 0.25, (B, N))], -1).astype(np.float32)
            e.step(act[lo:hi])
            out.append(e.get(capi.BUF_OBS).copy())
            e.auto_reset(3, t + 1, pf, pc)
        e.close()
        return np.stack(out)

    clean, noisy = run(False), run(True)
    assert make_config(Parameters(is_obs_noise=False, **kw), mp, 4).obs_noise_level == 0.0
    d = (noisy.astype(np.float64) - clean).ravel()
    assert d.min() >= -1e-7 and d.max() < level + 1e-7                      # [0, level)
    assert abs(d.mean() - level / 2) < 2e-4 and abs(d.var() - level ** 2 / 12) < 2e-5
    h, _ = np.histogram(d, bins=10, range=(0, level))
    assert h.min() > 0.9 * d.size / 10 and h.max() < 1.1 * d.size / 10      # flat
    n = noisy.astype(np.float64) - clean
    assert abs(np.corrcoef(n[2].ravel(), n[3].ravel())[0, 1]) < 0.02        # fresh numbers every step
    assert np.array_equal(noisy, run(True)) and not np.array_equal(noisy, run(True, seed=7))
    # the draws follow the env's index in the WHOLE batch: a shard sees what the unsharded batch sees
    assert np.array_equal(run(True, lo=32, hi=64), noisy[:, 32:64])


def test_mirror_classes_derive_from_the_vmas_bases():
    """the reference's WorldCustom / Vehicle / VehicleState are vmas subclasses (helper_training.py:791, helper_common.py:382, :290); with vmas importable the
    mirror's classes derive from the same bases -- ``isinstance`` holds --, the bases' allocating constructors are never called, no read-only base property (name,
    u_range, x_semidim, batch_dim ...) keeps the mirror from holding its own value, and the state accessors are still views of the library's buffer.  vmas itself is
    absent from the image: tests/fake_vmas.py provides the published class skeleton (its constructors and tensor-touching methods raise when reached)."""
    import importlib
    from types import SimpleNamespace

    import torch

    import fake_vmas
    import sigmarl_amd.scenario as sc

    try:
        with fake_vmas.installed():
            sc = importlib.reload(sc)
            core = sys.modules["vmas.simulator.core"]
            assert sc.BaseScenario is sys.modules["vmas.simulator.scenario"].BaseScenario
            B, N = 5, 3
            state = torch.arange(B * N * 8, dtype=torch.float32).reshape(B, N, 8)  # stands for SIGMAENV_BUF_STATE
            env = SimpleNamespace(B=B, N=N, device=torch.device("cpu"))
            world = sc.WorldCustom(env, 0.05, torch.tensor(4.5), torch.tensor(4.0))
            for i in range(N):
                world.add_agent(sc.Vehicle(name=f"agent_{i}", state_row=state[:, i], shape=sc.Box(0.16, 0.08), u_range=[1.0, 0.5], max_speed=1.0,
                                           dynamics=sc.KinematicBicycleModel(device="cpu")))
            assert isinstance(world, core.World) and isinstance(world, core.TorchVectorizedObject)
            a = world.agents[1]
            assert isinstance(a, core.Agent) and isinstance(a, core.Entity) and isinstance(a.state, core.AgentState) and isinstance(a.state, core.EntityState)
            # nothing of the bases shadows what the mirror keeps: names, ranges, dims, and the views themselves
            assert a.name == "agent_1" and a.u_range == [1.0, 0.5] and a.max_speed == 1.0 and a.silent is True and a.action_size == 2 and a.action_script is None
            assert a.batch_dim == B and a.state.batch_dim == B and world.batch_dim == B and world.dim_c == 0 and world.dim_p == 2 and float(world.x_semidim) == 4.5
            assert world.agents is world.policy_agents and world.entities == world.agents and world.scripted_agents == []
            assert a.state.pos.data_ptr() == state[:, 1, 0:2].data_ptr() and a.state.rot.shape == (B, 1) and a.state.vel.shape == (B, 2)
            a.set_pos(torch.tensor([7.0, 8.0]), batch_index=2)  # the mirror's setter (a write into the buffer), not Entity.set_pos
            assert state[2, 1, 0] == 7.0 and state[2, 1, 1] == 8.0
            a.set_rot(torch.full((B, 1), 0.25))
            assert torch.all(state[:, 1, 2] == 0.25)
            assert a.state.c is None and a.state.force is None  # the bases' own accessors find their (empty) backing fields
            world.reset(None); world.zero_grad(); world.to(torch.device("cpu"))  # the mirror's no-ops, not World's
            assert a.action.u is None and a.action.u_range_tensor.shape == (2,)
    finally:
        sc = importlib.reload(sc)  # back to the image's state (no vmas)
    assert sc.WorldCustom.__mro__[1] is object


_PARTIAL_CHUNK_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
from sigmarl_amd.shard import RolloutExchange, slab_width

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B, N, D, T = 3, 2, 5, 7            # 7 steps over 3 / 8 ranks: uneven (and, for 8 ranks, EMPTY) time slices
W = slab_width(N, D)
def cell(r, t, b):                 # what rank r records for step t of its env b: a value that names all three (and the chunk it belongs to, below)
    return float(1000 * r + 10 * t + b)
ex = RolloutExchange(B, N, D, T, "cpu", mode="alltoall")
assert ex.slices == [(r * T) // world for r in range(world + 1)]
for chunk_no, n_steps in enumerate([T, 4]):   # a full chunk through chunk() / commit(), then a LAST chunk of which only 4 rows are new: commit(4)
    buf = ex.chunk()
    buf.fill_(-1.0)                           # (rows beyond n_steps are stale by contract: they still travel, valid_steps says so)
    for t in range(n_steps):
        for b in range(B):
            buf[t, b, :] = cell(rank, t, b) + 100000.0 * chunk_no
    ex.commit(n_steps)
ex.wait_all()
assert ex.completed == [(0, T), (1, 4)] and ex.valid_steps == [T, 4], (ex.completed, ex.valid_steps)
lo, hi = ex.slices[rank], ex.slices[rank + 1]
for k, n_steps in ex.completed:
    mine = ex.time_slice(k)                   # [my steps, world * B, W]: steps [lo, hi) of the chunk for the envs of ALL ranks, in rank order
    assert tuple(mine.shape) == (hi - lo, world * B, W), mine.shape
    for t in range(lo, hi):
        for r in range(world):
            for b in range(B):
                want = cell(r, t, b) + 100000.0 * k if t < n_steps else -1.0    # (a stale row arrives as it was left; the consumer reads valid_steps)
                got = mine[t - lo, r * B + b]
                assert torch.all(got == want), (k, t, r, b, float(got[0]), want)
# what a learner rank may read of the last chunk: its slice cut at valid_steps
new_rows = max(0, min(hi, ex.valid_steps[1]) - lo)
assert new_rows == len([t for t in range(lo, hi) if t < 4])
dist.barrier()
if rank == 0: print("PARTIAL_OK", world)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,port", [(3, 29531), (8, 29533)])
def test_last_partial_chunk_ships_valid_steps_under_alltoall_with_uneven_slices(tmp_path, world, port):
    """``commit(n_steps < T)`` on the last chunk under the all-to-all exchange with uneven time slices, for world sizes 3 and 8 (gloo; 7 steps over
    8 ranks leaves one rank an EMPTY slice): ``completed`` / ``valid_steps`` carry the number of new rows, every rank receives exactly its slice of every rank's chunk,
    stale rows arrive as they were left."""
    script = tmp_path / "worker.py"
    script.write_text(_PARTIAL_CHUNK_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(script), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and f"PARTIAL_OK {world}" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
