"""The on-device networks: the exact-fp32 actor / critic (matrix cores, fp32 fma chains) against torch.nn in fp32; the bf16 inference variant against
torch.nn and a bf16-emulating restatement; the distribution head; rollouts without the host."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bf16(a):
    """round-to-nearest-even fp32 -> bf16 -> fp32 (numpy)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _emulated(mlp, obs):
    """What the kernel computes: bf16 inputs / weights / hidden activations, fp32 accumulation (the summation order inside the matrix
    cores differs, hence the tolerance)."""
    import torch
    lin = [m for m in mlp.modules() if isinstance(m, torch.nn.Linear)]
    x = _bf16(obs)
    for k, m in enumerate(lin):
        w, b = _bf16(m.weight.detach().numpy()), m.bias.detach().numpy().astype(np.float32)
        x = (x.astype(np.float64) @ w.T.astype(np.float64) + b).astype(np.float32)
        if k < 3:
            x = _bf16(np.tanh(x))
    return x


def _setup(B=64, N=16, seed=0, precision="bf16", **pkw):
    import torch
    from sigmarl_amd.actor import Actor, make_mlp
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters

    torch.manual_seed(seed)
    env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, **pkw),
                   n_envs=B, device="cuda:0")
    env.reset_random(seed=3)
    mlp = make_mlp(env.D)
    with torch.no_grad():  # larger weights than the default init so that tanh saturates somewhere and the last layer matters
        for m in mlp:
            if isinstance(m, torch.nn.Linear):
                m.weight.mul_(1.7)
                m.bias.uniform_(-0.3, 0.3)
    actor = Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], precision=precision)
    return torch, env, mlp, actor


def test_actor_matches_torch_and_the_bf16_restatement():
    torch, env, mlp, actor = _setup(B=70, N=16)   # 1120 rows: not a multiple of the 256 rows of a workgroup
    R = env.B * env.N
    obs = (torch.rand((R, env.D), device="cuda") * 2 - 1) * 1.5
    act = torch.zeros((env.B, env.N, 2), device="cuda")
    ls = torch.zeros((env.B, env.N, 4), device="cuda")
    lp = torch.zeros((env.B, env.N), device="cuda")
    actor.forward(env, act, lp, ls, obs=obs, deterministic=True)
    env.sync()
    ls_h = ls.reshape(R, 4).cpu().numpy()
    out_emul = _emulated(mlp, obs.cpu().numpy())
    with torch.no_grad():
        out_f32 = mlp(obs.cpu()).numpy()
    bias = np.log(np.expm1(0.99))  # torchrl's biased_softplus(1.0): softplus(x + inv_softplus(1.0 - 0.01)) + 0.01, clamped at 1e-4
    sp = lambda v: np.maximum(np.log1p(np.exp(v + bias)) + 0.01, 1e-4)  # noqa: E731
    # loc and scale against the bf16-emulating restatement (tolerance: fp32 accumulation order + an occasional flipped bf16 rounding)
    assert np.abs(ls_h[:, :2] - out_emul[:, :2]).max() <= 2e-2
    assert np.abs(ls_h[:, 2:] - sp(out_emul[:, 2:])).max() <= 2e-2
    assert np.abs(ls_h[:, :2] - out_emul[:, :2]).mean() <= 2e-3
    # ... and against the fp32 network (bf16 inference error)
    assert np.abs(ls_h[:, :2] - out_f32[:, :2]).max() <= 0.15 and np.abs(ls_h[:, :2] - out_f32[:, :2]).mean() <= 2e-2
    # deterministic action = squash(loc) between low and high
    a = act.reshape(R, 2).cpu().numpy()
    exp = np.tanh(ls_h[:, :2]) * np.array([1.0, 0.6], np.float32)
    assert np.abs(a - exp).max() <= 1e-5
    env.close()
    actor.close()


@pytest.mark.parametrize("mode", ["split", "exact"])
@pytest.mark.parametrize("scaled", [False, True])
def test_fp32_actor_matches_torch_nn(scaled, mode):
    """The fp32 actor (sigmaenv_actor_forward_f32) in both arithmetic modes -- "exact": v_mfma_f32_32x32x2_f32, an fp32 fma chain; "split" (the default): every
    operand as hi + lo fp16, three exact-product v_mfma_f32_32x32x16_f16 per fp32 product -- == torch.nn in fp32 within 1e-5 on the default initialisation
    (and within 1e-4 relative with 1.7 x larger weights, where the pre-activations are ~10): loc, scale and the deterministic action."""
    import torch
    from sigmarl_amd.actor import Actor, make_mlp
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters

    torch.manual_seed(4)
    env = SigmaEnv(Parameters(n_agents=16, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=70, device="cuda:0")
    env.reset_random(seed=3)
    mlp = make_mlp(env.D)
    if scaled:
        with torch.no_grad():
            for m in mlp:
                if isinstance(m, torch.nn.Linear):
                    m.weight.mul_(1.7)
                    m.bias.uniform_(-0.3, 0.3)
    actor = Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], mode=mode)  # precision "fp32" is the default
    assert actor._mlp32.set_mode(mode) == mode
    R = env.B * env.N
    obs = (torch.rand((R, env.D), device="cuda") * 2 - 1) * 1.5
    obs[:64, :4] *= 1e-4  # (values whose fp16 hi / lo parts are subnormal)
    obs[64:128, 4:8] *= 100.0
    act = torch.zeros((env.B, env.N, 2), device="cuda")
    ls = torch.zeros((env.B, env.N, 4), device="cuda")
    actor.forward(env, act, None, ls, obs=obs, deterministic=True)
    env.sync()
    with torch.no_grad():
        out = mlp(obs.cpu()).numpy()
    ls_h = ls.reshape(R, 4).cpu().numpy()
    tol = 1e-5 if not scaled else 1e-4
    assert np.abs(ls_h[:, :2] - out[:, :2]).max() <= tol * max(1.0, np.abs(out[:, :2]).max())
    sp = np.maximum(np.log1p(np.exp(out[:, 2:] + np.log(np.expm1(0.99)))) + 0.01, 1e-4)
    assert np.abs(ls_h[:, 2:] - sp).max() <= tol * max(1.0, sp.max())
    a = act.reshape(R, 2).cpu().numpy()
    assert np.abs(a - np.tanh(out[:, :2]).clip(-1 + 1e-6, 1 - 1e-6) * np.array([1.0, 0.6], np.float32)).max() <= 2e-5
    # the environment's own observation buffer as input (obs=None): same network on env.obs
    actor.forward(env, act, None, ls, deterministic=True)
    env.sync()
    with torch.no_grad():
        out2 = mlp(env.obs.reshape(R, env.D).cpu()).numpy()
    assert np.abs(ls.reshape(R, 4).cpu().numpy()[:, :2] - out2[:, :2]).max() <= tol * max(1.0, np.abs(out2[:, :2]).max())
    env.close()
    actor.close()


@pytest.mark.parametrize("mode", ["split", "exact"])
def test_fp32_critic_matches_torch_nn(mode):
    """The MAPPO critic (optimization_module.py:16-32: centralised, shared parameters, N D -> 256 -> 256 -> 256 -> 1, Tanh) in fp32 (both arithmetic
    modes) == torch.nn in fp32 within 1e-5 on the default initialisation; one value per env, handed to every agent."""
    import torch
    from sigmarl_amd.actor import Critic
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters

    torch.manual_seed(6)
    N = 16
    env = SigmaEnv(Parameters(n_agents=N, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False), n_envs=77, device="cuda:0")
    env.reset_random(seed=5)
    net = torch.nn.Sequential(torch.nn.Linear(N * env.D, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(), torch.nn.Linear(256, 256), torch.nn.Tanh(),
                              torch.nn.Linear(256, 1))
    critic = Critic(net, mode=mode)
    v = critic.values(env)
    env.sync()
    assert tuple(v.shape) == (env.B, N, 1)
    with torch.no_grad():
        want = net(env.obs.reshape(env.B, N * env.D).cpu()).numpy()
    assert np.abs(v[:, 0, 0].cpu().numpy() - want[:, 0]).max() <= 1e-5
    assert torch.equal(v[:, 0], v[:, N - 1])
    env.close()
    critic.close()


def test_sampling_statistics_and_log_prob():
    torch, env, mlp, actor = _setup(B=256, N=16)
    R = env.B * env.N
    obs = torch.zeros((R, env.D), device="cuda")  # same input everywhere: every row samples from the same distribution
    act = torch.zeros((env.B, env.N, 2), device="cuda")
    ls = torch.zeros((env.B, env.N, 4), device="cuda")
    lp = torch.zeros((env.B, env.N), device="cuda")
    actor.forward(env, act, lp, ls, obs=obs, seed=5, counter=1)
    act2 = torch.zeros_like(act)
    actor.forward(env, act2, None, None, obs=obs, seed=5, counter=1)
    act3 = torch.zeros_like(act)
    actor.forward(env, act3, None, None, obs=obs, seed=5, counter=2)
    env.sync()
    assert torch.equal(act, act2) and not torch.equal(act, act3)       # counter-based: reproducible, and fresh per counter
    loc, sc = ls[0, 0, :2].cpu().numpy(), ls[0, 0, 2:].cpu().numpy()
    half = np.array([1.0, 0.6], np.float32)
    y = (act.reshape(R, 2).cpu().numpy() / half).clip(-0.999999, 0.999999)
    x = np.arctanh(y)
    z = (x - loc) / sc                                                    # recovered standard normals
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1.0) < 0.05 and abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.05
    jac = 2.0 * (np.log(2.0) - x - np.logaddexp(0.0, -2.0 * x))          # log |d tanh / dx| in the stable form torch's TanhTransform uses
    ref_lp = (-0.5 * z * z - np.log(sc) - 0.5 * np.log(2 * np.pi) - jac - np.log(half)).sum(1)
    assert np.abs(lp.reshape(R).cpu().numpy() - ref_lp).max() <= 5e-3
    assert abs(float(sc[0]) - 1.0) < 0.5 and float(sc.min()) >= 0.01   # scale(raw) = softplus(raw + b) + 0.01: never below the 0.01 floor
    env.close()
    actor.close()


def test_rollout_without_the_host_equals_stepwise_calls():
    """sigmaenv_rollout == the same T steps issued one by one (policy, then the fused step / record / reset)."""
    from sigmarl_amd.shard import slab_width
    torch, env, mlp, actor = _setup(B=96, N=16, seed=1)
    torch2, env2, _, _ = _setup(B=96, N=16, seed=1)
    actor2 = __import__("sigmarl_amd.actor", fromlist=["Actor"]).Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], precision="bf16")
    T, W = 6, slab_width(env.N, env.D)
    slab = torch.zeros((T, env.B, W), device="cuda")
    lp = torch.zeros((T, env.B, env.N), device="cuda")
    acts = torch.zeros((T, env.B, env.N, 2), device="cuda")
    actor.rollout(env, T, slab=slab, log_prob=lp, actions=acts, seed=9, counter0=100)
    env.sync()
    a = torch.zeros((env2.B, env2.N, 2), device="cuda")
    slab2 = torch.zeros_like(slab)
    for t in range(T):
        actor2.forward(env2, a, seed=9, counter=100 + t)
        env2.set_slab(slab2[t])
        env2.step_autoreset(a, seed=9, counter=100 + t)
        env2.sync()
        assert torch.equal(a, acts[t])
    assert torch.equal(slab, slab2)
    assert torch.equal(env.obs, env2.obs) and torch.equal(env.state, env2.state)
    assert torch.isfinite(lp).all() and float(slab[..., -1].sum()) >= 0
    for e in (env, env2):
        e.close()
    actor.close()
    actor2.close()


def test_rollout_with_cbf_margin_reward_equals_stepwise_calls():
    """rew_method "cbf": sigmaenv_rollout runs policy -> CBF margin launch -> fused step per step, like the calls issued one by one."""
    from sigmarl_amd import capi
    from sigmarl_amd.shard import slab_width
    kw = dict(rew_method="cbf_sparse", is_solve_qp=False, is_using_cbf_training=True)
    torch, env, mlp, actor = _setup(B=48, N=16, seed=2, **kw)
    _, env2, _, _ = _setup(B=48, N=16, seed=2, **kw)
    actor2 = __import__("sigmarl_amd.actor", fromlist=["Actor"]).Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], precision="bf16")
    for e in (env, env2):
        e.cbf_attach()
    T, W = 5, slab_width(env.N, env.D)
    slab, slab2 = torch.zeros((T, env.B, W), device="cuda"), torch.zeros((T, env.B, W), device="cuda")
    actor.rollout(env, T, slab=slab, seed=4, counter0=50)
    env.sync()
    a = torch.zeros((env2.B, env2.N, 2), device="cuda")
    seen = 0
    for t in range(T):
        actor2.forward(env2, a, seed=4, counter=50 + t)
        env2.cbf_rewards(a)
        env2.set_slab(slab2[t])
        env2.step_autoreset(a, seed=4, counter=50 + t)
        env2.sync()
        seen += int((env2.buffer(capi.BUF_REWARD_INFO)[4:7] != 0).sum())
    assert seen > 0  # the margin channels were non-trivial
    assert torch.equal(slab, slab2)
    assert torch.equal(env.buffer(capi.BUF_REWARD_INFO), env2.buffer(capi.BUF_REWARD_INFO)) and torch.equal(env.state, env2.state)
    for e in (env, env2):
        e.close()
    actor.close()
    actor2.close()


def test_rollout_with_cbf_qp_equals_stepwise_calls():
    """is_solve_qp=True with is_apply_cbf_action: sigmaenv_rollout runs policy -> centralized QP -> fused step on the SAFE action."""
    from sigmarl_amd import capi
    from sigmarl_amd.shard import slab_width
    kw = dict(rew_method="cbf", is_solve_qp=True, is_using_cbf_training=True, is_apply_cbf_action=True)
    torch, env, mlp, actor = _setup(B=24, N=8, seed=5, **kw)
    _, env2, _, _ = _setup(B=24, N=8, seed=5, **kw)
    actor2 = __import__("sigmarl_amd.actor", fromlist=["Actor"]).Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], precision="bf16")
    for e in (env, env2):
        e.cbf_attach()
    T, W = 4, slab_width(env.N, env.D)
    slab, slab2 = torch.zeros((T, env.B, W), device="cuda"), torch.zeros((T, env.B, W), device="cuda")
    actor.rollout(env, T, slab=slab, seed=4, counter0=70)
    env.sync()
    a = torch.zeros((env2.B, env2.N, 2), device="cuda")
    for t in range(T):
        actor2.forward(env2, a, seed=4, counter=70 + t)
        safe = env2.cbf_qp(a)
        env2.set_slab(slab2[t])
        env2.step_autoreset(safe, seed=4, counter=70 + t)
        env2.sync()
    assert torch.equal(env.state, env2.state) and torch.equal(slab, slab2)  # the QP launch is bitwise repeatable
    assert torch.equal(env.buffer(capi.BUF_TIMER), env2.buffer(capi.BUF_TIMER))
    for e in (env, env2):
        e.close()
    actor.close()
    actor2.close()


def test_fp32_rollout_equals_stepwise_fp32_calls_with_observation_noise():
    """sigmaenv_rollout_f32 (the reference's precision AND its default observation noise, both on the device) == the same T steps issued one by one:
    sigmaenv_actor_forward_f32 on the (noisy) observation buffer, then the fused step / record / reset -- bit for bit."""
    from sigmarl_amd import capi
    from sigmarl_amd.actor import Actor
    from sigmarl_amd.shard import slab_width

    kw = dict(is_obs_noise=True, obs_noise_level=0.05, random_seed=3)
    setups = []
    for _ in range(2):
        import torch
        from sigmarl_amd.actor import make_mlp
        from sigmarl_amd.env import SigmaEnv
        from sigmarl_amd.params import Parameters

        torch.manual_seed(1)
        env = SigmaEnv(Parameters(n_agents=16, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, **kw), n_envs=96, device="cuda:0")
        env.reset_random(seed=3)
        mlp = make_mlp(env.D)
        setups.append((env, Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6])))  # default precision: fp32
    (env, actor), (env2, actor2) = setups
    assert actor.precision == "fp32" and env.cfg.obs_noise_level > 0
    T, W = 6, slab_width(env.N, env.D)
    slab, slab2 = torch.zeros((T, env.B, W), device="cuda"), torch.zeros((T, env.B, W), device="cuda")
    lp, lp2 = torch.zeros((T, env.B, env.N), device="cuda"), torch.zeros((T, env.B, env.N), device="cuda")
    acts = torch.zeros((T, env.B, env.N, 2), device="cuda")
    actor.rollout(env, T, slab=slab, log_prob=lp, actions=acts, seed=9, counter0=100)
    env.sync()
    a = torch.zeros((env2.B, env2.N, 2), device="cuda")
    for t in range(T):
        actor2.forward(env2, a, lp2[t], seed=9, counter=100 + t)
        env2.set_slab(slab2[t])
        env2.step_autoreset(a, seed=9, counter=100 + t)
        env2.sync()
        assert torch.equal(a, acts[t])
    assert torch.equal(slab, slab2) and torch.equal(lp, lp2)
    for w in (capi.BUF_OBS, capi.BUF_STATE, capi.BUF_TIMER, capi.BUF_REWARD):
        assert torch.equal(env.buffer(w), env2.buffer(w))
    for e, ac in setups:
        e.close()
        ac.close()


def test_two_env_shards_roll_out_into_one_record_buffer():
    """sigmaenv_set_rollout_slab_stride: a batch split over two handles (shard k = envs [k Bs, (k + 1) Bs), `env_index_base`) rolls out on two streams into ONE
    [T, B, W] record -- shard k's rows at slab + k Bs W with the step stride B W -- and the record, the log-probabilities and the final buffers equal the
    unsharded handle's bit for bit (every draw is keyed on the env's index in the whole batch); a stride below the handle's own block is refused."""
    import torch
    from sigmarl_amd import capi
    from sigmarl_amd.actor import Actor, make_mlp
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters
    from sigmarl_amd.shard import slab_width

    B, Bs, T = 64, 32, 5
    kw = dict(n_agents=16, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=True, obs_noise_level=0.05, random_seed=3)
    torch.manual_seed(1)
    whole = SigmaEnv(Parameters(**kw), n_envs=B, device="cuda:0")
    whole.reset_random(seed=3)
    mlp = make_mlp(whole.D)
    W = slab_width(whole.N, whole.D)
    a_whole = Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6])
    slab = torch.zeros((T, B, W), device="cuda")
    lp = torch.zeros((T, B, whole.N), device="cuda")
    a_whole.rollout(whole, T, slab=slab, log_prob=lp, seed=9, counter0=100)
    whole.sync()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    shards, actors = [], []
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            e = SigmaEnv(Parameters(**kw), n_envs=Bs, device="cuda:0", env_index_base=k * Bs)
            e.reset_random(seed=3)
            shards.append(e)
            actors.append(Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6]))
    with pytest.raises(RuntimeError):
        shards[0].set_rollout_slab_stride(Bs * W - 1)
    slab2 = torch.zeros((T, B, W), device="cuda")
    lp2 = [torch.zeros((T, Bs, whole.N), device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for k, e in enumerate(shards):
        e.set_rollout_slab_stride(B * W)
        actors[k].rollout(e, T, slab_ptr=slab2.data_ptr() + k * Bs * W * 4, log_prob=lp2[k], seed=9, counter0=100)
    for e in shards:
        e.sync()
    assert torch.equal(slab, slab2)
    assert torch.equal(lp, torch.cat(lp2, dim=1))
    for w in (capi.BUF_OBS, capi.BUF_STATE, capi.BUF_TIMER, capi.BUF_REWARD):
        assert torch.equal(whole.buffer(w), torch.cat([e.buffer(w) for e in shards], dim=0))
    # ONE Actor driving both shards on their streams: its scratch buffers are per env handle, so the shards' launches do not overwrite each other's actions
    shared = Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6])
    for rep in range(3):  # (a race shows up as a mismatch in some repetition; fresh shards: the episode counters key the sensor noise)
        fresh = []
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                e = SigmaEnv(Parameters(**kw), n_envs=Bs, device="cuda:0", env_index_base=k * Bs)
                e.reset_random(seed=3)
                e.set_rollout_slab_stride(B * W)
                fresh.append(e)
        slab3 = torch.zeros((T, B, W), device="cuda")
        lp3 = [torch.zeros((T, Bs, whole.N), device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        for k, e in enumerate(fresh):
            shared.rollout(e, T, slab_ptr=slab3.data_ptr() + k * Bs * W * 4, log_prob=lp3[k], seed=9, counter0=100)
        for e in fresh:
            e.sync()
        assert torch.equal(slab, slab3), f"repetition {rep}"
        assert torch.equal(lp, torch.cat(lp3, dim=1))
        for e in fresh:
            e.close()
    assert len(shared._scratch_by_env) >= 2
    shards[0].set_rollout_slab_stride(0)  # back to the handle's own [T, Bs, W] layout
    own = torch.zeros((2, Bs, W), device="cuda")
    actors[0].rollout(shards[0], 2, slab=own, seed=9, counter0=200)
    shards[0].sync()
    assert own.abs().sum().item() > 0
    for e in shards + [whole]:
        e.close()
    for a in actors + [a_whole]:
        a.close()


def test_actor_rollout_on_a_short_term_build_variant_equals_stepwise_calls():
    """An env with n_points_short_term = 5 lives in libsigmaenv_ns5.so; Actor / Critic must drive it through THAT library (they bound the
    default NS = 3 build, whose step kernel then ran on buffers laid out for NS = 5).  rollout == actor forward + fused step, call by call, bit for bit,
    and the critic reads the NS = 5 observation width."""
    import torch
    from sigmarl_amd import capi
    from sigmarl_amd.actor import Actor, Critic, make_mlp
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters
    from sigmarl_amd.shard import slab_width

    setups = []
    for _ in range(2):
        torch.manual_seed(2)
        env = SigmaEnv(Parameters(n_agents=8, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, n_points_short_term=5),
                       n_envs=64, device="cuda:0")
        env.reset_random(seed=4)
        setups.append((env, Actor(make_mlp(env.D), low=[-1.0, -0.6], high=[1.0, 0.6])))
    (env, actor), (env2, actor2) = setups
    assert env.lib.n_short_term() == 5 and env.lib is not actor.lib and env.D == 4 + 2 * 5 + 11 * 2
    T, W = 5, slab_width(env.N, env.D)
    slab, slab2 = torch.zeros((T, env.B, W), device="cuda"), torch.zeros((T, env.B, W), device="cuda")
    lp, lp2 = torch.zeros((T, env.B, env.N), device="cuda"), torch.zeros((T, env.B, env.N), device="cuda")
    acts = torch.zeros((T, env.B, env.N, 2), device="cuda")
    actor.rollout(env, T, slab=slab, log_prob=lp, actions=acts, seed=5, counter0=40)
    env.sync()
    a = torch.zeros((env2.B, env2.N, 2), device="cuda")
    for t in range(T):
        actor2.forward(env2, a, lp2[t], seed=5, counter=40 + t)
        env2.set_slab(slab2[t])
        env2.step_autoreset(a, seed=5, counter=40 + t)
        env2.sync()
        assert torch.equal(a, acts[t])
    assert torch.equal(slab, slab2) and torch.equal(lp, lp2)
    for w in (capi.BUF_OBS, capi.BUF_STATE, capi.BUF_TIMER, capi.BUF_REWARD, capi.BUF_SHORT_TERM):
        assert torch.equal(env.buffer(w), env2.buffer(w))
    assert torch.isfinite(env.buffer(capi.BUF_STATE)).all() and float(slab.abs().max()) < 1e3
    torch.manual_seed(7)
    ref = make_mlp(env.N * env.D, n_out=1).cuda()
    v = Critic(ref).values(env)
    want = ref(env.obs.reshape(env.B, -1)).detach().reshape(env.B, 1, 1).expand(env.B, env.N, 1)
    assert float((v - want).abs().max()) <= 1e-5
    for e, ac in setups:
        e.close()
        ac.close()


def test_actor_for_an_observation_width_the_bf16_kernel_does_not_take():
    """is_obs_steering gives obs_dim 35: the exact-fp32 network takes any width, the bf16 kernel only 8 / 16 / 24 / 32 -- an Actor can be built and run
    in fp32, and asking it for bf16 raises a clear error instead of failing at construction."""
    import torch
    from sigmarl_amd.actor import Actor, make_mlp
    from sigmarl_amd.env import SigmaEnv
    from sigmarl_amd.params import Parameters

    env = SigmaEnv(Parameters(n_agents=8, scenario_type="cpm_entire", is_use_mtv_distance=False, is_apply_mask=False, is_obs_noise=False, is_obs_steering=True),
                   n_envs=32, device="cuda:0")
    env.reset_random(seed=1)
    assert env.D == 35
    torch.manual_seed(0)
    mlp = make_mlp(env.D)
    actor = Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6])
    act = torch.zeros((env.B, env.N, 2), device="cuda")
    ls = torch.zeros((env.B, env.N, 4), device="cuda")
    actor.forward(env, act, None, ls, deterministic=True)
    env.sync()
    want = mlp(env.obs.reshape(-1, env.D).cpu()).detach().numpy()
    assert np.abs(ls.reshape(-1, 4)[:, :2].cpu().numpy() - want[:, :2]).max() <= 1e-5
    with pytest.raises(ValueError, match="bf16"):
        actor.forward(env, act, precision="bf16")
    with pytest.raises(ValueError, match="bf16"):
        Actor(mlp, low=[-1.0, -0.6], high=[1.0, 0.6], precision="bf16")
    env.close()
    actor.close()
