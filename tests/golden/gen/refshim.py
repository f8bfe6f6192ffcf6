"""Import shims for running the SigmaRL reference in THIS container only.

Golden-vector tooling, not product code.  The reference (``/root/reference``)
imports vmas / torchrl / tensordict / torchdiffeq / termcolor / cvxpy, none of
which are installed here.  ``install()`` registers minimal stand-ins in
``sys.modules`` so that ``sigmarl.scenarios.road_traffic`` imports, resets and
steps.  The stand-ins replace THIRD-PARTY code only; every line of SigmaRL's
own arithmetic that the goldens pin is executed from ``/root/reference``.

What the stand-ins restate (published behaviour of the pinned versions,
``requirements.txt:5-9``; parity of these pieces is "unpinned"):

* ``torchdiffeq.odeint(f, y0, t, method="euler")``: fixed-grid explicit Euler,
  ``y[k+1] = y[k] + (t[k+1]-t[k]) * f(t[k], y[k])``, returns ``stack(y)``.
* ``vmas.simulator.core``: ``World`` / ``Agent`` / ``AgentState`` / ``Box``
  containers with the batch-dim checked state setters and ``_spawn``/``_reset``.
* ``vmas.simulator.utils.TorchUtils.where_from_index``.

Never imported by tests that run on the GPU box, by ``bench.py`` or by the
product package.
"""
from __future__ import annotations

import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


# --------------------------------------------------------------------------- #
# termcolor
# --------------------------------------------------------------------------- #
def _mk_termcolor():
    m = types.ModuleType("termcolor")
    m.colored = lambda s, *a, **k: s
    m.cprint = lambda *a, **k: None
    return m


# --------------------------------------------------------------------------- #
# torchdiffeq
# --------------------------------------------------------------------------- #
def _mk_torchdiffeq():
    m = types.ModuleType("torchdiffeq")

    def odeint(func, y0, t, rtol=None, atol=None, method=None, options=None):
        assert method == "euler", "shim implements fixed-grid Euler only"
        ys = [y0]
        y = y0
        for k in range(t.shape[0] - 1):
            dt = t[k + 1] - t[k]
            y = y + dt * func(t[k], y)
            ys.append(y)
        return torch.stack(ys, dim=0)

    m.odeint = odeint
    return m


# --------------------------------------------------------------------------- #
# vmas
# --------------------------------------------------------------------------- #
class TorchVectorizedObject:
    def __init__(self, batch_dim=None, device=None):
        self._batch_dim = batch_dim
        self._device = device

    @property
    def batch_dim(self):
        return self._batch_dim

    @batch_dim.setter
    def batch_dim(self, v):
        assert self._batch_dim is None, "You can set batch dim only once"
        self._batch_dim = v

    @property
    def device(self):
        return self._device

    @device.setter
    def device(self, v):
        self._device = v

    def _check_batch_index(self, batch_index):
        if batch_index is not None:
            assert 0 <= batch_index < self.batch_dim


class TorchUtils:
    @staticmethod
    def where_from_index(env_index, new_value, old_value):
        mask = torch.zeros_like(old_value, dtype=torch.bool, device=old_value.device)
        mask[env_index] = True
        return torch.where(mask, new_value, old_value)

    @staticmethod
    def recursive_clone(v):
        if isinstance(v, torch.Tensor):
            return v.clone()
        if isinstance(v, dict):
            return {k: TorchUtils.recursive_clone(x) for k, x in v.items()}
        return v


def override(_cls):
    def deco(f):
        return f

    return deco


def _state_prop(name):
    private = "_" + name

    def fget(self):
        return getattr(self, private)

    def fset(self, value):
        assert self._batch_dim is not None and self._device is not None
        assert value.shape[0] == self._batch_dim, (
            f"Internal state must match batch dim, got {value.shape[0]}, "
            f"expected {self._batch_dim}"
        )
        setattr(self, private, value.to(self._device))

    return property(fget, fset)


class EntityState(TorchVectorizedObject):
    def __init__(self):
        super().__init__()
        self._pos = None
        self._vel = None
        self._rot = None
        self._ang_vel = None

    pos = _state_prop("pos")
    vel = _state_prop("vel")
    rot = _state_prop("rot")
    ang_vel = _state_prop("ang_vel")

    def _reset(self, env_index):
        for name in ["pos", "rot", "vel", "ang_vel"]:
            attr = getattr(self, name)
            if attr is not None:
                if env_index is None:
                    setattr(self, name, torch.zeros_like(attr))
                else:
                    setattr(self, name, TorchUtils.where_from_index(env_index, 0, attr))

    def zero_grad(self):
        pass

    def _spawn(self, dim_c, dim_p):
        self.pos = torch.zeros(self.batch_dim, dim_p, device=self.device, dtype=torch.float32)
        self.vel = torch.zeros(self.batch_dim, dim_p, device=self.device, dtype=torch.float32)
        self.rot = torch.zeros(self.batch_dim, 1, device=self.device, dtype=torch.float32)
        self.ang_vel = torch.zeros(self.batch_dim, 1, device=self.device, dtype=torch.float32)


class AgentState(EntityState):
    def __init__(self):
        super().__init__()
        self._c = None
        self._force = None
        self._torque = None

    def _reset(self, env_index):
        super()._reset(env_index)

    def _spawn(self, dim_c, dim_p):
        super()._spawn(dim_c, dim_p)


class Action(TorchVectorizedObject):
    def __init__(self, u_range, u_multiplier, action_size):
        super().__init__()
        self.u = None
        self.u_range = u_range
        self.u_multiplier = u_multiplier
        self.action_size = action_size

    def _reset(self, env_index):
        if self.u is not None:
            if env_index is None:
                self.u = torch.zeros_like(self.u)
            else:
                self.u = TorchUtils.where_from_index(env_index, 0, self.u)


class Box:
    def __init__(self, length=0.3, width=0.1):
        self.length = length
        self.width = width


class Dynamics:
    def __init__(self):
        self._agent = None

    def reset(self, index=None):
        return


class Entity(TorchVectorizedObject):
    def __init__(self, name, shape=None, color=None, collide=True, **kw):
        super().__init__()
        self.name = name
        self.shape = shape
        self.color = color
        self.collide = collide
        self._state = EntityState()

    @property
    def state(self):
        return self._state

    @TorchVectorizedObject.batch_dim.setter
    def batch_dim(self, v):
        TorchVectorizedObject.batch_dim.fset(self, v)
        self._state.batch_dim = v

    @TorchVectorizedObject.device.setter
    def device(self, v):
        TorchVectorizedObject.device.fset(self, v)
        self._state.device = v

    def _set_state_property(self, prop, entity, new, batch_index):
        assert self.batch_dim is not None
        new = new.to(self.device)
        if batch_index is None:
            if len(new.shape) > 1 and new.shape[0] == self.batch_dim:
                prop.fset(entity, new)
            else:
                prop.fset(entity, new.repeat(self.batch_dim, 1))
        else:
            value = prop.fget(entity)
            value[batch_index] = new

    def set_pos(self, pos, batch_index):
        self._set_state_property(EntityState.pos, self.state, pos, batch_index)

    def set_vel(self, vel, batch_index):
        self._set_state_property(EntityState.vel, self.state, vel, batch_index)

    def set_rot(self, rot, batch_index):
        self._set_state_property(EntityState.rot, self.state, rot, batch_index)

    def _spawn(self, dim_c, dim_p):
        self.state._spawn(dim_c, dim_p)

    def _reset(self, env_index):
        self.state._reset(env_index)


class Agent(Entity):
    def __init__(
        self,
        name,
        shape=None,
        color=None,
        collide=True,
        render_action=False,
        u_range=1.0,
        u_multiplier=1.0,
        max_speed=None,
        dynamics=None,
        **kw,
    ):
        super().__init__(name, shape=shape, color=color, collide=collide)
        self.max_speed = max_speed
        self.u_range = u_range
        self.u_multiplier = u_multiplier
        self.dynamics = dynamics
        self.render_action = render_action
        self._state = AgentState()
        self._action = Action(u_range, u_multiplier, 2)

    @property
    def action(self):
        return self._action

    @Entity.batch_dim.setter
    def batch_dim(self, v):
        Entity.batch_dim.fset(self, v)
        self._action.batch_dim = v

    @Entity.device.setter
    def device(self, v):
        Entity.device.fset(self, v)
        self._action.device = v

    def _reset(self, env_index):
        self.action._reset(env_index)
        super()._reset(env_index)


class World(TorchVectorizedObject):
    def __init__(self, batch_dim, device, dt=0.1, x_semidim=None, y_semidim=None, **kw):
        super().__init__(batch_dim, device)
        self._agents = []
        self._dt = dt
        self._x_semidim = x_semidim
        self._y_semidim = y_semidim
        self._dim_p = 2
        self._dim_c = 0

    @property
    def agents(self):
        return self._agents

    @property
    def entities(self):
        return self._agents

    @property
    def dt(self):
        return self._dt

    @property
    def x_semidim(self):
        return self._x_semidim

    @property
    def y_semidim(self):
        return self._y_semidim

    def add_agent(self, agent):
        agent.batch_dim = self._batch_dim
        agent.device = self._device
        agent._spawn(dim_c=self._dim_c, dim_p=self._dim_p)
        self._agents.append(agent)

    def reset(self, env_index):
        for e in self.entities:
            e._reset(env_index)

    def step(self):
        raise NotImplementedError


class BaseScenario:
    def __init__(self):
        self._world = None

    @property
    def world(self):
        return self._world

    def env_make_world(self, batch_dim, device, **kwargs):
        self._world = self.make_world(batch_dim, device, **kwargs)
        return self._world

    def env_reset_world_at(self, env_index):
        self.world.reset(env_index)
        self.reset_world_at(env_index)

    def pre_step(self):
        return

    def post_step(self):
        return


def _mk_vmas():
    vmas = types.ModuleType("vmas")
    vmas.render_interactively = lambda *a, **k: None
    sim = types.ModuleType("vmas.simulator")
    core = types.ModuleType("vmas.simulator.core")
    for c in (TorchVectorizedObject, EntityState, AgentState, Action, Box, Entity, Agent, World):
        setattr(core, c.__name__, c)
    scen = types.ModuleType("vmas.simulator.scenario")
    scen.BaseScenario = BaseScenario
    utils = types.ModuleType("vmas.simulator.utils")
    utils.TorchUtils = TorchUtils
    utils.override = override
    utils.save_video = lambda *a, **k: None
    dyn = types.ModuleType("vmas.simulator.dynamics")
    dync = types.ModuleType("vmas.simulator.dynamics.common")
    dync.Dynamics = Dynamics
    rendering = types.ModuleType("vmas.simulator.rendering")
    vmas.simulator = sim
    sim.core, sim.scenario, sim.utils, sim.dynamics, sim.rendering = core, scen, utils, dyn, rendering
    dyn.common = dync
    return {
        "vmas": vmas,
        "vmas.simulator": sim,
        "vmas.simulator.core": core,
        "vmas.simulator.scenario": scen,
        "vmas.simulator.utils": utils,
        "vmas.simulator.dynamics": dyn,
        "vmas.simulator.dynamics.common": dync,
        "vmas.simulator.rendering": rendering,
    }


# --------------------------------------------------------------------------- #
# attribute-returning dummies (tensordict / torchrl / cvxpy / cv2 / pyglet)
# --------------------------------------------------------------------------- #
class _DummyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _make_dummy_class(name)


def _make_dummy_class(name):
    return _DummyMeta(name, (), {"__init__": lambda self, *a, **k: None})


class _DummyModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        c = _make_dummy_class(name)
        setattr(self, name, c)
        return c


class _DummyFinder:
    PREFIXES = ("tensordict", "torchrl", "cvxpy", "cv2", "pyglet", "gymnasium")

    def find_spec(self, fullname, path=None, target=None):
        import importlib.machinery

        if fullname.split(".")[0] in self.PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _DummyModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        return


_installed = False


def install():
    """Register the stand-ins and put the reference on sys.path (idempotent)."""
    global _installed
    if _installed:
        return
    import os

    os.environ.setdefault("CICD_TESTING", "true")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.modules["termcolor"] = _mk_termcolor()
    sys.modules["torchdiffeq"] = _mk_torchdiffeq()
    sys.modules.update(_mk_vmas())
    sys.meta_path.append(_DummyFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _installed = True


class RefEnv:
    """Drives a reference ``ScenarioRoadTraffic`` in the VMAS >= 1.4 call order.

    step(): set ``agent.action.u`` (clamped to +-u_range, as VMAS ``_set_action``
    does) -> ``world.step()`` -> ``reward(a)`` for all a -> ``observation(a)`` for
    all a -> ``info(a)`` for all a -> ``done()``.  Resetting done envs
    (``world.reset(e)``; ``scenario.reset_world_at(e)``) is left to the caller.
    """

    def __init__(self, parameters, num_envs):
        install()
        from sigmarl.scenarios.road_traffic import ScenarioRoadTraffic

        self.scenario = ScenarioRoadTraffic()
        self.scenario.parameters = parameters
        self.num_envs = num_envs
        self.world = self.scenario.env_make_world(num_envs, torch.device("cpu"))
        self.n_agents = len(self.world.agents)
        self.scenario.env_reset_world_at(None)

    def observe(self):
        return [self.scenario.observation(a).clone() for a in self.world.agents]

    def set_actions(self, actions):
        """actions: [B, N, 2] float32."""
        for i, a in enumerate(self.world.agents):
            u = actions[:, i, :].clone().to(torch.float32)
            rng = torch.tensor(
                [float(a.u_range[0]), float(a.u_range[1])], dtype=torch.float32
            ).unsqueeze(0)
            u = u.clamp(-rng, rng)
            a.action.u = u

    def step(self, actions):
        self.set_actions(actions)
        self.world.step()
        rew = [self.scenario.reward(a).clone() for a in self.world.agents]
        obs = [self.scenario.observation(a).clone() for a in self.world.agents]
        info = [TorchUtils.recursive_clone(self.scenario.info(a)) for a in self.world.agents]
        done = self.scenario.done().clone()
        return obs, rew, done, info

    def reset_env(self, e):
        self.scenario.env_reset_world_at(int(e))
