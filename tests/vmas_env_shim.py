"""An ``Environment``-shaped driver: what ``vmas.simulator.environment.Environment`` (VMAS 1.4.3, the version the reference pins in
requirements.txt:9) does to a scenario between ``VmasEnv`` and the ``BaseScenario`` callbacks -- restated, with the same discipline as
``tests/golden/gen/refshim.py``: vmas is third-party, absent from the build container and not installable, so this follows its published source
from memory and is **unverified** against the real package.  Only what touches the scenario / world / agents is kept (no rendering, no gym
spaces objects, no discrete or multidiscrete actions, no communication actions -- SigmaRL uses continuous actions and ``dim_c = 0``).

Method by method (names as in vmas):
  __init__         scenario.env_make_world(num_envs, device, **kwargs); agents = world.policy_agents; reset(seed)
  reset            scenario.env_reset_world_at(None); steps := 0; observations of all agents
  reset_at         scenario.env_reset_world_at(index); steps[index] := 0
  step             checks one action tensor [num_envs, action_size] per policy agent; _set_action per agent; scenario.env_process_action per world
                   agent; scenario.pre_step(); world.step(); scenario.post_step(); steps += 1; get_from_scenario(obs, rewards, infos, dones)
  _set_action      clone / detach / to(device); NaN check; agent.action.u := zeros [B, action_size]; (clamp_actions: clamp to +-u_range_tensor);
                   asserts |action| <= u_range_tensor; agent.action.u := action (float32) * u_multiplier_tensor; u_noise; communication only if
                   world.dim_c > 0 and not agent.silent
  get_from_scenario  scenario.reward(agent).clone() for ALL agents, then the recursive clone of scenario.observation(agent) for all agents, then of
                   scenario.info(agent) for all agents, then done() once -- the order SURVEY.md (Appendix A) records for VMAS >= 1.4 and the one the
                   golden generator's stand-in uses.  SigmaRL depends on it: reward(last agent) refreshes every short-term reference path
                   (road_traffic.py:1243) that observation(agent 0) then reads for ALL agents (:1358-1359); an interleaved per-agent loop (older
                   VMAS) would observe stale paths.  [unverified-3p]
  done             scenario.done().clone() (+ steps >= max_steps when max_steps is given)
"""
from __future__ import annotations

from typing import List

import torch


def _recursive_clone(x):
    if isinstance(x, torch.Tensor):
        return x.clone()
    if isinstance(x, dict):
        return {k: _recursive_clone(v) for k, v in x.items()}
    return x


class EnvironmentShim:
    def __init__(self, scenario, num_envs: int = 32, device="cpu", max_steps=None, continuous_actions: bool = True, seed=None, clamp_actions: bool = False,
                 grad_enabled: bool = False, **kwargs):
        assert continuous_actions, "SigmaRL drives VMAS with continuous actions (mappo_cavs.py:170-177)"
        self.scenario = scenario
        self.num_envs = self.batch_dim = num_envs
        self.device = torch.device(device)
        self.max_steps = max_steps
        self.continuous_actions = continuous_actions
        self.clamp_action = clamp_actions
        self.grad_enabled = grad_enabled
        self.world = self.scenario.env_make_world(self.num_envs, self.device, **kwargs)
        self.agents = self.world.policy_agents
        self.n_agents = len(self.agents)
        self.steps = None
        self.reset(seed=seed)

    def seed(self, seed=None):
        if seed is None:
            seed = 0
        torch.manual_seed(seed)
        return [seed]

    def reset(self, seed=None, return_observations: bool = True, return_info: bool = False, return_dones: bool = False):
        if seed is not None:
            self.seed(seed)
        self.scenario.env_reset_world_at(env_index=None)
        self.steps = torch.zeros(self.num_envs, device=self.device)
        result = self.get_from_scenario(get_observations=return_observations, get_infos=return_info, get_rewards=False, get_dones=return_dones)
        return result[0] if result and len(result) == 1 else result

    def reset_at(self, index: int, return_observations: bool = True, return_info: bool = False, return_dones: bool = False):
        assert 0 <= index < self.num_envs
        self.scenario.env_reset_world_at(index)
        self.steps[index] = 0
        result = self.get_from_scenario(get_observations=return_observations, get_infos=return_info, get_rewards=False, get_dones=return_dones)
        return result[0] if result and len(result) == 1 else result

    def get_from_scenario(self, get_observations: bool, get_rewards: bool, get_infos: bool, get_dones: bool):
        if not (get_infos or get_dones or get_observations or get_rewards):
            return None
        obs, rewards, infos = [], [], []
        if get_rewards:
            for agent in self.agents:
                rewards.append(self.scenario.reward(agent).clone())
        if get_observations:
            for agent in self.agents:
                obs.append(_recursive_clone(self.scenario.observation(agent)))
        if get_infos:
            for agent in self.agents:
                infos.append(_recursive_clone(self.scenario.info(agent)))
        result = []
        if get_observations:
            result.append(obs)
        if get_rewards:
            result.append(rewards)
        if get_dones:
            result.append(self.done())
        if get_infos:
            result.append(infos)
        return result

    def get_agent_action_size(self, agent) -> int:
        return agent.action_size + (self.world.dim_c if not agent.silent else 0)

    def step(self, actions: List[torch.Tensor]):
        assert len(actions) == self.n_agents, f"Expecting actions for {self.n_agents}, got {len(actions)} actions"
        actions = list(actions)
        for i in range(len(actions)):
            if not isinstance(actions[i], torch.Tensor):
                actions[i] = torch.tensor(actions[i], dtype=torch.float32, device=self.device)
            if len(actions[i].shape) == 1:
                actions[i] = actions[i].unsqueeze(-1)
            assert actions[i].shape[0] == self.num_envs, f"Actions used in input of env must be of len {self.num_envs}, got {actions[i].shape[0]}"
            assert actions[i].shape[1] == self.get_agent_action_size(self.agents[i])
        for i, agent in enumerate(self.agents):
            self._set_action(actions[i], agent)
        for agent in self.world.agents:
            self.scenario.env_process_action(agent)
        self.scenario.pre_step()
        self.world.step()
        self.scenario.post_step()
        self.steps += 1
        return self.get_from_scenario(get_observations=True, get_infos=True, get_rewards=True, get_dones=True)

    def done(self):
        dones = self.scenario.done().clone()
        if self.max_steps is not None:
            dones = dones + (self.steps >= self.max_steps)
        return dones

    def _set_action(self, action, agent):
        action = action.clone()
        if not self.grad_enabled:
            action = action.detach()
        action = action.to(self.device)
        assert not action.isnan().any()
        agent.action.u = torch.zeros(self.batch_dim, agent.action_size, device=self.device, dtype=torch.float32)
        assert action.shape[1] == self.get_agent_action_size(agent)
        if self.clamp_action and self.continuous_actions:
            physical_action = action[..., : agent.action_size]
            a_range = agent.action.u_range_tensor.unsqueeze(0).expand(physical_action.shape)
            action = torch.cat([physical_action.clamp(-a_range, a_range), action[..., agent.action_size:]], dim=-1)
        action_index = 0
        if agent.movable and agent.action_script is None:
            physical_action = action[:, action_index: action_index + agent.action_size]
            action_index += agent.action_size
            assert not torch.any(torch.abs(physical_action) > agent.action.u_range_tensor), \
                f"Physical actions of agent {agent.name} are out of its range {agent.action.u_range}"
            agent.action.u = physical_action.to(torch.float32)
            agent.action.u = agent.action.u * agent.action.u_multiplier_tensor
            if agent.action.u_noise > 0:
                agent.action.u = agent.action.u + torch.randn(*agent.action.u.shape, device=self.device, dtype=torch.float32) * agent.action.u_noise
        if self.world.dim_c > 0 and not agent.silent:
            raise AssertionError("SigmaRL's world has no communication channel")

    def to(self, device):
        self.device = torch.device(device)
        self.world.to(device)
