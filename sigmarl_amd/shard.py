"""Env-shard partitioning across the GPUs of one node and the rollout-slab exchange.

Environments are independent units (no cross-env read anywhere in the step), so the data path
shards with NO collective: rank r owns a contiguous env range and its own ``SigmaEnv``.  The only exchange is the
learner-boundary concat of the per-step rollout slab (observation, reward, done) -- one gather to the learner rank
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).  It is issued asynchronously on the communication
stream so that it overlaps the next fused step.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(total_envs: int, rank: int, world_size: int):
    """Contiguous [begin, end) of the envs owned by ``rank`` (remainder spread over the first ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(total_envs), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pack_slab(obs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """[B,N,D] obs, [B,N] reward, [B] done(u8) -> one contiguous fp32 slab [B, N*(D+1)+1] (wire format of the exchange)."""
    B, N, D = obs.shape
    width = N * (D + 1) + 1
    if out is None:
        out = torch.empty((B, width), dtype=torch.float32, device=obs.device)
    out[:, : N * D].copy_(obs.reshape(B, N * D))
    out[:, N * D: N * D + N].copy_(reward)
    out[:, -1].copy_(done.to(torch.float32))
    return out


def unpack_slab(slab: torch.Tensor, N: int, D: int):
    B = slab.shape[0]
    obs = slab[:, : N * D].reshape(B, N, D)
    reward = slab[:, N * D: N * D + N]
    done = slab[:, -1] > 0.5
    return obs, reward, done


class RolloutGather:
    """Double-buffered asynchronous gather of the per-step slab to ``dst`` (the learner rank)."""

    def __init__(self, local_envs: int, n_agents: int, obs_dim: int, device, dst: int = 0, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dst = dst
        self.N, self.D = n_agents, obs_dim
        width = n_agents * (obs_dim + 1) + 1
        self.send = [torch.empty((local_envs, width), dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = None
        if self.rank == dst and self.world > 1:
            self.recv = [[torch.empty((local_envs, width), dtype=torch.float32, device=device) for _ in range(self.world)] for _ in range(2)]
        self.pending = [None, None]
        self.k = 0

    def submit(self, obs, reward, done):
        """Packs the slab of this step and starts its gather; returns immediately (the previous use of the buffer is waited for)."""
        k = self.k & 1
        if self.pending[k] is not None:
            self.pending[k].wait()
            self.pending[k] = None
        slab = pack_slab(obs, reward, done, self.send[k])
        if self.world > 1:
            self.pending[k] = dist.gather(slab, self.recv[k] if self.rank == self.dst else None, dst=self.dst, group=self.group, async_op=True)
        self.k += 1
        return k

    def wait_all(self):
        for k in (0, 1):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def gathered(self, k):
        """On the learner rank: list of per-rank slabs of buffer k (after ``wait_all``)."""
        if self.world == 1:
            return [self.send[k]]
        return self.recv[k] if self.rank == self.dst else None
