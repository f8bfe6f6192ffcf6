"""Env-shard partitioning across the GPUs of one node and the rollout-buffer exchange.

Environments are independent units (no cross-env read anywhere in the step), so the data path
shards with NO collective: rank r owns a contiguous env range and its own ``SigmaEnv``.  The only exchange is the
learner-boundary concat of the rollout buffer (observation, reward, done of every step): the fused step kernel writes each
step's slab row-wise into a ``[T, B, W]`` chunk buffer (``sigmaenv_set_slab`` / ``sigmaenv_step_autoreset_n``), and once per chunk ONE
asynchronous collective ships it while the next chunk is being stepped into the second buffer (RCCL over xGMI when the backend is
"nccl"; gloo in the CPU tests).  One large message per link instead of a small one per step.  Two forms (``RolloutExchange(mode=...)``):
  "alltoall" (the default, and ``bench.py``'s)  the concatenated buffer is spread over the ranks by TIME SLICES -- rank r receives steps
             [r T/W, (r+1) T/W) of every rank's chunk, all envs of the node for 1/W of the steps: a data-parallel learner.  xGMI is
             point-to-point, so the record crosses it once over all 56 directed links.
  "gather"   everything to ONE learner rank (``dst``): the 7 links into that GPU carry the whole node's record and bound the rate.
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


def slab_width(n_agents: int, obs_dim: int) -> int:
    return n_agents * (obs_dim + 1) + 1


def pack_slab(obs: torch.Tensor, reward: torch.Tensor, done: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """[B,N,D] obs, [B,N] reward, [B] done -> fp32 slab [B, N*(D+1)+1]: the wire format the step kernel writes natively."""
    B, N, D = obs.shape
    if out is None:
        out = torch.empty((B, slab_width(N, D)), dtype=torch.float32, device=obs.device)
    out[:, : N * D].copy_(obs.reshape(B, N * D))
    out[:, N * D: N * D + N].copy_(reward)
    out[:, -1].copy_(done.to(torch.float32))
    return out


def unpack_slab(slab: torch.Tensor, N: int, D: int):
    """[..., N*(D+1)+1] -> obs [..., N, D], reward [..., N], done [...] (bool)."""
    lead = slab.shape[:-1]
    obs = slab[..., : N * D].reshape(*lead, N, D)
    reward = slab[..., N * D: N * D + N]
    done = slab[..., -1] > 0.5
    return obs, reward, done


class RolloutExchange:
    """Double-buffered rollout chunks ``[T, B, W]`` + one asynchronous collective per chunk.

    mode "gather":   the chunk of every rank goes to ``dst`` (ONE learner rank holds the concatenated buffer).  The learner's 7 links
                     take the whole node's record: at 145 GB/s per producing rank (0.06 ms per step) they are the bottleneck.
    mode "alltoall": the concatenated buffer is distributed over the ranks BY TIME SLICES (a data-parallel learner: rank r receives
                     steps [r T/W, (r+1) T/W) of every rank's chunk, i.e. ALL envs of the node for 1/W of the steps).  The same
                     bytes cross xGMI once, but over all 56 directed links instead of the 7 into one GPU."""

    def __init__(self, local_envs: int, n_agents: int, obs_dim: int, chunk_steps: int, device, dst: int = 0, group=None,
                 force_collective: bool = False, mode: str = "alltoall"):
        if mode not in ("gather", "alltoall"):
            raise ValueError("mode must be 'gather' or 'alltoall'")
        self.mode = mode
        self.group = group
        self.force = bool(force_collective)  # issue the collective even with one rank (exercises the RCCL path on a 1-GPU box)
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or self.force
        self.dst = dst
        self.N, self.D, self.T = n_agents, obs_dim, int(chunk_steps)
        if self.collective and self.world > 1:
            # the collectives below size every peer's part like the local one: the shards of one exchange must hold the same number of envs
            # (shard_range spreads a remainder over the first ranks -- pad the batch or pick a divisible one instead)
            counts = [None] * self.world
            dist.all_gather_object(counts, int(local_envs), group=group)
            if any(c != counts[0] for c in counts):
                raise ValueError(f"RolloutExchange needs equal env counts on every rank, got {counts}")
        shape = (self.T, local_envs, slab_width(n_agents, obs_dim))
        self.chunks = [torch.empty(shape, dtype=torch.float32, device=device) for _ in range(2)]
        self.recv = None
        # mode "alltoall": rank r receives the time slice [slices[r], slices[r + 1]) of every rank's chunk; T need not be a multiple of the world
        # size (the driver's 20-step run on 8 GPUs: slices of 2 or 3 steps)
        self.slices = [(r * self.T) // self.world for r in range(self.world + 1)]
        if self.collective and mode == "alltoall":
            my = self.slices[self.rank + 1] - self.slices[self.rank]
            # [source rank, steps of my time slice, envs of the source rank, W]
            self.recv = [torch.empty((self.world, my) + shape[1:], dtype=torch.float32, device=device) for _ in range(2)]
            row = shape[1] * shape[2]
            self._in_splits = [(self.slices[r + 1] - self.slices[r]) * row for r in range(self.world)]
            self._out_splits = [my * row] * self.world
        elif self.rank == dst and self.collective:
            self.recv = [[torch.empty(shape, dtype=torch.float32, device=device) for _ in range(self.world)] for _ in range(2)]
        self.pending = [None, None]
        self.cur, self.t = 0, 0
        self.completed = []  # (buffer index, valid steps) of every chunk whose exchange has been issued, in order
        self.valid_steps = [0, 0]  # steps recorded in each buffer when it was shipped (a final partial chunk ships whole, rows beyond are stale)

    def slot(self, writer_streams=None) -> torch.Tensor:
        """The ``[B, W]`` row block the NEXT step must be recorded into (pass it to ``SigmaEnv.set_slab``).  ``writer_streams``:
        the streams whose kernels write the rows when they are not torch's current stream (env shards on their own streams)."""
        if self.t == 0 and self.pending[self.cur] is not None:  # the buffer is about to be overwritten: its gather must be done
            if writer_streams:
                for st in writer_streams:  # Work.wait() orders the CURRENT stream behind the collective, nothing else
                    with torch.cuda.stream(st):
                        self.pending[self.cur].wait()
            else:
                self.pending[self.cur].wait()
            self.pending[self.cur] = None
        return self.chunks[self.cur][self.t]

    def chunk(self, writer_streams=None) -> torch.Tensor:
        """The whole ``[T, B, W]`` buffer the next T steps must be recorded into by ONE n-step launch (``SigmaEnv.step_autoreset_n``); follow
        the launch with ``commit``."""
        if self.t != 0:
            raise RuntimeError("chunk(): the current buffer is partly filled by per-step slots")
        self.slot(writer_streams)
        return self.chunks[self.cur]

    def commit(self, n_steps: int | None = None, writer_streams=None):
        """The launch that fills ``chunk()`` (its first ``n_steps`` rows, default all) has been enqueued: ship the buffer."""
        self.t = self.T if n_steps is None else int(n_steps)
        self.flush(writer_streams)

    def advance(self, writer_streams=None):
        """Call after the step that filled ``slot()`` has been enqueued; ships the chunk when it is full.  ``writer_streams``: as in ``slot``."""
        self.t += 1
        if self.t == self.T:
            self.flush(writer_streams)

    def flush(self, writer_streams=None):
        """Ships the current chunk (``valid_steps[k]`` of its T rows are new).  The collective is issued on torch's current stream, which is
        first ordered behind ``writer_streams`` (the streams whose kernels wrote the rows)."""
        if self.t == 0:
            return
        k = self.cur
        if writer_streams:
            cur = torch.cuda.current_stream(self.chunks[k].device)
            for st in writer_streams:
                if st != cur:
                    cur.wait_stream(st)
        self.valid_steps[k] = self.t
        if self.collective and self.mode == "alltoall":
            self.pending[k] = dist.all_to_all_single(self.recv[k].view(-1), self.chunks[k].view(-1), self._out_splits, self._in_splits, group=self.group,
                                                     async_op=True)
        elif self.collective:
            self.pending[k] = dist.gather(self.chunks[k], self.recv[k] if self.rank == self.dst else None, dst=self.dst, group=self.group,
                                          async_op=True)
        self.completed.append((k, self.t))
        self.cur ^= 1
        self.t = 0

    def wait_all(self):
        for k in (0, 1):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

    def gathered(self, k):
        """mode "gather", on the learner rank: per-rank chunk buffers ``[T, B_r, W]`` of buffer k (after ``wait_all``)."""
        if not self.collective:
            return [self.chunks[k]]
        if self.mode == "alltoall":
            raise RuntimeError("alltoall exchange: use time_slice(k)")
        return self.recv[k] if self.rank == self.dst else None

    def time_slice(self, k):
        """mode "alltoall": this rank's share of buffer k (after ``wait_all``): ``[my steps, W * B, W_row]`` -- steps
        ``[slices[rank], slices[rank + 1])`` of the chunk for the envs of ALL ranks (ranks own contiguous env ranges, in rank order)."""
        if not self.collective:
            return self.chunks[k]
        r = self.recv[k]  # [source rank, T / W, B, W_row]
        return r.permute(1, 0, 2, 3).reshape(r.shape[1], r.shape[0] * r.shape[2], r.shape[3])
