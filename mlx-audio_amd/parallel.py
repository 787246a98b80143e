"""Utterance sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm).  The path has exactly one exchange step: the finished waveforms travel to rank 0.  Utterances
(text chunks, pipeline.py:199-226) are independent, the 82M-parameter model is replicated, so there is no
other collective anywhere in the forward."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def balanced_assignment(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of utterances (cost ~ predicted frames) to ranks."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: loads[k])
        out[r].append(i)
        loads[r] += costs[i]
    return out


def gather_waveforms(local: torch.Tensor, out_root: Optional[torch.Tensor], dist, dst: int = 0, async_op: bool = False):
    """Equal-shape gather: rank r's [B, N] block lands in out_root[r*B:(r+1)*B] on `dst`.  With RCCL this is one
    grouped send/recv: every peer writes to the root over its own xGMI link (no ring).  async_op=True returns the work handle:
    the exchange then runs on the collective's own stream while the next batch is being synthesised (the caller must keep `local`
    and `out_root` untouched until handle.wait())."""
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        B = local.shape[0]
        chunks = [out_root[r * B : (r + 1) * B] for r in range(world)]
        return dist.gather(local, gather_list=chunks, dst=dst, async_op=async_op)
    return dist.gather(local, gather_list=None, dst=dst, async_op=async_op)


def gather_ragged(local: torch.Tensor, nsamples: torch.Tensor, dist, dst: int = 0):
    """Variable-length gather (RCCL has no gatherv).  Ranks may hold DIFFERENT numbers of utterances (the shards `shard_range` deals
    when the batch does not divide by the world size): first the per-rank batch sizes are all-gathered, then the per-utterance sample
    counts (padded to the largest batch), then every rank sends its [B_r, Nmax_r] block trimmed to its own longest utterance.
    Returns the list of 1-D waveforms in global order on `dst`, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    B = int(local.shape[0])
    if int(nsamples.numel()) != B:
        raise ValueError(f"gather_ragged: {B} waveforms but {int(nsamples.numel())} sample counts")
    nb = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(nb, torch.tensor([B], dtype=torch.int64, device=local.device))
    Bs = [int(t.item()) for t in nb]
    Bmax = max(Bs)
    mine = torch.zeros(Bmax, dtype=nsamples.dtype, device=nsamples.device)
    mine[:B] = nsamples
    counts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine)
    counts = [c[: Bs[r]] for r, c in enumerate(counts)]
    widths = [int(c.max().item()) if c.numel() else 0 for c in counts]
    if rank == dst:
        bufs = [torch.empty((Bs[r], widths[r]), dtype=local.dtype, device=local.device) for r in range(world)]
        reqs = []
        for r in range(world):
            if r == dst:
                bufs[r].copy_(local[:, : widths[r]])
            elif widths[r] > 0 and Bs[r] > 0:
                reqs.append(dist.irecv(bufs[r], src=r))
        for q in reqs:
            q.wait()
        out = []
        for r in range(world):
            cr = counts[r].tolist()
            for b in range(Bs[r]):
                out.append(bufs[r][b, : int(cr[b])])
        return out
    if widths[rank] > 0 and B > 0:
        dist.send(local[:, : widths[rank]].contiguous(), dst=dst)
    return None
