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


class ShardedSynth:
    """The product multi-GPU entry: ONE request's text chunks (the reference's shard unit, pipeline.py:199-226,371) synthesised by all
    the GPUs of a node, waveforms back on rank 0 in text order.  One process per GPU; every rank constructs it around its own KokoroPipeline
    (the 164 MB model is replicated) and calls it with the same arguments (SPMD); only rank 0's `text` is used.

      1. rank 0 runs the front end (split, G2P, the 510-phoneme chunk planner) and broadcasts the phoneme strings;
      2. chunks are dealt to ranks by predicted work, longest first (`balanced_assignment`, cost = phonemes x `frames_per_phoneme` prior:
         sum of F sets a rank's time; the predicted durations are only known after the text stage);
      3. every rank runs ITS chunks as padded batches (`KokoroPipeline.plan_batches` + `Model.batch_call`);
      4. the ONE exchange of the path: `gather_ragged` (sample counts all-gathered, then every peer sends its block to rank 0 over its own
         xGMI link; RCCL has no gatherv) + a gather of the small per-chunk duration vectors;
      5. rank 0 restores text order and joins the timestamps (pipeline.py:292-328).
    Returns the list of KokoroPipeline.Result on rank 0, None on the others.  dist=None (or world size 1) runs the same plan without any
    collective and is bit-identical to `pipeline(text, voice, speed, batch_size=...)`."""

    def __init__(self, pipeline, dist=None, batch_size: int = 32, frames_per_phoneme: float = 5.0, dst: int = 0, device=None):
        self.pipeline, self.dist, self.batch_size, self.fpp, self.dst = pipeline, dist, int(batch_size), float(frames_per_phoneme), dst
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.device = device
        self.last_assignment: List[List[int]] = []

    def plan(self, lengths: Sequence[int]) -> List[List[int]]:
        """chunk indices per rank, each list in text order (so that a rank's batch plan is `plan_batches` of a text-ordered list, as in the
        single-process pipeline)."""
        return [sorted(g) for g in balanced_assignment([n * self.fpp for n in lengths], self.world)]

    def __call__(self, text, voice: str, speed=1, split_pattern: Optional[str] = r"\n+", seed: Optional[int] = None):
        import numpy as np

        pipe, dist = self.pipeline, self.dist
        if voice is None:
            raise ValueError("Specify a voice")
        chunks = None
        if self.rank == self.dst:
            chunks = list(pipe._chunks(text, split_pattern))  # (text_index, graphemes, phonemes, tokens | None)
        phon = [[c[2] for c in chunks]] if chunks is not None else [None]
        if self.world > 1:
            dist.broadcast_object_list(phon, src=self.dst)
        phon = phon[0]
        assign = self.plan([len(p) for p in phon])
        self.last_assignment = assign
        mine = assign[self.rank]
        pack = pipe.load_voice(voice)
        outs = {}
        for bi, idx in enumerate(type(pipe).plan_batches([len(phon[i]) for i in mine], self.batch_size)):
            ps_list = [phon[mine[j]] for j in idx]
            rows = np.stack([np.asarray(pack[len(ps) - 1], np.float32).reshape(256) for ps in ps_list])  # pipeline.py:236
            kw = {} if seed is None else {"seed": int(seed) + 1000 * self.rank + bi}
            for j, o in zip(idx, pipe.model.batch_call(ps_list, rows, speed, **kw)):
                outs[mine[j]] = o
        if self.world == 1:
            return self._results(chunks, [outs[i].audio for i in range(len(phon))], [outs[i].pred_dur for i in range(len(phon))])
        # ---- the exchange: waveforms (device, ragged) + duration vectors (tiny, host objects)
        wavs = [outs[i].audio.reshape(-1) for i in mine]
        dev = self.device if self.device is not None else (wavs[0].device if wavs else torch.device("cpu"))
        ns = torch.tensor([int(w.numel()) for w in wavs], dtype=torch.int64, device=dev)
        local = torch.zeros((len(wavs), int(ns.max().item()) if len(wavs) else 0), dtype=torch.float32, device=dev)
        for k, w in enumerate(wavs):
            local[k, : w.numel()] = w
        flat = gather_ragged(local, ns, dist, dst=self.dst)
        durs = [None] * self.world if self.rank == self.dst else None
        dist.gather_object([(None if outs[i].pred_dur is None else torch.as_tensor(outs[i].pred_dur).cpu()) for i in mine], durs, dst=self.dst)
        if self.rank != self.dst:
            return None
        order = [i for r in range(self.world) for i in assign[r]]  # gather_ragged returns rank-major order
        audio, pred = [None] * len(phon), [None] * len(phon)
        dflat = [d for r in range(self.world) for d in durs[r]]
        for pos, i in enumerate(order):
            audio[i], pred[i] = flat[pos].reshape(1, -1), dflat[pos]
        return self._results(chunks, audio, pred)

    def _results(self, chunks, audio, pred):
        from .kokoro import Model

        pipe = self.pipeline
        res = []
        for (gi, gs, ps, tks), a, d in zip(chunks, audio, pred):
            if tks is not None and d is not None:
                type(pipe).join_timestamps(tks, d)
            res.append(pipe.Result(graphemes=gs, phonemes=ps, tokens=tks, output=Model.Output(audio=a, pred_dur=d), text_index=gi))
        return res
