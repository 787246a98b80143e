"""Multi-process (gloo, world_size 2) tests of the utterance-shard + gather path; runs on CPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mlx_audio_amd import parallel as par


def test_shard_range_covers_everything():
    for n in (1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = par.shard_range(n, world, r)
                got += list(range(lo, hi))
            assert got == list(range(n))
            sizes = [par.shard_range(n, world, r)[1] - par.shard_range(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_assignment():
    costs = [650, 10, 640, 20, 300, 310, 5, 5]
    a = par.balanced_assignment(costs, 2)
    assert sorted(a[0] + a[1]) == list(range(8))
    loads = [sum(costs[i] for i in g) for g in a]
    assert abs(loads[0] - loads[1]) <= 30


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B, N = 3, 50
        local = torch.arange(B * N, dtype=torch.float32).reshape(B, N) + 1000 * rank
        out = torch.zeros((world * B, N)) if rank == 0 else None
        par.gather_waveforms(local, out, dist)
        ok = True
        if rank == 0:
            for r in range(world):
                ok &= bool(torch.equal(out[r * B : (r + 1) * B], torch.arange(B * N, dtype=torch.float32).reshape(B, N) + 1000 * r))
        # asynchronous form, two batches in flight on alternating buffers (what bench.py does to hide the exchange)
        outs = [torch.zeros((world * B, N)) if rank == 0 else None for _ in range(2)]
        locs = [local + 7.0 * j for j in range(2)]
        hs = [par.gather_waveforms(locs[j], outs[j], dist, async_op=True) for j in range(2)]
        for h in hs:
            h.wait()
        if rank == 0:
            for j in range(2):
                for r in range(world):
                    ok &= bool(torch.equal(outs[j][r * B : (r + 1) * B], torch.arange(B * N, dtype=torch.float32).reshape(B, N) + 1000 * r + 7.0 * j))
        # ragged
        ns = torch.tensor([10 + rank, 50, 3 * (rank + 1)], dtype=torch.int32)
        rag = par.gather_ragged(local, ns, dist)
        if rank == 0:
            ok &= len(rag) == world * B
            for r in range(world):
                for b in range(B):
                    n = [10 + r, 50, 3 * (r + 1)][b]
                    exp = (torch.arange(B * N, dtype=torch.float32).reshape(B, N) + 1000 * r)[b, :n]
                    ok &= bool(torch.equal(rag[r * B + b], exp))
        else:
            ok &= rag is None
        # ragged with a DIFFERENT number of utterances per rank (7 utterances over 2 ranks: shards of 4 and 3)
        lo, hi = par.shard_range(7, world, rank)
        Bl = hi - lo
        loc = torch.arange(Bl * N, dtype=torch.float32).reshape(Bl, N) + 1000 * rank
        ns = torch.tensor([5 + 3 * (lo + b) for b in range(Bl)], dtype=torch.int32)
        rag = par.gather_ragged(loc, ns, dist)
        if rank == 0:
            ok &= len(rag) == 7
            for g in range(7):
                r = next(k for k in range(world) if par.shard_range(7, world, k)[0] <= g < par.shard_range(7, world, k)[1])
                b = g - par.shard_range(7, world, r)[0]
                Br = par.shard_range(7, world, r)[1] - par.shard_range(7, world, r)[0]
                exp = (torch.arange(Br * N, dtype=torch.float32).reshape(Br, N) + 1000 * r)[b, : 5 + 3 * g]
                ok &= bool(torch.equal(rag[g], exp))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gather_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE")}
    return env


def test_bench_launcher_spawns_n_ranks_gloo():
    """`python bench.py --gpus 2` with no torchrun environment starts two ranks by itself (the launcher process never touches a GPU),
    they rendezvous on 127.0.0.1, and exactly ONE JSON line -- rank 0's, with n_gpus == --gpus -- comes out on stdout."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["max_over_ranks"] == 2.0  # MAX over ranks of (rank + 1)


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    # and under a torchrun-style environment that matches, the same script runs as a rank
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and '"n_gpus": 1' in r.stdout


# ---- the product entry: ShardedSynth with a stub model that returns a deterministic waveform per chunk ------------------------------
class _StubModel:
    """batch_call of kokoro.Model, no GPU: chunk `ps` with style row r -> 600 * len(ps) samples = f(ps, r[0], position), durations 1..T."""

    def __init__(self):
        self.calls = []

    def batch_call(self, phonemes, ref_s, speed=1, seed=None):
        from mlx_audio_amd.kokoro import Model

        self.calls.append(list(phonemes))
        out = []
        for p, r in zip(phonemes, ref_s):
            n = 600 * len(p)
            wav = (torch.arange(n, dtype=torch.float32) * 1e-3 + float(sum(map(ord, p)) % 97) + float(r[0])).reshape(1, n)
            out.append(Model.Output(audio=wav, pred_dur=torch.arange(1, len(p) + 3, dtype=torch.int32)))
        return out

    def __call__(self, ps, ref_s, speed=1, return_output=False):
        return self.batch_call([ps], np.asarray(ref_s).reshape(1, 256), speed)[0]


def _stub_pipeline():
    from mlx_audio_amd.pipeline import KokoroPipeline

    p = KokoroPipeline(lang_code="e", model=_StubModel(), repo_id="m", g2p=lambda t: (t, None))  # identity G2P, non-English branch
    p.voices["v"] = np.repeat(np.arange(510, dtype=np.float32)[:, None, None], 256, axis=2)
    return p


_TEXT = "\n".join("abcdefghij"[: 1 + (7 * i) % 10] * (1 + (5 * i) % 9) for i in range(23))


def _expected():
    """The plain single-process pipeline, chunk by chunk."""
    p = _stub_pipeline()
    return [(r.text_index, r.phonemes, np.asarray(r.audio).reshape(-1)) for r in p(_TEXT, voice="v")]


def test_sharded_synth_world1_equals_the_plain_pipeline():
    exp = _expected()
    p = _stub_pipeline()
    got = par.ShardedSynth(p, dist=None, batch_size=4)(_TEXT, voice="v")
    assert [(r.text_index, r.phonemes) for r in got] == [(a, b) for a, b, _ in exp]
    for r, (_, _, w) in zip(got, exp):
        np.testing.assert_array_equal(np.asarray(r.audio).reshape(-1), w)
    assert max(len(c) for c in p.model.calls) <= 4 and sum(len(c) for c in p.model.calls) == len(exp)


def _synth_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = _stub_pipeline()
        synth = par.ShardedSynth(p, dist=dist, batch_size=4, device=torch.device("cpu"))
        res = synth(_TEXT if rank == 0 else "ignored on this rank", voice="v")
        mine = synth.last_assignment[rank]
        ran = sorted(ps for c in p.model.calls for ps in c)
        if rank == 0:
            q.put((rank, [(r.text_index, r.phonemes, np.asarray(r.audio).reshape(-1), np.asarray(r.pred_dur)) for r in res], synth.last_assignment, ran))
        else:
            q.put((rank, res, mine, ran))
    finally:
        dist.destroy_process_group()


def test_sharded_synth_world2_gloo():
    """Rank 0 plans the chunks, both ranks synthesise their balanced share, rank 0 gets every waveform and duration vector back in TEXT order;
    the other rank returns None and ran only its own chunks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_synth_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = {}
    for _ in ps:
        item = q.get(timeout=180)
        got[item[0]] = item[1:]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = _expected()
    res, assign, ran0 = got[0]
    assert got[1][0] is None
    assert [(a, b) for a, b, _, _ in res] == [(a, b) for a, b, _ in exp]
    for (_, ps_, w, d), (_, _, we) in zip(res, exp):
        np.testing.assert_array_equal(w, we)
        np.testing.assert_array_equal(d, np.arange(1, len(ps_) + 3))
    # the work was really split, and balanced by predicted frames
    assert sorted(assign[0] + assign[1]) == list(range(len(exp))) and assign[0] and assign[1]
    loads = [sum(len(exp[i][1]) for i in g) for g in assign]
    assert abs(loads[0] - loads[1]) <= max(len(e[1]) for e in exp)
    assert ran0 == sorted(exp[i][1] for i in assign[0]) and got[1][2] == sorted(exp[i][1] for i in assign[1])
