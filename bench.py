#!/usr/bin/env python3
"""Headline benchmark: audio-seconds per wall-second (xRT), Kokoro-82M, batch = 32 fixed 128-phoneme
utterances per GPU (BASELINE.json configs[1]); utterances are sharded over ranks (weak scaling) and the
waveforms are gathered to rank 0 with one RCCL gather per step.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one kk_forward over one batch (the whole acoustic path incl. duration prediction; the durations that
are REALISED are pinned to 5 frames per token so every rank does identical work: T = 130, F = 650,
390 000 samples = 16.25 s per utterance).  Inputs (ids, style rows, weights) are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT,):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

N_PHONEMES = 128
FRAMES_PER_TOKEN = 5
BATCH_PER_GPU = 32
SR = 24000

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (dense; never the 2:1 sparse marketing numbers)
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
PEAK_HBM_GBS = 8000.0
PEAK_FP8_TFLOPS = 5000.0  # dense block-scaled fp8 (MI355X_MICROARCH.md)


def cpu_baseline(cfg, w, utt, ref_s):
    """The CPU oracle (kind "port": our restatement of the reference's algorithm) on the host cores of this
    box, on a bounded sample of the same workload: ONE config-2 utterance (T = 130, F = 650, 16.25 s audio)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kokoro_oracle as O

    orc = O.KokoroOracle(w, cfg)
    cores = torch.get_num_threads()
    dur = np.full(N_PHONEMES + 2, FRAMES_PER_TOKEN, np.int32)
    noise = np.random.default_rng(0).standard_normal((1, 600 * int(dur.sum()), 9)).astype(np.float32)
    t0 = time.time()
    a, _ = orc.forward(utt, ref_s, 1.0, forced_dur=dur, sine_noise=noise)
    dt = time.time() - t0
    return {"value": (a.shape[0] / SR) / dt, "unit": "audio-sec/sec", "cores": int(cores), "kind": "port",
            "sample": f"1 utterance of the same workload (T=130, F=650, 16.25 s audio) in {dt:.1f} s; oracle/kokoro_oracle.py, torch-CPU fp32"}


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N fresh child processes of this script, one per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), BEFORE this process makes any GPU call -- it never does:
    it only waits.  Rank 0 inherits stdout, so its single JSON line is this command's output; the other ranks' stdout goes to stderr.
    Returns non-zero if any rank failed."""
    import subprocess

    env = dict(os.environ)
    env.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=e, stdout=None if r == 0 else sys.stderr))
    rc = 0
    for r, p in enumerate(procs):
        c = p.wait()
        if c != 0:
            print(f"bench.py: rank {r} exited with code {c}", file=sys.stderr)
            rc = rc or (c if c > 0 else 1)
    return rc


def dry_run(args, rank, world):
    """Launcher rehearsal without a GPU (tests/test_parallel_cpu.py): rendezvous over gloo, the bench's barrier + MAX-over-ranks
    timing reduction and rank 0's JSON line -- everything around the timed region, none of the engine."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "dry-run (launcher rehearsal, no GPU work)", "value": 0.0, "unit": "audio-sec/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "max_over_ranks": float(t.item())}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default=os.environ.get("KK_BENCH_DTYPE", "bfloat16"), choices=["float32", "bfloat16"])
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event brackets")
    ap.add_argument("--no-latency", action="store_true", help="skip the B=1 p50 latency measurement")
    ap.add_argument("--quantized", action="store_true",
                    help="BASELINE config 5: the checkpoint goes through the MLX 8-bit group quantisation (group 64) of the reference's predicate and "
                         "its linears run on the fp8 matrix instruction; use with --batch 64.  The default run is config 2 (the headline).")
    ap.add_argument("--quantization-kernel", default="exact", choices=["exact", "mxfp8"],
                    help="with --quantized: 'exact' (default) = the dequantised weights scale*q+bias on the bf16 MFMA kernels, the reference's arithmetic "
                         "(tts/utils.py:241-260); 'mxfp8' = the opt-in e4m3 kernels.  The other one is measured too and reported next to `value`.")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel of the forward eagerly (default: hipGraph replay, kk_set_graph_mode)")
    ap.add_argument("--streams", type=int, default=None,
                    help="batches in flight (kokoro): consecutive steps alternate over this many HIP streams, each with its own engine instance, "
                         "workspace and captured graph, so the latency-bound text / LSTM phases of step i+1 overlap the conv-bound generator of step i "
                         "(1 = strictly one step after the other; default 2).  csm: whole B-stream jobs in flight, each on its own stream, host thread "
                         "and model instance (default 4)")
    ap.add_argument("--config", default="kokoro", choices=["kokoro", "csm"],
                    help="kokoro (default): BASELINE configs[1], the headline.  csm: configs[3], CSM-1B + Mimi at the SURVEY 8(d) pin -- B = 8 streams, each a "
                         "10 s reference-audio prompt (125 Mimi frames) + 64 text ids, 125 frames generated greedily, Mimi decode included; a step is one "
                         "whole generation.")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal on CPU (gloo): rendezvous, barrier, MAX reduction, rank 0's line")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched (or will touch) a GPU.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # n_gpus in the JSON line can never disagree with --gpus
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus} (or unset WORLD_SIZE)")
    if args.dry_run:
        return dry_run(args, rank, world)
    torch.cuda.set_device(local_rank)
    if args.config == "csm":
        out = bench_csm(args, rank, world)
        if rank == 0:
            print(json.dumps(out))
        if world > 1:
            import torch.distributed as dist_mod

            dist_mod.destroy_process_group()
        return
    out = bench_kokoro(args, rank, world)
    if args.quantized:  # (every rank: the second configuration has the same collectives)
        # config 5 reports BOTH arithmetic choices for the quantised layer set: the default ("exact": dequantised weights, bf16 activations -- the
        # reference's semantics) is `value`; the opt-in fp8 kernels (or vice versa with --quantization-kernel mxfp8) ride along
        other = "mxfp8" if args.quantization_kernel == "exact" else "exact"
        alt = bench_kokoro(args, rank, world, quantization_kernel=other, brief=True)
        out["quantization_kernel"] = args.quantization_kernel
        out["other_quantization_kernel"] = {"kernel": other, "value": alt["value"], "ms_per_step": alt["ms_per_step"],
                                            "note": "mxfp8 = e4m3 weights AND activations on v_mfma_scale_f32_32x32x64_f8f6f4 (opt-in, narrower than the reference's arithmetic)"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist_mod

        dist_mod.destroy_process_group()


def bench_csm(args, rank, world):
    """BASELINE configs[3] at the SURVEY 8(d) pin.  Replicas only: the streams of a batch are independent and there is no exchange step, every
    rank runs its own B = 8 batch; `value` is the aggregate."""
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    import mlx_audio_amd.params as P
    from mlx_audio_amd.mimi import Mimi, mimi_202407
    from mlx_audio_amd.sesame import Model, Segment

    B, REF_S, N_TEXT, FRAMES = (args.batch if args.batch != BATCH_PER_GPU else 8), 10.0, 64, 125
    cfg = P.csm_config()
    w = P.csm_synth_checkpoint(cfg, 0)
    mcfg = P.mimi_config(32)
    mw = P.mimi_synth_checkpoint(mcfg, 0, encode=True)
    # NS jobs in flight (--streams, default 2): each job is one whole B-stream batch (prompt encode -> prompt block -> FRAMES frames -> decode) on
    # its own HIP stream and host thread with its own KV caches and graphs on ONE shared copy of the weights.  A frame is ~900 launches of a few microseconds each, so one
    # job leaves most of the chip idle; two interleave.  The K timed steps are dealt to the jobs from one queue.
    NS = max(1, int(args.streams if args.streams is not None else 4))
    # ONE copy of the CSM-1B and Mimi weights; every further job in flight shares them (kk_csm_share, Mimi.share: own KV caches / graphs / workspaces)
    models = [Model(cfg, mimi=Mimi(mimi_202407(32), mw, compute_dtype="bfloat16"), weights=w, weight_dtype="bfloat16")]  # bf16 checkpoint: matrices streamed as bf16, fp32 arithmetic
    models += [models[0].share() for _ in range(NS - 1)]
    model = models[0]
    rng = np.random.default_rng(1000 + rank)
    ctx, texts = [], []
    for b in range(B):
        ref = (0.1 * rng.standard_normal(int(24000 * REF_S))).astype(np.float32)
        ctx.append([Segment(speaker=0, text=rng.integers(0, cfg["text_vocab_size"], N_TEXT // 2).tolist(), audio=ref)])
        texts.append(rng.integers(0, cfg["text_vocab_size"], N_TEXT // 2).tolist())
    dev = model.model.device
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(NS - 1)]

    def step(k=0):
        # the whole job: reference audio -> Mimi.encode -> prompt frames -> prompt block -> FRAMES frames (greedy: deterministic) -> Mimi.decode
        prompts = models[k].prompt_frames_batch(ctx, texts, 0, voice_match=False)  # the B reference clips in one Mimi.encode call
        return models[k].generate_batch(prompts, max_audio_length_ms=80 * FRAMES, temperature=0.0, stop_on_eos=False), prompts[0][0].shape[0]

    def run_steps(n, jobs):
        """n steps over `jobs` jobs in flight (threads pull step numbers from one counter; a thread's launches go to its own stream)."""
        if jobs == 1:
            for _ in range(n):
                step(0)
            return
        import threading

        lock, left = threading.Lock(), [n]

        def worker(k):
            with torch.cuda.stream(streams[k]):
                while True:
                    with lock:
                        if left[0] <= 0:
                            break
                        left[0] -= 1
                    step(k)
                streams[k].synchronize()

        th = [threading.Thread(target=worker, args=(k,)) for k in range(jobs)]
        for t in th:
            t.start()
        for t in th:
            t.join()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    S = 0
    for k in range(NS):  # warm-up one job at a time: each model captures its frame graph here
        with torch.cuda.stream(streams[k]):
            for _ in range(max(1, args.warmup)):
                res, S = step(k)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, NS)
    barrier()
    dt = time.perf_counter() - t0
    dt_one = None
    if NS > 1:  # the same K steps with one job in flight, for reference
        barrier()
        t1 = time.perf_counter()
        run_steps(args.steps, 1)
        barrier()
        dt_one = time.perf_counter() - t1
    # the frame step on its own, HIP events on the stream the frames are launched on (torch's current stream): FRAMES single-token frames
    csm = model.model
    n = cfg["audio_num_codebooks"]
    tok = torch.zeros((B, 1, n + 1), dtype=torch.int32, device=csm.device)
    msk = torch.zeros((B, 1, n + 1), dtype=torch.float32, device=csm.device)
    msk[:, 0, :n] = 1
    for _ in range(3):
        codes = csm.generate_frame(tok, msk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nfr = 50
    e0.record()
    for _ in range(nfr):
        codes = csm.generate_frame(tok, msk)
    e1.record()
    torch.cuda.synchronize()
    ms_frame = e0.elapsed_time(e1) / nfr
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=csm.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    def lin(a):  # parameters of one Llama layer's linears (attention.py / mlx_lm LlamaModel: q, k, v, o, gate, up, down)
        H, KV, hd, D, I = a["num_heads"], a["num_kv_heads"], a["head_dim"], a["hidden"], a["intermediate"]
        return D * (H + 2 * KV) * hd + H * hd * D + 3 * D * I

    bb, dc, V, D, Dd = cfg["backbone"], cfg["decoder"], cfg["audio_vocab_size"], cfg["backbone"]["hidden"], cfg["decoder"]["hidden"]
    # algorithmic bytes of one frame: every Linear matrix it multiplies by, once per use, in the checkpoint's bf16 (sesame.py:349-395: the
    # backbone once, codebook0_head, then 31 x {projection, the 4-layer depth decoder, audio_head[i-1]}); activations / KV are negligible
    params_frame = bb["num_layers"] * lin(bb) + D * V + (n - 1) * (D * Dd + dc["num_layers"] * lin(dc) + Dd * V)
    bytes_frame = 2.0 * params_frame
    audio_s = world * B * FRAMES * 0.08 * args.steps
    out = {
        "metric": "audio-sec/sec (xRT), CSM-1B + Mimi, batch=8 streams, reference-audio prompt, end to end",
        "value": audio_s / dt, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16 weights, f32 arithmetic (frame generator); fp32 Mimi.encode, bf16 Mimi.decode",
        "data": "synthetic (seeded random-init CSM-1B and Mimi weights, noise reference audio, random token ids, greedy frames, EOS ignored)",
        "config": {"workload": f"CSM-1B (llama-1B backbone + llama-100M depth decoder) + Mimi codec: B={B} streams/GPU, prompt = {REF_S:.0f} s reference "
                               f"audio ({S - N_TEXT} Mimi frames incl. the EOS frame) + {N_TEXT} text ids = {S} positions, {FRAMES} frames ({FRAMES * 0.08:.0f} s) "
                               f"generated per stream, Mimi.decode included; replicas x{world}" + (f"; {NS} such jobs in flight per GPU (own stream, thread, KV caches; one copy of the weights)" if NS > 1 else ""),
                   "global_batch": B * world, "parallelism": f"replicas x{world}"},
        "ms_per_frame": ms_frame, "frames_per_step": FRAMES, "jobs_in_flight": NS,
        "ms_per_step_one_job_in_flight": (dt_one / args.steps * 1e3) if dt_one is not None else None,
        "roofline": {"bound": "hbm", "kernel": "CSM frame step (six launches per Llama layer, Linears on gemvm_kernel (matrix cores); 16 + 31 x 4 layer passes)",
                     "achieved": bytes_frame / (ms_frame * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": bytes_frame / (ms_frame * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                     "bytes_per_frame": bytes_frame, "note": "algorithmic bytes = every Linear matrix a frame multiplies by, in bf16; one launch = one frame step "
                     "(hipGraph replay), duration from HIP events on the launch stream over 50 frames"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_csm(cfg, w, mcfg, mw, res, S)
    return out


def cpu_baseline_csm(cfg, w, mcfg, mw, res, S):
    """The CPU oracles (kind "port") on a bounded sample of the same workload: ONE stream, a prompt block of the same length, 2 frames, and
    the Mimi decode of those frames."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import csm_oracle as CO
    import mimi_oracle as MO

    n = cfg["audio_num_codebooks"]
    rng = np.random.default_rng(0)
    wb = {k: torch.tensor(np.asarray(v, np.float32)).to(torch.bfloat16).float().numpy() for k, v in w.items()}
    orc = CO.CsmOracle(wb, cfg)
    tok = np.zeros((1, S, n + 1), np.int64)
    msk = np.zeros((1, S, n + 1), np.float32)
    tok[:, :, -1] = rng.integers(0, cfg["text_vocab_size"], (1, S))
    msk[:, :, -1] = 1
    cores = torch.get_num_threads()
    t0 = time.time()
    frames = []
    for _ in range(2):
        c = orc.generate_frame(tok, msk)
        frames.append(c)
        tok = np.zeros((1, 1, n + 1), np.int64)
        tok[:, 0, :n] = c
        msk = np.zeros((1, 1, n + 1), np.float32)
        msk[:, 0, :n] = 1
    MO.MimiOracle(mw, mcfg).decode(np.stack(frames, 2))
    dt = time.time() - t0
    return {"value": 2 * 0.08 / dt, "unit": "audio-sec/sec", "cores": int(cores), "kind": "port",
            "sample": f"1 stream: prompt block of {S} positions + 2 frames (0.16 s of audio) + Mimi decode in {dt:.1f} s; oracle/csm_oracle.py + oracle/mimi_oracle.py, torch-CPU fp32"}


def bench_kokoro(args, rank, world, quantization_kernel=None, brief=False):
    """One configuration of the Kokoro forward: builds the engine, times K steps, (unless brief) brackets the kernels and measures the B = 1
    latency and the CPU baseline.  Returns the JSON object (rank 0; other ranks return a stub)."""
    quantization_kernel = quantization_kernel or args.quantization_kernel
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=rank, world_size=world)

    import mlx_audio_amd.params as P
    from mlx_audio_amd import _lib
    from mlx_audio_amd.engine import KokoroEngine
    from mlx_audio_amd.parallel import gather_waveforms, shard_range

    cfg = P.kokoro_config()
    w = P.synth_checkpoint(cfg, 0)
    quantization = None
    if args.quantized:
        if args.dtype != "bfloat16":
            raise SystemExit("--quantized is the bf16 + fp8 configuration")
        from mlx_audio_amd.quant import dequantize_checkpoint, quantize_checkpoint

        quantization = {"group_size": 64, "bits": 8} if quantization_kernel == "mxfp8" else None
        w = dequantize_checkpoint(quantize_checkpoint(w, 64, 8), 64, 8)  # what load_model hands the engine for an 8-bit checkpoint
        w = {k: torch.tensor(np.asarray(v, np.float32)).to(torch.bfloat16) for k, v in w.items()}  # an 8-bit checkpoint of the bf16 model
    if args.dtype == "bfloat16" and not args.quantized:
        w = {k: torch.tensor(v).to(torch.bfloat16) for k, v in w.items()}  # the checkpoint dtype of the named config
    NS = max(1, int(args.streams if args.streams is not None else 2))
    engs = [KokoroEngine(cfg, w, compute_dtype=args.dtype, quantization=quantization)]
    engs += [engs[0].new_context() for _ in range(NS - 1)]  # ONE kk_model (one copy of the weights); a kk_context + workspace + graph per stream
    eng = engs[0]
    if os.environ.get("KK_BENCH_FORCE"):  # A/B experiments only: kk_debug_force_generic flags (e.g. 64 = conv variant 5 where eligible)
        for e in engs:
            e.lib.kk_debug_force_generic(e._h, int(os.environ["KK_BENCH_FORCE"]))
    if args.quantized:
        assert eng.quantized_layers() == (6 if quantization_kernel == "mxfp8" else 0)
    dev = eng.device

    # ---- synthetic workload: global batch = world * B utterances, this rank takes its contiguous shard
    B = args.batch
    Bglob = B * world
    rng = np.random.default_rng(0)
    all_utts = [rng.integers(1, 178, N_PHONEMES).tolist() for _ in range(Bglob)]
    rows = np.load(os.path.join(ROOT, "tests", "golden", "af_heart_rows.npz"))["rows"]
    all_ref = rows[rng.integers(0, rows.shape[0], Bglob)].astype(np.float32)
    lo, hi = shard_range(Bglob, world, rank)
    utts, ref_np = all_utts[lo:hi], all_ref[lo:hi]
    ids, lens, Tmax = eng.pack_ids(utts)
    ref_s = torch.tensor(ref_np, device=dev)
    speed = torch.ones(B, device=dev)
    forced = torch.full((B, Tmax), FRAMES_PER_TOKEN, dtype=torch.int32, device=dev)
    Fmax = FRAMES_PER_TOKEN * Tmax
    # two waveform buffers: with N > 1 the gather of batch i (the path's one exchange step: every shard's waveforms land on rank 0)
    # runs on RCCL's stream while batch i+1 is synthesised into the other buffer
    nbuf = max(NS, 2 if world > 1 else 1)
    wavs = [torch.empty((B, 600 * Fmax), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    gathered = [torch.empty((Bglob, 600 * Fmax), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None for _ in range(nbuf)]
    pending = [None] * nbuf
    for e in engs:
        e.workspace(B, Tmax, Fmax)
    # NS batches in flight: step i runs on stream i % NS with engine i % NS (its own workspace and captured graph) into waveform buffer i % nbuf
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(NS - 1)]

    def step(i, single=False):
        j = i % nbuf
        k = 0 if single else i % NS
        with torch.cuda.stream(streams[k]):
            if pending[j] is not None:
                pending[j].wait()  # the batch that used this buffer last has left it
                pending[j] = None
            engs[k].forward(ids, lens, ref_s, speed, Fmax, forced_dur=forced, noise_mode=_lib.NOISE_PHILOX, seed=1000 + i, out=wavs[j])
            if world > 1:
                pending[j] = gather_waveforms(wavs[j], gathered[j], dist, async_op=True)

    def drain():
        for j in range(nbuf):
            if pending[j] is not None:
                pending[j].wait()
                pending[j] = None

    def barrier():
        drain()  # every exchange of the region has completed before the clock is read
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region runs the product configuration: graph replay of the forward (one hipGraphLaunch per step; the first
    # two warm-up steps are the eager run and the capture).  The per-kernel HIP-event brackets need eager launches, so the
    # roofline figures come from a SECOND, separately timed pass of the same K steps (its wall time is reported too).
    use_graph = not args.no_graph
    if use_graph:
        for e in engs:
            e.set_graph_mode(True)
    nwarm = max(args.warmup, 2 * nbuf) if use_graph else args.warmup  # eager run + capture per engine / output buffer happen before the clock starts
    for i in range(nwarm):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(nwarm + i)
    barrier()
    dt = time.perf_counter() - t0
    dt_one = None
    if NS > 1 and not brief:  # the same K steps strictly one after the other (engine 0, one stream), for reference
        for i in range(2 * nbuf):
            step(i, single=True)  # (engine 0 captures its graph for every output buffer)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(i, single=True)
        barrier()
        dt_one = time.perf_counter() - t1
    prof, dt_prof = None, None
    if not args.no_profile and not brief:
        eng.profile_begin()  # forwards with an open profile run eagerly
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(nwarm + args.steps + i, single=True)
        barrier()
        dt_prof = time.perf_counter() - t1
        prof = eng.profile_end()

    # p50 per-utterance latency (second half of BASELINE.json's metric): B = 1, same shapes, after the timed region
    p50_ms = None
    if rank == 0 and not args.no_latency and not brief:
        i1, l1, T1 = eng.pack_ids(utts[:1])
        r1, s1, f1 = ref_s[:1].contiguous(), speed[:1].contiguous(), forced[:1].contiguous()
        w1 = torch.empty((1, 600 * Fmax), dtype=torch.float32, device=dev)
        lat = []
        for i in range(12):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.forward(i1, l1, r1, s1, Fmax, forced_dur=f1, noise_mode=_lib.NOISE_PHILOX, seed=7 + i, out=w1)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        p50_ms = float(np.median(lat[2:]))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    audio_s = Bglob * (600 * Fmax) / SR * args.steps
    dtype_tag = ("bf16" if not args.quantized else ("bf16+fp8(e4m3, quantised linears)" if quantization_kernel == "mxfp8" else
                 "bf16 (8-bit affine weights dequantised: scale*q+bias)")) if args.dtype == "bfloat16" else "f32"
    out = {
        "metric": f"audio-sec/sec (xRT), Kokoro-82M{' 8-bit quantised' if args.quantized else ''} batch={B} fixed 128-phoneme utterances per GPU",
        "value": audio_s / dt,
        "unit": "audio-sec/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": nwarm,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_tag,
        "data": "synthetic (seeded random-init checkpoint of the Kokoro-82M architecture, random phoneme ids, real af_heart style rows, Philox noise)",
        "config": {"workload": f"Kokoro-82M {dtype_tag}, batch={B}/GPU fixed {N_PHONEMES}-phoneme utterances (T={Tmax}, F={Fmax}, {600 * Fmax} samples = "
                               f"{600 * Fmax / SR:.2f} s each), utterance-sharded over {world} GPU(s)" + (" + RCCL gather to rank 0" if world > 1 else ""),
                   "global_batch": Bglob, "parallelism": f"utterance-shard x{world}" + (f", {NS} batches in flight per GPU" if NS > 1 else "")},
    }
    if rank == 0:
        out["batches_in_flight"] = NS
        if dt_one is not None:
            out["ms_per_step_one_batch_in_flight"] = dt_one / args.steps * 1e3  # the same K steps strictly one after the other
        if p50_ms is not None:
            out["p50_latency_ms_b1"] = p50_ms
            out["p50_latency_note"] = "one 128-phoneme utterance (16.25 s audio), kk_forward + sync, median of 10 after 2 warm-ups"
        pmc = None
        try:
            # PMC byte counts were taken at one batch size per configuration: B = 32 for the bf16 / fp32 entries, B = 64 for config 5
            key, pmc_batch = ("bfloat16+fp8 (config 5, B=64, r01_i)", 64) if (args.quantized and quantization_kernel == "mxfp8") else (args.dtype, 32)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json"))).get(key) if B == pmc_batch else None
        except Exception:
            pmc = None
        if prof is not None:
            # dominant kernel = the convolution family (carries ~99 % of the algorithmic FLOPs)
            conv = prof["conv_mfma"] if prof["conv_mfma"]["ms"] > prof["conv_generic"]["ms"] else prof["conv_generic"]
            peak = PEAK_TFLOPS["bf16" if conv is prof["conv_mfma"] else "f32"]
            ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12 if conv["ms"] > 0 else 0.0
            traffic = None
            if pmc and conv is prof["conv_mfma"] and "conv_mfma" in pmc:
                c = pmc["conv_mfma"]  # separate --pmc passes, see profiles/r02_pmc_traffic.json (read side reported raw)
                traffic = (c["fetch_raw_bytes_per_step"] + c["write_bytes_per_step"]) / c["launches_per_step"]
            if prof.get("linear_mxfp8", {}).get("ms", 0) > 0:
                q = prof["linear_mxfp8"]  # activation pre-pass + block-scaled fp8 product, bracketed together
                qa = q["flops"] / (q["ms"] * 1e-3) / 1e12
                qt = None
                if pmc and "linear_mxfp8" in pmc:  # product + activation pre-pass, like the bracket
                    qt = sum(pmc[k]["fetch_raw_bytes_per_launch"] + pmc[k]["write_bytes_per_launch"] for k in ("linear_mxfp8", "mxfp8_quant_rows"))
                out["roofline_linear_mxfp8"] = {"bound": "mfma", "achieved": qa, "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s", "frac": qa / PEAK_FP8_TFLOPS,
                                                "traffic": qt, "launches_per_step": q["launches"] / args.steps, "ms_per_step": q["ms"] / args.steps}
            out["roofline"] = {"bound": "mfma", "kernel": "conv_mfma" if conv is prof["conv_mfma"] else "conv_generic (fp32 VALU implicit GEMM)",
                               "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                               "launches_per_step": conv["launches"] / args.steps, "ms_per_step": conv["ms"] / args.steps}
            ih = prof["istft_head"]
            if ih["ms"] > 0:
                gbs = ih["bytes"] / (ih["ms"] * 1e-3) / 1e9
                tr = None
                if pmc and "istft_head" in pmc:
                    tr = pmc["istft_head"]["fetch_corrected_bytes_per_launch"] + pmc["istft_head"]["write_bytes_per_launch"]
                out["roofline_istft_head"] = {"bound": "hbm", "kernel": "conv_post_istft (conv_post + iSTFT + overlap-add fused; stand-alone iSTFT head in fp32 mode)", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                              "traffic": tr, "us_per_launch": ih["ms"] / ih["launches"] * 1e3}
            out["launch_mode"] = ("hipGraph replay" if use_graph else "eager") + (f", {NS} batches in flight on {NS} HIP streams" if NS > 1 else "")
            out["ms_per_step_eager_profiled_pass"] = dt_prof / args.steps * 1e3
            tot = sum(v["ms"] for v in prof.values())
            out["kernel_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in prof.items() if v["launches"]}
            out["kernel_ms_per_step"]["_sum_bracketed"] = round(tot / args.steps, 3)
        if world == 1 and not args.no_cpu_baseline and not brief:
            out["cpu_baseline"] = cpu_baseline(cfg, P.synth_checkpoint(cfg, 0), utts[0], ref_np[0:1])
    del eng, engs
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
