#!/usr/bin/env python3
"""bench.py — throughput of the MI355X-native VITS2 hot path (BASELINE.json metric: audio
samples/sec + real-time factor).

    python bench.py --gpus 1 --steps 50 --warmup 5            # default workload = BASELINE configs[1] (C2)
    python bench.py --gpus N ...                              # launches its own N ranks (one process per GPU, host-side barrier)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # the driver's form: same ranks
Ranks only meet at a host-side barrier and a max over their timings (torch.distributed, gloo on CPU tensors: no RCCL on any path,
there is nothing to exchange -- SURVEY.md 8e).  With N > 1 the default line still leads with BASELINE's metric on configs[1] per
replica (weak scaling) and adds "batch256_sharded": BASELINE configs[3], the 256-request list dealt to the N ranks by
vosk_tts_amd.batching.plan_shards, value = all 256 requests' samples / slowest rank (STRONG scaling: the one place where length
imbalance between shards can cost anything), plus "multi_device_synth": the in-process front door on the same list.

A "step" is one full forward of SynthesizerTrn.infer (text encoder -> stochastic duration predictor ->
length regulator -> flow -> decoder) over one synthetic batch whose inputs are already resident in HBM.
Durations are pinned to 3 frames/token so the work is fixed, but the duration predictor IS
executed (its logw is simply not used), so every row of §8a is inside the timed region.  The timed region
is repeated in blocks of --steps steps until it spans at least --min-seconds; ms_per_step is the median
block.  Each rank is an independent replica (no collective on the data path) and `value`
is the whole-job aggregate.  Next to this device-resident figure the default line carries "host_api": the
drop-in path a vosk-tts user calls (ids on the host -> int16 PCM on the host, free-running durations, a
fresh seed per request; what vosk_tts/synth.py:122-131 brackets).  Weights are seeded synthetic tensors of the ru-0.9-multi-shaped
MB-iSTFT-VITS2 architecture (the real checkpoint cannot be downloaded offline).

Workloads (SURVEY.md §8 sizes; durations pinned to 3 frames/token so the work is fixed):
  c2  B=1,  T_x=50            -> T_y=150,  38 400 samples (1.74 s)      [default: BASELINE configs[1]]
  c3  B=32, T_x in [20,200]   -> padded to max, ragged masks             [configs[2], fp32 throughout]
  c4  B=256 ragged, sharded over the ranks by vosk_tts_amd.batching.plan_shards (32 per GPU at 8)  [configs[3]]
  c5  B=1,  T_x=2000          -> T_y=6000, 69.7 s of audio               [configs[4]]
The default run (c2) also measures c3 and reports it under "batch32" in the same JSON line, because
BASELINE.json quotes the metric at "batch=1 and 32".
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 matrix peak (same guide; the headline figures with 2:1 sparsity are not used)
SAMPLE_RATE = 22050
PROFILE_ROUND = "r6"  # prefix of the committed rocprofv3 / PMC summaries under profiles/ quoted beside the live figures
# the reference's own PyTorch modules (SynthesizerTrn.infer, eager, fp32) on the survey container's 8 vCPU
# Xeon @ 2.1 GHz for the c2 shape (50 tokens -> 150 frames, durations pinned): the only executable form of the reference.
REFERENCE_PYTORCH_CPU = {"x_realtime": [7.0, 7.5], "samples_per_s": 1.6e5, "cores": 8, "infer_s": [0.23, 0.25],
                         "what": "reference PyTorch modules, eager CPU, torch.set_num_threads(8), B=1 50 tokens -> 150 frames",
                         "source": "BASELINE.md section 3 (measured in the survey container, not on the GPU box)"}


GC_NOTE = ("heap frozen before every timed region (gc.collect + gc.freeze: what Model.warmup(freeze_gc=True) does for a server), collector "
           "enabled during the steps; a default server without that call sees 25-60 ms older-generation pauses (profiles/r5_m2_gc.txt)")


def settle_gc():
    """Called in front of every timed region.  CPython's cyclic collector stops THIS process for 25-60 ms per older-generation pass --
    the heap of a process that imported torch and generated synthetic weights is a few million objects -- which is longer than most
    timed regions here (measured with tools/m2_steps.py, profiles/r5_m2_gc.txt: calls 12 and 105 of the 4.2 ms m2 loop took 25 and
    42-61 ms, none with the collector off; where the passes fall moves with every unrelated allocation).  Everything allocated so far is
    collected once and moved to the permanent generation -- gc.freeze(), what a pre-fork server does after loading its models -- and the
    collector STAYS ENABLED during the timed steps: it only has the steps' own garbage to look at."""
    gc.collect()
    gc.freeze()


def make_workload(name, rng, rank=0, world=1):
    if name == "c4":
        from vosk_tts_amd.batching import plan_shards

        all_len = rng.integers(20, 201, size=256).astype(np.int64)
        mine = plan_shards(all_len, world, max_batch=(256 + world - 1) // world)[rank]
        lengths = all_len[mine]
        B, Tx = len(lengths), int(lengths.max())
        ids = rng.integers(1, 62, size=(B, Tx)).astype(np.int64)
        dur = np.where(np.arange(Tx)[None] < lengths[:, None], 3, 0).astype(np.int32)
        return ids, lengths, dur
    if name == "c2":
        lengths = np.array([50], np.int64)
    elif name == "c3":
        lengths = rng.integers(20, 201, size=32).astype(np.int64)
    elif name == "c5":
        lengths = np.array([2000], np.int64)
    elif name == "c1":
        lengths = np.array([10], np.int64)
    elif name == "s8":  # a coalesced burst of a thread-pool server: 8 requests of ~47 tokens (host_api.concurrent's text)
        lengths = np.full(8, 47, np.int64)
    elif name == "s16":
        lengths = np.full(16, 47, np.int64)
    elif name[0] == "u" and name[1:].isdigit():  # one utterance of N tokens (kernel-selection sweeps between c2 and c5)
        lengths = np.array([int(name[1:])], np.int64)
    else:
        raise ValueError(name)
    B, Tx = len(lengths), int(lengths.max())
    ids = rng.integers(1, 62, size=(B, Tx)).astype(np.int64)
    dur = np.where(np.arange(Tx)[None] < lengths[:, None], 3, 0).astype(np.int32)
    return ids, lengths, dur


def stts_algorithmic_flops(hp, voc_model, Tx, Ty, n_steps):
    """MatchaTTS.synthesise + vocoder, multiply-adds x 2: dp_encoder per symbol; per frame the estimator
    (in_proj, 6 DiT blocks, 3 long-skip convs, final_proj) x (1 + CFG) x n_steps plus cond_proj once per CFG branch."""
    He, Fe, Hd, Fd, NF, K = hp.enc_hidden, hp.enc_filter, hp.dec_hidden, hp.dec_filter, hp.n_feats, hp.dec_kernel
    tok = 2 * hp.bert_dim * hp.bert_proj_dim + hp.enc_layers * (2 * 4 * He * He + 2 * 2 * He * Fe * hp.enc_kernel) + 2 * He * hp.dp_out
    tok_quad = hp.enc_layers * 4 * He
    blk = 2 * 4 * Hd * Hd + 2 * 2 * Hd * Fd * K
    est = 2 * (NF + Hd) * Hd + hp.dec_layers * blk + (hp.dec_layers // 2) * 2 * (2 * Hd) * Hd * K + 2 * Hd * NF
    est_quad = hp.dec_layers * 4 * Hd
    cond = 2 * K * (hp.enc_hidden * Fd + Fd * Fd + Fd * Hd)
    nb = 2 if hp.guidance_scale > 0 else 1
    frame = nb * (n_steps * est + cond)
    return Tx * (tok + tok_quad * Tx) + Ty * (frame + nb * n_steps * est_quad * Ty) + voc_model.algorithmic_flops(1, 0, Ty)


_EPI_NAMES = {"0": "STORE", "1": "GATE", "2": "RESSKIP", "3": "COUPLE"}


def rocprof_avg_us(csv_path, kernel):
    """AverageNs of `kernel` (the engine's name, e.g. conv_mfma_ks_kernel<1,1,STORE,1>) in a rocprofv3 kernel_stats.csv, or None"""
    import csv
    import re

    if not os.path.exists(csv_path):
        return None
    with open(csv_path, newline="") as f:
        for row in csv.DictReader(f):
            m = re.match(r"(?:void )?(\w+)<([^>]*)>", row.get("Name", ""))
            if not m:
                m0 = re.match(r"(?:void )?(\w+)\(", row.get("Name", ""))
                if m0 and m0.group(1) == kernel:
                    return round(float(row["AverageNs"]) / 1e3, 2)
                continue
            args = [a.strip() for a in m.group(2).split(",")]
            epi_pos = 2 if m.group(1) == "conv_mfma_ks_kernel" else 4 if m.group(1) == "conv_mfma_kernel" else None
            if epi_pos is not None and epi_pos < len(args):
                args[epi_pos] = _EPI_NAMES.get(args[epi_pos], args[epi_pos])
            if f"{m.group(1)}<{','.join(args)}>" == kernel:
                return round(float(row["AverageNs"]) / 1e3, 2)
    return None


def bench_multistream(args, torch, rank, world, local_rank, dist, as_object=False):
    """configs[1]-shaped single utterance on the StableTTS / Matcha family: 50 symbols, 3 frames per
    symbol pinned through phone_duration_extra, 5 Euler steps with guidance, bundled HiFi-GAN V1 vocoder; host entry point
    (ids in, PCM-ready float waveform out), so the few KB of H2D and the 150 KB D2H are inside the timed region."""
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi import VitsLib
    from vosk_tts_amd.capi_stts import SttsModel

    lib = VitsLib()
    vhp = W.hifigan_v1_vocoder_hparams()
    if getattr(args, "precision", "f32") == "bf16x3":
        vhp.conv_precision = 1  # the vocoder's 256- / 128-channel ResBlock convs as split-bf16 at batch size (m3)
    vblob = W.synthetic_blob(vhp, 1234)
    hp = S.default_hparams(62, 5)
    blob = S.synthetic_blob(hp, 1234)
    voc = lib.create(vblob, local_rank)
    model = SttsModel(lib, blob, voc, local_rank)
    rng = np.random.default_rng(1234 + rank)
    scales = np.array([0.8, 1.0, 0.8], np.float32)
    batched = args.workload == "m3"
    if batched:  # configs[2] shape: 32 ragged utterances of 20..200 symbols, 3 frames per symbol
        B = 32
        lengths = rng.integers(20, 201, size=B).astype(np.int64)
        Tx = int(lengths.max())
        ids = rng.integers(1, 62, size=(B, 5, Tx)).astype(np.int64)
        pde = np.full((B, Tx), 3.0, np.float32)
        sid = np.full(B, 2, np.int64)

        def step(i):
            a, ol = model.synthesize_batch(ids, lengths, scales, sid, None, pde, seed=7 + i)
            return a, ol
    else:
        Tx = 50
        ids = rng.integers(1, 62, size=(5, Tx)).astype(np.int64)
        pde = np.full(Tx, 3.0, np.float32)

        def step(i):
            return model.synthesize(ids, scales, 2, None, pde, seed=7 + i, want_mel=False)[0]

    for i in range(args.warmup):
        step(i)
    settle_gc()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        audio = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if batched:
        audio, olen = audio
        S_ = int(olen.sum())
        Ty = int(olen.max()) // hp.hop_length
        assert np.isfinite(audio).all() and np.array_equal(olen, lengths * 3 * hp.hop_length)
        flops = sum(stts_algorithmic_flops(hp, voc, int(l), int(3 * l), hp.n_timesteps) for l in lengths)
    else:
        S_ = int(audio.shape[0])
        Ty = S_ // hp.hop_length
        assert np.isfinite(audio).all() and Ty == 3 * Tx
        flops = stts_algorithmic_flops(hp, voc, Tx, Ty, hp.n_timesteps)
    value = S_ * world * args.steps / elapsed
    if as_object:  # secondary figure inside the default line: the second model family on the same utterance shape
        ms = elapsed / args.steps * 1e3
        model.close()
        return {"value": round(value, 1), "unit": "samples/s", "ms_per_step": round(ms, 4), "x_realtime": round(S_ / SAMPLE_RATE / (ms * 1e-3), 1),
                "workload": f"m2: StableTTS/Matcha multistream graph + HiFi-GAN V1, B=1, {Tx} symbols -> T_y={Ty}, {hp.n_timesteps} Euler steps, "
                            "host entry point stts_synthesize (python bench.py --workload m2 for the full line)",
                "algorithmic_gflop": round(flops / 1e9, 3), "achieved_tflops": round(flops / (ms * 1e-3) / 1e12, 3),
                "frac_of_fp32_mfma_peak": round(flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        so = os.path.join(ROOT, "oracle", "libvits_oracle.so")
        if os.path.exists(so):
            import ctypes

            olib = VitsLib(so, "vitsref_")
            ref = SttsModel(olib, blob, olib.create(vblob))
            olib.lib.vitsref_num_threads.restype = ctypes.c_int
            n, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds and n < 20:
                c0 = time.perf_counter()
                if batched:  # bounded sample: the first utterance of the batch
                    ref.synthesize(ids[0][:, :int(lengths[0])], scales, 2, None, pde[0][:int(lengths[0])], seed=7, want_mel=False)
                else:
                    ref.synthesize(ids, scales, 2, None, pde, seed=7, want_mel=False)
                t_cpu += time.perf_counter() - c0
                n += 1
            cs = int(lengths[0]) * 3 * hp.hop_length if batched else S_
            cpu_baseline = {"value": round(cs * n / t_cpu, 1), "unit": "samples/s", "cores": int(olib.lib.vitsref_num_threads()), "kind": "port",
                            "sample": f"{n} forward(s) of {'the first utterance of the batch' if batched else 'the same utterance'} ({cs} samples) through "
                                      f"oracle/libvits_oracle.so (sttsref_synthesize, OpenMP), {t_cpu:.1f} s",
                            "x_realtime": round(cs * n / t_cpu / SAMPLE_RATE, 2)}
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        emit_line({
            "metric": "audio_samples_per_sec", "value": round(value, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if getattr(args, "precision", "f32") == "f32" else "bf16x3", "data": "synthetic", "gc": GC_NOTE, "rtf": round(ms * 1e-3 / (S_ / SAMPLE_RATE), 6), "x_realtime": round(S_ / SAMPLE_RATE / (ms * 1e-3), 1),
            "config": {"workload": f"{args.workload}: StableTTS/Matcha multistream graph (seeded synthetic weights) + bundled HiFi-GAN V1, "
                                   + (f"B=32 ragged {int(lengths.min())}..{int(lengths.max())} symbols" if batched else f"B=1, {Tx} symbols") + " x 5 streams, "
                                   f"zero BERT vectors, durations pinned 3/symbol -> T_y<={Ty}, {hp.n_timesteps} Euler steps with guidance {hp.guidance_scale:g}, "
                                   f"{S_} valid samples/step/GPU, host entry point " + ("stts_synthesize_batch" if batched else "stts_synthesize"),
                       "batch": 32 if batched else 1, "T_x": Tx, "T_y": Ty,
                       "samples_per_step_per_gpu": S_, "parallelism": f"replicas x{world} (no collective)", "hipgraph": False},
            "roofline": {"bound": "mfma", "achieved": round(flops / (ms * 1e-3) / 1e12, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                         "kernel": "whole forward (per-kernel breakdown: rocprofv3 --kernel-trace of this command)",
                         "forward": {"algorithmic_gflop": round(flops / 1e9, 3)}},
            "cpu_baseline": cpu_baseline})
    if dist is not None:
        dist.destroy_process_group()



def vits_cpu_baseline(blob, hp, ids, lengths, dur, workload, cpu_seconds):
    """TEST-INFRASTRUCTURE leg of the default run: the CPU oracle (plain C restatement of the reference arithmetic) timed on this
    box's host cores on a bounded sample of the same workload.  Needs no GPU (tests/test_host_api.py runs it on a toy model)."""
    import ctypes

    from vosk_tts_amd.capi import VitsLib

    so = os.path.join(ROOT, "oracle", "libvits_oracle.so")
    if not os.path.exists(so):
        return None
    ref_lib = VitsLib(so, "vitsref_")
    ref = ref_lib.create(blob)
    ref_lib.lib.vitsref_num_threads.restype = ctypes.c_int
    all_threads = int(ref_lib.lib.vitsref_num_threads())
    scales = np.array([0.8, 1.0, 0.8], np.float32)
    nb = min(ids.shape[0], 2)  # bounded sample: at most 2 utterances of the batch per call
    sub = (ids[:nb], lengths[:nb], dur[:nb])
    sub_samples = int(sub[2].sum()) * hp.hop_length

    def forward():
        c0 = time.perf_counter()
        ref.synthesize(sub[0], sub[1], scales, np.full(nb, 2), forced_durations=sub[2], seed=7)
        return time.perf_counter() - c0

    # thread count: a single utterance has limited parallel grain (256 output rows x a few hundred columns per conv), so all
    # hardware threads of a 128-thread host are slower than a fraction of them -- two forwards per candidate, keep the fastest
    scan = {}
    set_threads = getattr(ref_lib.lib, "vitsref_set_num_threads", None)
    if set_threads is not None:
        set_threads.restype = None
        for nt in sorted({all_threads, max(1, all_threads // 2), max(1, all_threads // 4), max(1, all_threads // 8), min(all_threads, 8)}, reverse=True):
            set_threads(ctypes.c_int(nt))
            scan[nt] = round(min(forward(), forward()), 4)  # (the first call after a change of team size pays the thread start-up)
            if scan[nt] > 8.0:
                break
        cores = min(scan, key=scan.get)
        set_threads(ctypes.c_int(cores))
    else:
        cores = all_threads
    n, t_cpu = 0, 0.0
    while t_cpu < cpu_seconds and n < 50:
        t_cpu += forward()
        n += 1
    return {"value": round(sub_samples * n / t_cpu, 1), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{n} forward(s) of {nb} utterance(s) of workload {workload} "
                      f"({sub_samples} samples each) through oracle/libvits_oracle.so (OpenMP, {cores} threads), {t_cpu:.1f} s",
            "x_realtime": round(sub_samples * n / t_cpu / SAMPLE_RATE, 2),
            "threads_available": all_threads, "seconds_per_forward_by_threads": {str(k): v for k, v in scan.items()},
            "note": "the port is a plain-C checker, not a tuned CPU implementation: the reference's own PyTorch CPU path "
                    "is the faster CPU data point (reference_pytorch_cpu); neither ratio is a statement about kernel quality",
            "reference_pytorch_cpu": REFERENCE_PYTORCH_CPU}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the same environment
    torch.distributed.run would set up), let rank 0 print the line, return the worst exit code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def dry_run(args, torch, dist, rank, world):
    """--dry-run: no GPU and no model -- every rank runs the launch / barrier / max-over-ranks / aggregation path of the real line
    on a pretend step (tests/test_multiproc.py runs it with 2 processes on CPU)."""
    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        time.sleep(0.0005)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))  # ranks differ on purpose: the line must carry the slowest one
    own = time.perf_counter() - t0
    barrier()
    el = time.perf_counter() - t0
    seen = 1
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        c = torch.tensor([1.0], dtype=torch.float64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        seen = int(c.item())
    samples = 38400
    # the per-rank fields of the real line travel the same way (all_gather of one float per rank) ...
    rank_ms, rank_persist = [round(own / args.steps * 1e3, 4)], [1.0]
    if dist is not None:
        g = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(g, torch.tensor([rank_ms[0], 1.0], dtype=torch.float64))
        rank_ms = [round(float(x[0].item()), 4) for x in g]
        rank_persist = [float(x[1].item()) for x in g]
    # ... and configs[3]'s request list is dealt to the ranks exactly as the real batch256_sharded leg deals it
    _, c4_len, _ = make_workload("c4", np.random.default_rng(1234), rank, world)
    shard = torch.tensor([float(len(c4_len))], dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(shard, op=dist.ReduceOp.SUM)
    # what the first real N-GPU run can be checked against: the work plan_shards deals to every rank (valid frames = 3 x tokens here,
    # and padded frames = shard size x its longest item, what a dense padded batch would execute), slowest rank over the mean
    own_work = torch.tensor([float(c4_len.sum()) * 3.0, float(len(c4_len)) * float(c4_len.max()) * 3.0], dtype=torch.float64)
    work = [own_work]
    if dist is not None:
        work = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(work, own_work)
    valid_fr = [float(w[0].item()) for w in work]
    padded_fr = [float(w[1].item()) for w in work]
    predicted = {"valid_frames_per_rank": [int(v) for v in valid_fr], "padded_frames_per_rank": [int(v) for v in padded_fr],
                 "imbalance_valid_max_over_mean": round(max(valid_fr) / (sum(valid_fr) / len(valid_fr)), 4),
                 "imbalance_padded_max_over_mean": round(max(padded_fr) / (sum(padded_fr) / len(padded_fr)), 4),
                 "reads": "batch256_sharded is strong scaling: value = all samples / the slowest rank; with ragged decoding a rank's time follows its "
                          "VALID frames (plus ~30 halo frames per item), so expect the real line's max(rank_ms) / mean(rank_ms) near imbalance_valid"}
    # the order of the legs that open devices: every rank but 0 lets go of its device before rank 0's in-process multi-device leg
    barrier()  # "ranks > 0 have closed their models"
    barrier()  # "rank 0 is done with all devices"
    if rank == 0:
        emit_line({"metric": "audio_samples_per_sec", "value": round(samples * world * args.steps / el, 1), "unit": "samples/s", "n_gpus": world,
                          "rank_ms": rank_ms, "rank_persistent_launches_per_forward": rank_persist, "batch256_requests_seen": int(shard.item()),
                          "batch256_predicted": predicted, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "gc": GC_NOTE, "dry_run": True, "ranks_seen": seen,
                          "launched_by": "bench.py" if os.environ.get("BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if dist is not None else "single process"),
                          "config": {"workload": "dry run: no GPU work, launch / barrier / aggregation only"}})
    if dist is not None:
        dist.destroy_process_group()


def multi_device_synth_leg(hp, lengths_all, n_devices, reps=3):
    """The in-process front door on BASELINE configs[3]'s request list: vosk_tts_amd.batching.MultiDeviceSynth (one Model replica and
    one worker thread per device, plan_shards, padded solo batches of <= 32 through vits_synthesize_pcm16, int16 back in request
    order), free-running durations.  Host-to-host: token ids on the host in, PCM on the host out."""
    import tempfile

    from vosk_tts_amd.batching import MultiDeviceSynth
    from vosk_tts_amd.toymodel import PHONEMES, write_toy_model
    from vosk_tts_amd import weights as W

    rng = np.random.default_rng(4321)
    hp_t = W.default_hparams(n_vocab=len(PHONEMES))
    hp_t.conv_precision = hp.conv_precision
    tokens = [rng.integers(1, len(PHONEMES), size=int(n)).tolist() for n in lengths_all]
    with tempfile.TemporaryDirectory() as d:
        write_toy_model(d, hp_t)
        mds = MultiDeviceSynth(model_path=d, devices=list(range(n_devices)))
        try:
            mds.synth_tokens(tokens[:2 * n_devices], speaker_ids=2)  # graphs / workspaces warm
            mds.synth_tokens(tokens, speaker_ids=2)
            settle_gc()
            times, samples = [], 0
            for _ in range(reps):
                t0 = time.perf_counter()
                pcm = mds.synth_tokens(tokens, speaker_ids=2)
                times.append(time.perf_counter() - t0)
                samples = int(sum(len(p) for p in pcm))
        finally:
            mds.close()
    el = float(np.median(times))
    return {"requests": len(tokens), "devices": n_devices, "ms": round(el * 1e3, 2), "value": round(samples / el, 1), "unit": "samples/s",
            "x_realtime": round(samples / SAMPLE_RATE / el, 1), "samples": samples,
            "what": "MultiDeviceSynth.synth_tokens: 256 requests of 20..200 tokens, free-running durations, solo batches of <= 32 per device, "
                    "token ids on the host -> int16 PCM on the host, median of %d" % reps}

def streaming_leg(model, ids, lengths, dur, chunk=128, reps=5, warm=4):
    """configs[4]: chunked streaming through the host API (vits_stream_*): time to first audio on the host and total time for all
    chunks, next to the one-shot host call (all three include H2D of ids and D2H of audio)."""
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    ttfa, total, oneshot, warm_ttfa = [], [], [], []
    Ty = int(dur[:1].sum())
    settle_gc()
    for it in range(reps + warm):
        t0 = time.perf_counter()
        g = model.stream(ids[:1], sc, 2, chunk_frames=chunk, forced_durations=dur[:1], seed=7)
        first = next(g)
        t1 = time.perf_counter()
        n = len(first) + sum(len(c) for c in g)
        t2 = time.perf_counter()
        c0 = time.perf_counter()
        a, _ = model.synthesize(ids[:1], lengths[:1], sc, [2], forced_durations=dur[:1], seed=7)
        c1 = time.perf_counter()
        assert n == a.shape[1]
        # The first iterations are warm-up: the stream runs the eager stage path on a pooled session and alternates here with the
        # graph-replayed one-shot call of the same size; the first opens pay the workspace allocations (11 + 28 ms at this size) and the
        # first open AFTER the first graph-replayed call pays ~20 ms once more in its first eager launches (tools/stream_warm_probe.py:
        # VitsSession.warmup(stream_chunk_frames=...) takes both off a server's first requests); an open then costs 5 ms (round 6: with one warm-up iteration the median of five flipped between 5 and 26 ms from run to
        # run).  `warmup_ms` keeps those iterations visible.
        if it >= warm:
            ttfa.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3); oneshot.append((c1 - c0) * 1e3)
        else:
            warm_ttfa.append(round((t1 - t0) * 1e3, 2))
    audio_s = n / SAMPLE_RATE
    return {"workload": f"c5: B=1 T_x={ids.shape[1]} -> T_y={Ty} ({audio_s:.1f} s of audio), durations pinned 3/token, fp32, host API",
            "chunk_frames": chunk, "chunk_sec": round(chunk * 256 / SAMPLE_RATE, 3), "chunks": -(-Ty // chunk),
            "time_to_first_audio_ms": round(float(np.median(ttfa)), 3), "all_chunks_ms": round(float(np.median(total)), 3),
            "one_shot_host_call_ms": round(float(np.median(oneshot)), 3),
            "time_to_first_audio_ms_all": [round(v, 2) for v in ttfa], "time_to_first_audio_ms_warmup_iterations": warm_ttfa,
            "x_realtime_one_shot": round(audio_s / (float(np.median(oneshot)) * 1e-3), 1),
            "x_realtime_streamed": round(audio_s / (float(np.median(total)) * 1e-3), 1),
            "note": "host API incl. H2D/D2H; acoustic half once over the utterance; first chunk decoded alone, then one 8-chunk window per decode, double-buffered against the chunk copies"}


def concurrency_leg(hp, device, seconds=0.4):
    """The reference's serving shape (server/tts_server.py:35-57): ONE Synth shared by a thread pool, every thread calling
    synth_audio for one utterance.  1 / 4 / 16 Python threads on one Model (text in -> int16 PCM out, free-running durations, a
    fresh seed per request): requests/s, latency percentiles, how the request coalescer batched them and how many persistent
    launches ran (2 per single-utterance call that took the persistent path)."""
    import ctypes
    import tempfile
    import threading

    from vosk_tts_amd import Model, Synth
    from vosk_tts_amd import weights as W
    from vosk_tts_amd.toymodel import PHONEMES, write_toy_model

    hp_t = W.default_hparams(n_vocab=len(PHONEMES))
    hp_t.conv_precision = hp.conv_precision
    text = "привет мир привет мир."
    out = {"text_tokens": None, "what": "N threads x Synth.synth_audio(text) on ONE Model (server/tts_server.py shape); text -> int16 PCM on the host, "
                                        "free-running durations, fresh seed per request"}
    with tempfile.TemporaryDirectory() as d:
        write_toy_model(d, hp_t)
        model = Model(model_path=d, device=device)
        synth = Synth(model)
        sess = model.onnx
        out["text_tokens"] = len(synth.g2p_noembed(synth.normalize(text)))
        lib = sess._lib.lib
        lib.vits_debug_persist_runs.restype = ctypes.c_int
        lib.vits_debug_persist_runs.argtypes = [ctypes.c_void_p]
        runs = lambda: int(lib.vits_debug_persist_runs(sess._model._h))

        def leg(n_threads, coalesce, inflight=None):
            co = sess.coalescer
            if not coalesce:
                sess.coalescer = None
            keep_inflight = co.max_inflight if co is not None else 1
            if co is not None and inflight is not None:
                co.max_inflight = inflight
            try:
                for _ in range(4):
                    synth.synth_audio(text, speaker_id=2)
                if n_threads > 1:  # the same traffic untimed first: every batch-size / length bucket it produces gets its workspace and graphs
                    wstop = time.perf_counter() + 0.5

                    def warm():
                        while time.perf_counter() < wstop:
                            synth.synth_audio(text, speaker_id=2)

                    wt = [threading.Thread(target=warm) for _ in range(n_threads)]
                    [t.start() for t in wt]
                    [t.join() for t in wt]
                lat, samples = [[] for _ in range(n_threads)], [0] * n_threads
                settle_gc()
                stop = time.perf_counter() + seconds
                c0 = (co.calls, co.requests) if co is not None else (0, 0)
                r0 = runs()

                def worker(k):
                    while time.perf_counter() < stop:
                        t0 = time.perf_counter()
                        pcm = synth.synth_audio(text, speaker_id=2)
                        lat[k].append(time.perf_counter() - t0)
                        samples[k] += int(pcm.shape[-1])

                t_all = time.perf_counter()
                th = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                el = time.perf_counter() - t_all
                allat = np.array([x for l in lat for x in l])
                res = {"threads": n_threads, "requests": int(allat.size), "requests_per_s": round(allat.size / el, 1),
                       "ms_p50": round(float(np.percentile(allat, 50)) * 1e3, 3), "ms_p90": round(float(np.percentile(allat, 90)) * 1e3, 3),
                       "samples_per_s": round(sum(samples) / el, 1), "x_realtime": round(sum(samples) / SAMPLE_RATE / el, 1),
                       "persistent_launches": runs() - r0}
                if coalesce and co is not None:
                    calls, reqs = co.calls - c0[0], co.requests - c0[1]
                    res.update({"engine_calls": calls, "mean_batch": round(reqs / max(calls, 1), 2), "largest_batch": co.largest})
                return res
            finally:
                sess.coalescer = co
                if co is not None:
                    co.max_inflight = keep_inflight

        threads = tuple(int(v) for v in os.environ.get("BENCH_THREADS", "1,4,16").split(","))  # (tools: other client counts; the line's shape is 1 / 4 / 16)
        out["coalesced"] = [leg(n, True) for n in threads]
        out["uncoalesced_16_threads"] = leg(16, False)
        if os.environ.get("BENCH_COALESCE_SWEEP"):  # tools: engine calls allowed in flight at once
            out["inflight_sweep_16_threads"] = {str(k): leg(16, True, k) for k in (2, 3, 4, 8)}
        one = out["coalesced"][0]["requests_per_s"]
        out["speedup_16_threads_over_1"] = round(out["coalesced"][-1]["requests_per_s"] / max(one, 1e-9), 2)
        sess.close()
    return out


def bert_voice_leg(hp, device, seconds=0.4):
    """What a request costs on a BERT-conditioned VITS voice (vosk_tts/synth.py:25-44,88-99): per request the WordPiece tokenizer and the
    BERT encoder run in front of the synthesis (get_word_bert: 12-layer BERT-base geometry here, hidden_states[-3] -> 10 layers executed,
    on the device through stts_bert_encode), then g2p fans the word vectors out to the phonemes and the bert [1,768,T] feed goes through
    enc_p.bert_proj.  One thread, text -> int16 PCM on the host, free-running durations; `bert_ms` is get_word_bert alone."""
    import tempfile

    from vosk_tts_amd import Model, Synth
    from vosk_tts_amd import weights as W
    from vosk_tts_amd.toymodel import PHONEMES, write_bert_dir, write_toy_model

    hp_t = W.default_hparams(n_vocab=len(PHONEMES))
    hp_t.conv_precision = hp.conv_precision
    text = "привет мир привет мир."
    with tempfile.TemporaryDirectory() as d:
        write_toy_model(d, hp_t, bert=True)
        write_bert_dir(os.path.join(d, "bert"), 1234, n_layers=12)  # BERT-base depth (the toy default is 4 layers)
        model = Model(model_path=d, device=device)
        synth = Synth(model)
        for _ in range(6):
            synth.synth_audio(text, speaker_id=2)
        lat, bert_lat, samples = [], [], 0
        settle_gc()
        stop = time.perf_counter() + seconds
        while time.perf_counter() < stop or len(lat) < 30:
            t0 = time.perf_counter()
            pcm = synth.synth_audio(text, speaker_id=2)
            lat.append(time.perf_counter() - t0)
            samples += int(pcm.shape[-1])
        for _ in range(30):
            t0 = time.perf_counter()
            rows = synth.get_word_bert(text)
            bert_lat.append(time.perf_counter() - t0)
        n_tok = len(synth.g2p(text, rows)[0])
        model.onnx.close()
    tot = float(np.sum(lat))
    return {"requests": len(lat), "tokens": n_tok, "bert_layers_run": 10, "ms_median": round(float(np.median(lat)) * 1e3, 4), "ms_p90": round(float(np.percentile(lat, 90)) * 1e3, 4),
            "bert_ms_median": round(float(np.median(bert_lat)) * 1e3, 4), "requests_per_s": round(len(lat) / tot, 1),
            "x_realtime": round(samples / SAMPLE_RATE / tot, 1),
            "what": "Synth.synth_audio on a BERT-conditioned VITS voice (synthetic weights, BERT-base geometry): tokenizer + BERT encoder + g2p + synthesis per request, one thread"}


_JSON_FD = None


def emit_line(obj):
    """the bench line: to the process's ORIGINAL stdout (main() points fd 1 at stderr so that library chatter cannot precede it)"""
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def _workload_name(v):
    import argparse
    if v in ("c1", "c2", "c3", "c4", "c5", "m2", "m3", "s8", "s16") or (v[:1] == "u" and v[1:].isdigit() and 1 <= int(v[1:]) <= 4000):
        return v
    raise argparse.ArgumentTypeError("c1..c5, m2, m3, s8, s16 or u<N> (one utterance of N tokens)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", type=_workload_name,
                    help="c1..c5: BASELINE configs on the VITS2 graph; m2 / m3: the configs[1] / configs[2] shapes on the StableTTS (multistream) family")
    ap.add_argument("--no-batch32", action="store_true", help="skip the extra c3 (batch=32) measurement of the default run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget for the CPU-oracle baseline leg")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="the timed region repeats blocks of --steps steps until it spans this long")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"],
                    help="bf16x3: model created with hparams.conv_precision = 1 (split-bf16 decoder ResBlock convs at batch size)")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-to-host drop-in path leg of the default run")
    ap.add_argument("--no-extras", action="store_true", help="skip the streaming (c5) and concurrency legs of the default run")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise launch, barrier and aggregation only (CPU test of the multi-process path)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)  # python bench.py --gpus N: this process becomes the launcher of N ranks
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)

    # stdout carries ONE JSON line and nothing else: native libraries write there too (c10d's "[Gloo] Rank 0 is connected to ..." at
    # rendezvous, for one), so from here on file descriptor 1 points at stderr and the line is written to the saved descriptor
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)

    import torch

    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):  # a rank of a multi-process run (ours or torchrun's)
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)  # host-side barrier + max: no RCCL
        dist = dist_mod
    if args.dry_run:
        return dry_run(args, torch, dist, rank, world)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # (more ranks than devices: replicas share a device)
    torch.cuda.set_device(local_rank)

    from vosk_tts_amd import weights as W
    from vosk_tts_amd.capi import VitsDeviceSession, VitsLib

    if args.workload in ("m2", "m3"):
        return bench_multistream(args, torch, rank, world, local_rank, dist)

    hp = W.default_hparams()
    if args.precision == "bf16x3":
        hp.conv_precision = 1
    blob = W.synthetic_blob(hp, 1234)
    lib = VitsLib()
    model = lib.create(blob, local_rank)

    def measure(wname, steps, warmup, min_seconds, model=model):
        rng = np.random.default_rng(1234)
        ids, lengths, dur = make_workload(wname, rng, rank, world)
        B, Tx = ids.shape
        Ty = int(dur.sum(1).max())
        S = Ty * hp.hop_length
        valid_samples = int(dur.sum()) * hp.hop_length
        scales = np.array([0.8, 1.0, 0.8], np.float32)  # runtime defaults (vosk_tts/synth.py:50-54)
        dev = torch.device("cuda", local_rank)
        d_ids = torch.from_numpy(ids).to(dev)
        d_len = torch.from_numpy(lengths).to(dev)
        d_sid = torch.full((B,), 2, dtype=torch.int64, device=dev)
        d_dur = torch.from_numpy(dur).to(dev)
        d_audio = torch.empty((B, S), dtype=torch.float32, device=dev)
        sess = VitsDeviceSession(model, B, Tx, Ty)
        sess.set_options(use_graph=not args.no_graph, profile=False)
        sess.set_sdp_always(True)  # durations pinned for fixed work, duration predictor still executed (rows a6-a9)

        def step():
            sess.synthesize_device(d_ids.data_ptr(), d_len.data_ptr(), B, Tx, scales, d_sid.data_ptr(), d_dur.data_ptr(), Ty, 7,
                                   d_audio.data_ptr(), S)

        def barrier():
            if dist is not None:
                dist.barrier()

        import ctypes
        _runs_fn = model.lib.lib.vits_debug_persist_runs
        _runs_fn.restype = ctypes.c_int
        _runs_fn.argtypes = [ctypes.c_void_p]
        persist_runs = lambda: int(_runs_fn(model._h))
        for _ in range(warmup):
            step()
        sess.sync()
        persist_r0, persist_steps = persist_runs(), 0

        def timed_block():
            settle_gc()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            sess.sync()  # waits for the session stream and surfaces any deferred device-side error
            torch.cuda.synchronize()
            barrier()
            el = time.perf_counter() - t0
            own_times.append(el)
            if dist is not None:
                t = torch.tensor([el], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el

        # EXACTLY `steps` steps per block; blocks repeat until the timed region spans min_seconds (all ranks agree: the
        # decision uses the max-over-ranks times), ms_per_step is the median block
        own_times = []
        blocks = [timed_block()]
        while sum(blocks) < min_seconds and len(blocks) < 2000:
            blocks.append(timed_block())
        elapsed = float(np.median(blocks))
        timed_region_s = float(sum(blocks))
        graph_nodes = sess.graph_nodes()
        # did the single-utterance persistent programs run?  (one owner per device -- a token inside the process, a lock file across
        # processes: a second process on the device silently takes the launch path, ~1.5 x slower at c2 -- so the line says which it was)
        persistent_per_step = (persist_runs() - persist_r0) / max(steps * len(blocks), 1)
        if B == 1 and persistent_per_step == 0 and not os.environ.get("VITS_NO_PERSIST"):
            print(f"bench.py: rank {rank}: no persistent-program launches in the timed region (another process owns this device's programs, "
                  "or the shape is not eligible): this is the launch path", file=sys.stderr)
        finite_checked = not os.environ.get("BENCH_SKIP_FINITE_CHECK")  # (tools/ A/B builds with garbage results set it; recorded in the line)
        assert not finite_checked or torch.isfinite(d_audio).all().item(), "non-finite audio"

        ms_per_step = elapsed / steps * 1e3
        job_samples = valid_samples * world
        rank_ms = [float(np.median(own_times)) / steps * 1e3]
        rank_persist = [round(persistent_per_step, 3)]
        if dist is not None:
            tt = torch.tensor([float(valid_samples), 1.0], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            job_samples = int(tt[0].item())
            assert int(tt[1].item()) == world
            g = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(g, torch.tensor([rank_ms[0], persistent_per_step], dtype=torch.float64))
            rank_ms = [round(float(x[0].item()), 4) for x in g]
            rank_persist = [round(float(x[1].item()), 3) for x in g]
        total_samples = job_samples * steps
        value = total_samples / elapsed
        audio_sec_per_step = job_samples / SAMPLE_RATE
        rtf = (elapsed / steps) / audio_sec_per_step

        # ---- per-kernel device time with HIP events on the session stream (eager, profiled forwards)
        # useful work: every item at its OWN lengths (padding columns are not algorithmic work)
        ylens = dur.sum(1)
        flops_fwd = sum(model.algorithmic_flops(1, int(lengths[i]), int(ylens[i])) for i in range(B))
        # the engine's per-launch FLOP tally assumes dense [B, T] tensors; the decoder families are scaled to the VALID frames of
        # every item (round-5 review: the earlier min(T_y, len + 32) factor counted halo columns the per-layer ragged limits of
        # DESIGN.md §5 no longer execute, and halo / tile-padding columns are not algorithmic work in the first place)
        dec_exec_frac = float(ylens.sum()) / float(B * Ty) if B > 1 else 1.0
        sess.set_options(use_graph=False, profile=True)
        nprof = 3
        for _ in range(nprof):
            step()
        rep = sess.profile_report()  # {(op, kernel instantiation): (launches, ms, flops)}
        # masked stages of a ragged batch run on valid columns only: scale their dense tallies by the valid fraction
        enc_frac = float(lengths.sum()) / float(B * Tx)
        flow_frac = float(ylens.sum()) / float(B * Ty)

        def exec_frac(op):
            if op.startswith("dec."):
                return dec_exec_frac
            if op.startswith("flow."):
                return flow_frac
            if op == "attention":  # QK^T + PV over valid (query, key) pairs only
                num = n_enc_layers * float((lengths.astype(np.float64) ** 2).sum()) + n_flow_layers * float((ylens.astype(np.float64) ** 2).sum())
                return num / (n_enc_layers * B * float(Tx) ** 2 + n_flow_layers * B * float(Ty) ** 2)
            if op.startswith("enc."):  # the flow's pre-transformer FFN/attention launches share the enc.* labels
                return None
            return 1.0

        n_flow_layers, n_enc_layers = hp.flow_n_flows, hp.n_layers
        mix = (n_enc_layers * Tx * enc_frac + n_flow_layers * Ty * flow_frac) / max(n_enc_layers * Tx + n_flow_layers * Ty, 1)
        rep = {k: (v[0], v[1], v[2] * (exec_frac(k[0]) if exec_frac(k[0]) is not None else (enc_frac if k[0] == "enc.proj" else mix)))
               for k, v in rep.items()}
        sess.set_options(use_graph=not args.no_graph, profile=False)
        by_op, by_kernel = {}, {}
        for (op, kern), v in rep.items():
            a = by_op.setdefault(op, [0, 0.0, 0.0]); a[0] += v[0]; a[1] += v[1]; a[2] += v[2]
            a = by_kernel.setdefault(kern, [0, 0.0, 0.0]); a[0] += v[0]; a[1] += v[1]; a[2] += v[2]
        # dominant kernel = the MFMA conv instantiation (the name rocprofv3 reports) with the most device time
        convk = {k: v for k, v in by_kernel.items() if k.startswith("conv") and v[2] > 0}
        dom_name, dom = max(convk.items(), key=lambda kv: kv[1][1]) if convk else ("none", [1, 1.0, 0.0])
        fam_launches, fam_ms, fam_flops = dom
        dom_ops = sorted({op for (op, kern) in rep if kern == dom_name})
        achieved = fam_flops / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        dev_ms_all = sum(v[1] for v in rep.values()) / nprof
        # the split-bf16 kernel issues 3 bf16 MFMAs per product: its ceiling is the dense bf16 peak / 3
        peak_of = lambda name: PEAK_BF16_MFMA_TFLOPS / 3.0 if name.startswith("conv_bf3") else PEAK_FP32_MFMA_TFLOPS
        conv_block = {"kernel": dom_name, "kernel_serves": dom_ops, "kernel_launches_per_forward": fam_launches // nprof,
                      "kernel_ms_per_forward": round(fam_ms / nprof, 4), "avg_launch_us": round(fam_ms / max(fam_launches, 1) * 1e3, 2),
                      "algorithmic_flops_per_launch": fam_flops / max(fam_launches, 1), "achieved": round(achieved, 3),
                      "peak": round(peak_of(dom_name), 1), "unit": "TFLOP/s", "frac": round(achieved / peak_of(dom_name), 4),
                      "frac_valid": round(achieved / peak_of(dom_name), 4)}
        # THE roofline object describes the kernel with the most device time of this workload (round-3 review: at c2 that is the
        # persistent step program of text encoder .. flow, not the decoder's conv kernel); when the MFMA conv kernel with the most
        # time is a different one it follows as "conv_kernel"
        big_name, big = max(((k, v) for k, v in by_kernel.items() if v[2] > 0), key=lambda kv: kv[1][1], default=(dom_name, dom))
        if big_name != dom_name:
            dom_name, (fam_launches, fam_ms, fam_flops) = big_name, big
            dom_ops = sorted({op for (op, kern) in rep if kern == dom_name})
            achieved = fam_flops / (fam_ms * 1e-3) / 1e12 if fam_ms > 0 else 0.0
        kpeak = peak_of(dom_name)
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 3), "peak": round(kpeak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / kpeak, 4), "frac_valid": round(achieved / kpeak, 4),
            "flops_are": "algorithmic FLOPs of the VALID columns of every item (padding, halo and tile-rounding columns a launch executes are not counted)",
            "traffic": None,
            "kernel": dom_name, "kernel_serves": dom_ops, "kernel_launches_per_forward": fam_launches // nprof,
            "kernel_ms_per_forward": round(fam_ms / nprof, 4),
            "avg_launch_us": round(fam_ms / max(fam_launches, 1) * 1e3, 2),
            "algorithmic_flops_per_launch": fam_flops / max(fam_launches, 1),
            "conv_kernel": conv_block if conv_block["kernel"] != dom_name else None,
            "forward": {"algorithmic_gflop": round(flops_fwd / 1e9, 3),
                        "achieved_tflops": round(flops_fwd / (elapsed / steps) / 1e12, 3),
                        "frac": round(flops_fwd / (elapsed / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                        "frac_is": "achieved_tflops / the fp32 MFMA peak (157.3)",
                        "sum_kernel_ms_eager": round(dev_ms_all, 4)},
            "by_kernel_ms_per_forward": {k: round(v[1] / nprof, 4) for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])},
            "by_op_ms_per_forward": {k: round(v[1] / nprof, 4) for k, v in sorted(by_op.items(), key=lambda kv: -kv[1][1])},
            "by_op_tflops": {k: round(v[2] / (v[1] * 1e-3) / 1e12, 2) for k, v in sorted(by_op.items(), key=lambda kv: -kv[1][1])
                             if v[2] > 0 and v[1] > 0},
        }

        # The event brackets above include the dispatch of the bracketed launch (~2.5 us at B=1); rocprofv3 --kernel-trace --stats
        # reports kernel begin -> end.  The committed summary of this workload's run is quoted beside the live figure.
        # (file-sourced fields are grouped under "committed" and say so: they are measurements of an earlier run of this command)
        committed = {"source": "committed"}
        try:  # the git commit the committed profile files were measured at (tools/evidence_run.sh writes it; the GPU box has no .git)
            with open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_HEAD.txt")) as f:
                committed["measured_at_git_head"] = f.read().strip()
        except OSError:
            committed["measured_at_git_head"] = None
        prof_csv = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{wname}_only_bench_rocprofv3_kernel_stats.csv")
        if not os.path.exists(prof_csv) and wname in ("c2", "c3"):  # the default command runs the c2 and the c3 leg
            prof_csv = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_default_bench_rocprofv3_kernel_stats.csv")
        avg_prof = rocprof_avg_us(prof_csv, dom_name)
        if avg_prof is not None:
            committed["avg_launch_us_rocprofv3"] = avg_prof
            committed["rocprofv3_summary"] = os.path.relpath(prof_csv, ROOT)

        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; they are collected
        # by tools/pmc_passes.sh (separate rocprofv3 --pmc passes of THIS command, FETCH_SIZE/WRITE_SIZE calibrated with
        # tools/pmc_calib as MI355X_MICROARCH.md prescribes) and committed as profiles/<PROFILE_ROUND>_pmc_<workload>.json.
        pmc_path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc_{wname}.json")
        if os.path.exists(pmc_path):
            try:
                with open(pmc_path) as f:
                    pk = json.load(f)["kernels"].get(dom_name)
                if pk and "hbm_bytes_per_launch" in pk:
                    roofline["traffic"] = round(pk["hbm_bytes_per_launch"])
                    committed["traffic"] = "HBM bytes per launch of the dominant kernel, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_passes.sh)"
                    committed["traffic_source"] = os.path.relpath(pmc_path, ROOT)
                    committed["hbm_gbps_at_avg_launch"] = round(pk["hbm_bytes_per_launch"] / (fam_ms / max(fam_launches, 1) * 1e-3) / 1e9, 1)
                    if "mfma_flops_per_launch" in pk:
                        committed["mfma_flops_executed_per_launch_pmc"] = pk["mfma_flops_per_launch"]
            except (OSError, ValueError, KeyError):
                pass
        roofline["committed"] = committed

        # Shader clock while this workload runs (round 6).  Why it is measured: single dense conv launches on N(0,1) operands (tools/bt_conv.py)
        # run at 1.86 - 2.11 GHz, and a roofline quoted at 2.4 GHz would then be unreachable; the forwards timed here hold 2.35 - 2.40
        # (profiles/r6_clock_in_forward.txt: inside the ResBlock launches of a c3 forward), so the nominal peak IS the yardstick -- the
        # line carries the measurement so that this stays checkable.  A burst of forwards is enqueued on the session stream (asynchronous
        # graph replays) and a probe of one-wave workgroups on a stream of its own compares s_memtime with the 100 MHz wall clock next to them.
        try:
            sess.set_options(use_graph=not args.no_graph, profile=False)
            n_burst = int(max(8, min(400, 0.06 / max(elapsed / steps, 1e-5))))  # ~60 ms of forwards
            lib.clock_probe(device=local_rank, duration_us=50, n=64)  # (creates the probe's stream and buffer: not while the burst is in flight)
            t_b0 = time.perf_counter()
            for _ in range(n_burst):
                step()
            t_b1 = time.perf_counter()
            ghz = lib.clock_probe(device=local_rank, duration_us=int(0.5 * n_burst * (elapsed / steps) * 1e6), n=64)
            t_b2 = time.perf_counter()
            sess.sync()
            t_b3 = time.perf_counter()
            clk = ghz[len(ghz) // 2]
            roofline["clock"] = {"shader_ghz_median_during_forwards": round(clk, 3), "p10": round(ghz[len(ghz) // 10], 3), "p90": round(ghz[len(ghz) * 9 // 10], 3),
                                 "nominal_ghz": 2.4, "peak_at_measured_clock": round(kpeak * clk / 2.4, 1),
                                 "frac_at_measured_clock": round(achieved / (kpeak * clk / 2.4), 4),
                                 "forward_frac_at_measured_clock": round(flops_fwd / (elapsed / steps) / 1e12 / (PEAK_FP32_MFMA_TFLOPS * clk / 2.4), 4),
                                 "burst": {"forwards": n_burst, "enqueue_ms": round((t_b1 - t_b0) * 1e3, 2), "probe_ms": round((t_b2 - t_b1) * 1e3, 2),
                                           "drain_after_probe_ms": round((t_b3 - t_b2) * 1e3, 2)},
                                 "how": "vits_debug_clock_probe: 64 one-wave workgroups on their own stream for half of a burst of forwards, s_memtime / s_memrealtime; "
                                        "the whole forward's average (every kernel and the gaps between them), not the dominant kernel's alone"}
        except Exception as e:  # a probe failure must not cost the bench line
            roofline["clock"] = {"error": str(e)[:200]}

        sess.close()
        return dict(B=B, Tx=Tx, Ty=Ty, lengths=lengths, ids=ids, dur=dur, valid_samples=valid_samples, job_samples=job_samples,
                    ms_per_step=ms_per_step, value=value, rtf=rtf, roofline=roofline, scales=scales,
                    timed_region_s=timed_region_s, blocks=len(blocks), launches=graph_nodes or sum(v[0] for v in rep.values()) // nprof,
                    rank_ms=rank_ms, finite_checked=finite_checked, persistent_per_step=persistent_per_step, rank_persist=rank_persist)

    R = measure(args.workload, args.steps, args.warmup, args.min_seconds)
    B, Tx, Ty, lengths, ids, dur = R["B"], R["Tx"], R["Ty"], R["lengths"], R["ids"], R["dur"]
    valid_samples, ms_per_step, value, rtf, roofline, scales = R["valid_samples"], R["ms_per_step"], R["value"], R["rtf"], R["roofline"], R["scales"]
    batch32 = None
    if args.workload == "c2" and not args.no_batch32:
        R3 = measure("c3", max(3, min(args.steps, 10)), 2, min(args.min_seconds, 0.5))
        batch32 = {"value": round(R3["value"], 1), "unit": "samples/s", "ms_per_step": round(R3["ms_per_step"], 4),
                   "x_realtime": round(1.0 / R3["rtf"], 1), "batch": R3["B"], "T_x": R3["Tx"], "T_y": R3["Ty"],
                   "samples_per_step_per_gpu": R3["valid_samples"],
                   "workload": "c3: B=32 ragged 20..200 tokens padded, durations pinned 3/token, fp32",
                   "timed_region_s": round(R3["timed_region_s"], 3),
                   "roofline": {k: R3["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_valid", "flops_are", "traffic", "kernel", "avg_launch_us", "forward", "clock", "committed") if k in R3["roofline"]}}

    # BASELINE configs[2] as written ("bf16 acoustic + fp32 vocoder" is allowed there): the same batch on a model created with
    # hparams.conv_precision = 1, i.e. the decoder's ResBlock convs as split-bf16 (3 bf16 MFMAs per product, fp32-class accuracy:
    # tests/test_hip_parity.py::test_bf16x3_decoder_variant).  A SECOND line: fp32 stays the default and the headline.
    batch32_bf16x3 = None
    if args.workload == "c2" and not args.no_batch32:
        hp3 = W.default_hparams()
        hp3.conv_precision = 1
        model3 = lib.create(W.synthetic_blob(hp3, 1234), local_rank)
        R4 = measure("c3", max(3, min(args.steps, 10)), 2, min(args.min_seconds, 0.5), model=model3)
        batch32_bf16x3 = {"value": round(R4["value"], 1), "unit": "samples/s", "ms_per_step": round(R4["ms_per_step"], 4), "dtype": "bf16x3",
                          "x_realtime": round(1.0 / R4["rtf"], 1), "batch": R4["B"], "T_x": R4["Tx"], "T_y": R4["Ty"],
                          "workload": "c3 as batch32, ResBlock, encoder / flow STORE and WaveNet gate convs split-bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate), the rest fp32",
                          "timed_region_s": round(R4["timed_region_s"], 3),
                          "roofline": {k: R4["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_us", "forward", "clock", "by_kernel_ms_per_forward") if k in R4["roofline"]}}
        # the whole-forward fraction of THIS line is priced against its own ceiling: split-bf16 issues 3 bf16 MFMAs per product
        fw = dict(batch32_bf16x3["roofline"].get("forward", {}))
        if fw:
            fw["frac"] = round(fw["achieved_tflops"] / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4)
            fw["frac_is"] = "achieved_tflops / (dense bf16 MFMA peak / 3 = 833.3): an upper bound on the ceiling, the fp32 stages of this forward have a lower one"
            batch32_bf16x3["roofline"]["forward"] = fw
        model3.close()

    # BASELINE configs[3]: the 256-request list dealt to the ranks by plan_shards (rank r runs shard r as one padded batch):
    # STRONG scaling -- total work is fixed, value = all requests' samples / the slowest rank
    batch256 = None
    mds_leg = None
    if args.workload == "c2" and not args.no_batch32:
        R5 = measure("c4", 3, 1, 0.0)
        batch256 = {"value": round(R5["value"], 1), "unit": "samples/s", "scaling": "strong", "ms_per_step": round(R5["ms_per_step"], 3),
                    "x_realtime": round(1.0 / R5["rtf"], 1), "requests": 256, "ranks": world, "requests_on_rank0": R5["B"],
                    "rank_ms": R5["rank_ms"], "imbalance": round(max(R5["rank_ms"]) / max(min(R5["rank_ms"]), 1e-9), 4),
                    "workload": "c4: 256 ragged requests (20..200 tokens, durations pinned 3/token) sharded over the ranks by "
                                "vosk_tts_amd.batching.plan_shards, one padded batch per rank, fp32"}

    streaming = None
    if rank == 0 and (args.workload == "c5" or (args.workload == "c2" and not args.no_extras)):
        # BASELINE configs[4] (2000 phonemes -> 6000 frames, 69.7 s of audio) through the host API, in every default line
        if args.workload == "c5":
            s_ids, s_len, s_dur = ids, lengths, dur
        else:
            s_ids, s_len, s_dur = make_workload("c5", np.random.default_rng(1234))
        streaming = streaming_leg(model, s_ids, s_len, s_dur)

    host_api = None
    if args.workload == "c2" and rank == 0 and not args.no_host_api:
        # The drop-in path: what Synth.synth_audio brackets (vosk_tts/synth.py:122-131): ids on the host -> int16 PCM on the
        # host through the C ABI (vits_synthesize_pcm16), one request per call, a fresh seed per request.
        #   "free_running": durations from the stochastic duration predictor (the production case; T_y varies per request)
        #   "pinned":       durations pinned 3/token (fixed work, directly comparable with the device-resident headline)
        sc = np.array([0.8, 1.0, 0.8], np.float32)

        def host_leg(forced):
            lat, samples = [], 0
            for i in range(8):
                model.synthesize_pcm16(ids, lengths, sc, [2], forced_durations=forced, seed=1000 + i)
            settle_gc()
            t_all = time.perf_counter()
            i = 0
            while i < 50 or time.perf_counter() - t_all < 0.5:
                t0 = time.perf_counter()
                pcm, ol = model.synthesize_pcm16(ids, lengths, sc, [2], forced_durations=forced, seed=1 + i)
                lat.append(time.perf_counter() - t0)
                samples += int(ol.sum())
                i += 1
                if i >= 5000:
                    break
            total = time.perf_counter() - t_all
            med = float(np.median(lat))
            return {"requests": i, "timed_region_s": round(total, 3), "ms_median": round(med * 1e3, 4), "ms_p90": round(float(np.percentile(lat, 90)) * 1e3, 4),
                    "ms_mean": round(total / i * 1e3, 4), "value": round(samples / total, 1), "unit": "samples/s",
                    "mean_samples_per_request": round(samples / i, 1), "x_realtime": round(samples / SAMPLE_RATE / total, 1)}

        host_api = {"entry_point": "vits_synthesize_pcm16 (ids on host -> int16 PCM on host; graphs replayed over shape buckets, "
                                   "scalars in a device block, pinned staging; includes H2D, the T_y round trip and D2H)",
                    "free_running": host_leg(None), "pinned": host_leg(dur),
                    "device_session_ms_per_step": round(ms_per_step, 4)}
        if not args.no_extras:
            try:
                host_api["concurrent"] = concurrency_leg(hp, local_rank)
            except Exception as e:  # the leg must never take the line down
                host_api["concurrent"] = {"error": repr(e)}
            try:
                host_api["bert_voice"] = bert_voice_leg(hp, local_rank)
            except Exception as e:
                host_api["bert_voice"] = {"error": repr(e)}

    multistream = None
    if args.workload == "c2" and not args.no_batch32:
        import copy

        a2 = copy.copy(args)
        a2.steps, a2.warmup = max(30, min(args.steps, 50)), 5  # (3 warm-up calls left the first timed requests on cold graphs / clocks)
        multistream = bench_multistream(a2, torch, rank, world, local_rank, dist, as_object=True)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = vits_cpu_baseline(blob, hp, ids, lengths, dur, args.workload, args.cpu_seconds)

    # The in-process front door over ALL the job's devices (rank 0 opens one Model replica per device): last, and only after every
    # other rank has let go of its device -- its model closed, so neither its weights nor its persistent-program lock are in the way
    # and nobody sits in a barrier holding a model while rank 0 opens `world` more.
    if args.workload == "c2" and not args.no_batch32:
        if rank != 0:
            model.close()
        if dist is not None:
            dist.barrier()
        if rank == 0:
            all_len = np.random.default_rng(1234).integers(20, 201, size=256)
            try:
                mds_leg = multi_device_synth_leg(hp, all_len, min(world, torch.cuda.device_count()))
            except Exception as e:  # the leg must never take the line down
                mds_leg = {"error": repr(e)}
        if dist is not None:
            dist.barrier()

    if rank == 0:
        line = {
            "metric": "audio_samples_per_sec", "value": round(value, 1), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if args.workload == "c4" else "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16x3", "data": "synthetic", "gc": GC_NOTE,
            "rtf": round(rtf, 6), "x_realtime": round(1.0 / rtf, 1),
            "timed_region_s": round(R["timed_region_s"], 3), "timed_blocks": R["blocks"], "launches_per_forward": R["launches"],
            "persistent_launches_per_forward": round(R["persistent_per_step"], 3),
            "rank_persistent_launches_per_forward": R["rank_persist"],
            "config": {"workload": f"{args.workload}: MB-iSTFT-VITS2 (ru-0.9-multi-shaped, SEEDED SYNTHETIC weights: timings are shape-exact, "
                                   f"dynamic range of a trained voice untested), B={B} T_x={Tx} (lengths {int(lengths.min())}..{int(lengths.max())}), "
                                   f"durations pinned 3/token -> T_y={Ty} with the duration predictor executed, "
                                   f"{valid_samples} valid samples/step/GPU, sid=2, scales=[0.8,1.0,0.8], inputs resident in HBM",
                       "batch": B, "T_x": Tx, "T_y": Ty, "samples_per_step_per_gpu": valid_samples,
                       "parallelism": f"replicas x{world} (no collective)", "hipgraph": not args.no_graph},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "host_api": host_api, "batch32": batch32, "batch32_bf16x3": batch32_bf16x3, "multistream": multistream,
            "streaming": streaming, "batch256_sharded": batch256, "multi_device_synth": mds_leg,
            "ranks_seen": len(R["rank_ms"]), "rank_ms": R["rank_ms"], "finite_check": R["finite_checked"],
            "launched_by": "bench.py" if os.environ.get("BENCH_SELF_LAUNCHED") else ("torch.distributed.run" if dist is not None else "single process"),
            "value_is": "device-resident session (inputs in HBM when the timed region starts, as the bench contract requires); the drop-in host "
                        "path (ids on the host -> int16 on the host, free-running) is host_api",
        }
        emit_line(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
