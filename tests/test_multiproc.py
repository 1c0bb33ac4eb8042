"""The N>1 path on CPU: world_size 2 over gloo.  Each rank takes its shard of a ragged request list
(no collective on the data path), synthesises it, and the gathered result must equal the
single-process result request by request.  The compute here is the CPU oracle on the tiny graph —
this test is about sharding / ordering / rank plumbing, the kernels are covered by -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, pickle
import numpy as np
sys.path.insert(0, os.environ["VITS_ROOT"])
import torch.distributed as dist
from vosk_tts_amd import weights as W
from vosk_tts_amd.batching import plan_shards, pad_batch, scatter_results
from vosk_tts_amd.capi import VitsLib

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["VITS_PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(3)                       # every rank derives the same request list
reqs = [rng.integers(1, 20, size=int(n)).tolist() for n in rng.integers(4, 30, size=7)]
shards = plan_shards([len(r) for r in reqs], world)
lib = VitsLib(os.path.join(os.environ["VITS_ROOT"], "oracle", "libvits_oracle.so"), "vitsref_")
model = lib.create(W.synthetic_blob(W.tiny_hparams(), 1234))
ids, lens = pad_batch(reqs, shards[rank])
dur = np.where(np.arange(ids.shape[1])[None] < lens[:, None], 2, 0).astype(np.int32)
audio, olen = model.synthesize(ids, lens, [0.0, 1.0, 0.0], np.zeros(len(lens), np.int64), forced_durations=dur)
mine = [audio[r, :olen[r]].copy() for r in range(len(lens))]
dist.barrier()
gathered = [None] * world
dist.all_gather_object(gathered, mine)               # results only; the data path itself needs no collective
t = __import__("torch").tensor([float(rank + 1)])
dist.all_reduce(t, op=dist.ReduceOp.MAX)             # the bench's max-over-ranks timing reduction
assert t.item() == world
if rank == 0:
    out = scatter_results(len(reqs), shards, gathered)
    with open(os.environ["VITS_OUT"], "wb") as f:
        pickle.dump({"reqs": reqs, "out": out, "shards": shards}, f)
dist.destroy_process_group()
'''


def test_plan_shards_balances_and_covers():
    from vosk_tts_amd.batching import plan_shards, predicted_cost

    rng = np.random.default_rng(0)
    lengths = rng.integers(20, 201, size=256)
    shards = plan_shards(lengths, 8, max_batch=32)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(256)) and all(len(s) == 32 for s in shards)
    load = np.array([predicted_cost(lengths[s]).sum() for s in shards])
    assert load.max() / load.min() < 1.05
    # within a shard requests are length-sorted (tight padding)
    assert all(list(lengths[s]) == sorted(lengths[s], reverse=True) for s in shards)


def test_sharding_helpers_properties():
    """plan_shards / pad_batch / scatter_results for arbitrary request lists: the shards partition the requests, respect the
    per-shard cap, are length-sorted; padding keeps every token; results come back in request order."""
    from hypothesis import given, settings, strategies as st

    from vosk_tts_amd.batching import pad_batch, plan_shards, scatter_results

    @settings(max_examples=200, deadline=None)
    @given(st.lists(st.integers(1, 300), min_size=0, max_size=70), st.integers(1, 9), st.data())
    def run(lengths, n_shards, data):
        cap = data.draw(st.one_of(st.none(), st.integers(-(-max(len(lengths), 1) // n_shards), 80)))
        shards = plan_shards(lengths, n_shards, max_batch=cap)
        assert len(shards) == n_shards
        assert sorted(i for s in shards for i in s) == list(range(len(lengths)))
        assert cap is None or all(len(s) <= cap for s in shards)
        assert all([lengths[i] for i in s] == sorted((lengths[i] for i in s), reverse=True) for s in shards)
        # no shard is left empty while another holds two requests more than it needs to
        sizes = [len(s) for s in shards]
        assert len(lengths) < n_shards or min(sizes) >= 1
        reqs = [list(range(1, n + 1)) for n in lengths]
        outs = []
        for s in shards:
            ids, lens = pad_batch(reqs, s)
            assert ids.shape == (len(s), max([lengths[i] for i in s], default=0)) and list(lens) == [lengths[i] for i in s]
            for r, i in enumerate(s):
                assert list(ids[r, :lens[r]]) == reqs[i] and not ids[r, lens[r]:].any()
            outs.append([int(ids[r, :lens[r]].sum()) for r in range(len(s))])
        assert scatter_results(len(lengths), shards, outs) == [n * (n + 1) // 2 for n in lengths]

    run()
    with pytest.raises(ValueError):
        plan_shards([5, 6, 7], 1, max_batch=2)
    with pytest.raises(ValueError):
        scatter_results(3, [[0, 1]], [["a", "b"]])


def test_two_process_replicas_match_single_process(tmp_path, oracle_lib, tiny_blob):
    import pickle
    import socket

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "out.pkl"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", VITS_PORT=str(port), VITS_ROOT=ROOT, VITS_OUT=str(out),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = pickle.load(open(out, "rb"))
    model = oracle_lib.create(tiny_blob)
    assert sorted(i for s in got["shards"] for i in s) == list(range(len(got["reqs"])))
    for req, audio in zip(got["reqs"], got["out"]):
        ids = np.array([req], np.int64)
        want, olen = model.synthesize(ids, [len(req)], [0.0, 1.0, 0.0], [0], forced_durations=np.full((1, len(req)), 2, np.int32))
        assert audio.shape[0] == olen[0] == len(req) * 2 * 256
        # a padded batch differs from a solo run only in the decoder tail of shorter items:
        # compare the part no padding can reach (receptive field < 25 frames)
        safe = max(0, (len(req) * 2 - 25)) * 256
        np.testing.assert_allclose(audio[:safe], want[0, :safe], rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()))


def test_bench_reads_rocprofv3_kernel_stats(tmp_path):
    """bench.rocprof_avg_us maps the engine's kernel names (EPI by name) onto the template arguments rocprofv3 prints"""
    import bench

    p = tmp_path / "k.csv"
    p.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                 '"void conv_mfma_ks_kernel<1, 1, 0, 1>(ConvParams)",10,158400,15840.4,50.0,1,2,3\n'
                 '"void conv_mfma_kernel<2, 2, 2, 2, 0>(ConvParams)",5,5630300,1126060.0,43.0,1,2,3\n'
                 '"void conv_mfma_ks_kernel<2, 1, 1, 1>(ConvParams)",16,271000,16954.5,7.0,1,2,3\n')
    assert bench.rocprof_avg_us(str(p), "conv_mfma_ks_kernel<1,1,STORE,1>") == 15.84
    assert bench.rocprof_avg_us(str(p), "conv_mfma_kernel<2,2,2,2,STORE>") == 1126.06
    assert bench.rocprof_avg_us(str(p), "conv_mfma_ks_kernel<2,1,GATE,1>") == 16.95
    assert bench.rocprof_avg_us(str(p), "conv_mfma_ks_kernel<1,1,COUPLE,1>") is None
    assert bench.rocprof_avg_us(str(tmp_path / "absent.csv"), "x") is None


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_multi_rank_entry_point_dry_run(launcher):
    """`python bench.py --gpus 2 ...` must work PLAINLY (it starts its own ranks) and under the driver's
    `python -m torch.distributed.run ...` form; ranks meet at a host-side (gloo) barrier and a max over their timings.  --dry-run
    skips the GPU work and runs exactly that plumbing: 2 processes on CPU, one JSON line from rank 0 that saw both ranks and
    carries the slower rank's time (rank r sleeps (1 + r) ms per step)."""
    import json
    import socket

    bench = os.path.join(ROOT, "bench.py")
    args = ["--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"]
    if launcher == "self":
        cmd = [sys.executable, bench] + args
    else:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench] + args
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dry_run"] is True
    assert d["ms_per_step"] >= 2.0  # the slower rank (2 ms per step), not the faster one
    assert d["launched_by"] == ("bench.py" if launcher == "self" else "torch.distributed.run")
    assert r.stdout.strip() == lines[0], "stdout must carry the one JSON line and nothing else (c10d's [Gloo] messages used to precede it)"
    assert d["rank_ms"][1] > d["rank_ms"][0] >= 1.0 and d["rank_persistent_launches_per_forward"] == [1.0, 1.0]
    assert d["batch256_requests_seen"] == 256  # configs[3]'s list dealt over the ranks: every request on exactly one rank


def test_bench_eight_rank_dry_run():
    """The driver's 8-GPU form (`--gpus 8`, here self-launched and on CPU): eight ranks rendezvous on 127.0.0.1, meet at the host-side
    barriers, and rank 0's line carries all eight per-rank timings, the slowest rank's time, the 256 requests of configs[3] dealt
    32 per rank, and per-rank persistent-launch counts."""
    import json

    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "10", "--warmup", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and len(d["rank_ms"]) == 8 and len(d["rank_persistent_launches_per_forward"]) == 8
    assert d["ms_per_step"] >= 8.0 and d["rank_ms"][7] >= 8.0 > d["rank_ms"][0]
    assert d["batch256_requests_seen"] == 256
    # round 6: the work plan_shards deals to every rank, so that the first real 8-GPU run can be checked against it
    pr = d["batch256_predicted"]
    assert len(pr["valid_frames_per_rank"]) == 8 and len(pr["padded_frames_per_rank"]) == 8
    assert 1.0 <= pr["imbalance_valid_max_over_mean"] < 1.01 and 1.0 <= pr["imbalance_padded_max_over_mean"] < 1.05
    assert "gc" in d
    import bench as B

    for rank in range(8):  # the real leg's shard of every rank: 32 requests each, all 256 covered
        ids, lens, dur = B.make_workload("c4", np.random.default_rng(1234), rank, 8)
        assert ids.shape[0] == 32 and (dur.sum(1) == 3 * lens).all()
