"""Pins the CPU oracle (oracle/vits_oracle.c) against fixtures produced by the
reference's own PyTorch modules (oracle/gen_golden.py).  CPU-only, fast."""
import os

import numpy as np
import pytest

from conftest import assert_close, golden

# fp32 restatement vs PyTorch-CPU fp32: different summation order / libm => ~1e-6; bar 2e-5
TOL = 2e-5


def _full(model, g, tol=TOL):
    ids, lengths, sid, scales = g["ids"], g["lengths"], g["sid"], g["scales"]
    x, m_p, logs_p = model.text_encoder(ids, lengths, sid)
    assert_close("x", g["x"], x, tol)
    assert_close("m_p", g["m_p_tok"], m_p, tol)
    assert_close("logs_p", g["logs_p_tok"], logs_p, tol)
    logw = model.duration(g["x"], lengths, sid, g["noise_dp"], float(scales[2]))
    assert_close("logw", g["logw"], logw, 5 * tol)
    # free-running durations agree except where exp(logw)*ls sits within fp32 noise of an integer
    dur_free, _, _ = model.regulate(g["logw"], None, lengths, float(scales[1]), g["m_p_tok"], g["logs_p_tok"], None, 0.0, 4096)
    assert np.array_equal(dur_free, g["w_ceil_free"])
    Ty = int(g["y_lengths"].max())
    dur, ylen, z_p = model.regulate(None, g["forced_durations"], lengths, float(scales[1]), g["m_p_tok"], g["logs_p_tok"],
                                    g["noise_prior"], float(scales[0]), Ty)
    assert np.array_equal(ylen, g["y_lengths"])
    assert_close("z_p", g["z_p"], z_p, tol)
    z = model.flow(g["z_p"], g["y_lengths"], sid)
    assert_close("z", g["z"], z, tol)
    mask = (np.arange(Ty)[None, :] < g["y_lengths"][:, None])[:, None, :]
    audio, mb = model.decoder(g["z"] * mask)
    assert_close("audio_mb", g["audio_mb"], mb, tol)
    assert_close("audio", g["audio"], audio, tol)
    # whole path through the one-call entry point (what Session.run uses)
    audio2, olen = model.synthesize(ids, lengths, scales, sid, noise_dp=g["noise_dp"], noise_prior=g["noise_prior"],
                                    forced_durations=g["forced_durations"])
    assert np.array_equal(olen, g["y_lengths"] * model.hp.hop_length)
    assert_close("audio(e2e)", g["audio"], audio2, 5 * tol)


def test_full_c1(oracle_default):
    _full(oracle_default, golden("full_c1"))


def test_full_b2_ragged(oracle_default):
    _full(oracle_default, golden("full_b2"))


def test_full_heavy_tailed_weights(oracle_lib):
    """The full path on HEAVY-TAILED weights (per-row log-normal scales, sigma 1: oracle/gen_golden_heavy.py ran the reference's own
    modules on them): |z| reaches 50, the gates and the exp() of the iSTFT heads leave the O(1) regime of the other fixtures."""
    from vosk_tts_amd import weights as W

    g = golden("full_heavy")
    assert np.abs(g["z"]).max() > 20  # the fixture really is heavy-tailed
    model = oracle_lib.create(W.synthetic_blob(W.default_hparams(), 1234, heavy_sigma=1.0))
    _full(model, g)


def test_tiny_b3_ragged(oracle_tiny):
    _full(oracle_tiny, golden("tiny_b3"))


def test_edge_empty_item_single_token_zero_durations(oracle_default):
    """lengths (6, 0, 1): an empty item (T_y clamps to 1, models.py:1691), a one-token item, zero durations at both ends"""
    g = golden("edge_b3")
    assert g["y_lengths"].tolist() == [12, 1, 2]
    _full(oracle_default, g)


def test_free_running_infer(oracle_default):
    """The real SynthesizerTrn.infer() call (free-running durations through ceil)."""
    g = golden("free_c1")
    audio, olen = oracle_default.synthesize(g["ids"], g["lengths"], g["scales"], g["sid"], noise_dp=g["noise_dp"],
                                            noise_prior=g["noise_prior"])
    assert olen[0] == g["y_lengths"][0] * 256
    assert_close("audio", g["audio"], audio, 5 * TOL)


def test_spline_linear_tails(oracle_default):
    g = golden("tails")
    logw = oracle_default.duration(g["x"], g["lengths"], g["sid"], g["noise_dp"], float(g["noise_scale_w"]))
    assert np.abs(g["noise_dp"] * 6.0).max() > 5.0  # the fixture really exercises |z| > tail_bound
    assert_close("logw", g["logw"], logw, 5 * TOL)


@pytest.mark.parametrize("T", [1, 3, 4, 5, 9])
def test_encoder_relative_window_edges(oracle_default, T):
    g = golden(f"enc_T{T}")
    x, m_p, logs_p = oracle_default.text_encoder(g["ids"], g["lengths"], g["sid"])
    assert_close("x", g["x"], x, TOL)
    assert_close("m_p", g["m_p_tok"], m_p, TOL)


def test_plain_generator_variant(oracle_lib):
    """Generator (models.py:845-898) incl. x = conv_pre(x) + cond(g), tanh tail, ups [8,8,2,2]."""
    from vosk_tts_amd import weights as W

    g = golden("plain_b2")
    m = oracle_lib.create(W.synthetic_blob(W.plain_hparams(), 1234))
    audio, mb = m.decoder(g["z"], sid=g["sid"])
    assert mb is None and audio.shape == g["audio"].shape
    assert_close("audio", g["audio"], audio, TOL)


def test_stabletts_hifigan_v1_vocoder(oracle_lib):
    """SURVEY.md 8f rank 3, vocoder stage: the HiFi-GAN V1 generator bundled with StableTTS (matcha/hifigan/models.py:
    148-199: 80 mels, conv_post with bias, no conditioning) as a vocoder-only blob (n_vocab = 0)."""
    from vosk_tts_amd import weights as W

    g = golden("hifigan_v1")
    m = oracle_lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234))
    audio, mb = m.decoder(g["mel"])
    assert mb is None and audio.shape == g["audio"].shape == (2, 10 * 256)
    assert_close("audio", g["audio"], audio, TOL)


def test_monotonic_alignment_search(oracle_lib):
    """the C restatement of core.pyx:7-42 reproduces the compiled Cython core bit for bit
    (ragged extents, t_x == t_y, single token, exact ties); the caller's scores are left untouched."""
    g = golden("mas")
    v = g["values"].copy()
    paths = oracle_lib.mas_maximum_path(v, g["t_ys"], g["t_xs"])
    assert np.array_equal(v, g["values"])
    assert np.array_equal(paths, g["paths"].astype(np.int32))
    # structure: one token per frame inside the extents, monotone, starts at token 0 and ends at t_x-1
    for b in range(len(g["t_ys"])):
        ty, tx = int(g["t_ys"][b]), int(g["t_xs"][b])
        assert paths[b, :ty].sum(1).tolist() == [1] * ty and paths[b, ty:].sum() == 0 and paths[b, :, tx:].sum() == 0
        idx = paths[b, :ty].argmax(1)
        assert idx[0] == 0 and idx[-1] == tx - 1 and np.all((np.diff(idx) == 0) | (np.diff(idx) == 1))
    from vosk_tts_amd.capi import VitsError

    with pytest.raises(VitsError):
        oracle_lib.mas_maximum_path(v, g["t_ys"] + 1000, g["t_xs"])


def test_monotonic_alignment_search_vs_the_compiled_reference(oracle_lib):
    """oracle/_ref/mas: the reference's own core.pyx compiled from /root/reference (build()); the restatement must agree bit
    for bit on random ragged batches, including what it does to `values` (the reference accumulates in place; ours does not)"""
    from conftest import load_reference_mas, random_mas_cases

    ref = load_reference_mas()
    if ref is None:
        pytest.skip("oracle/_ref/mas not built (no /root/reference at build time)")
    for v, t_ys, t_xs in random_mas_cases(91):
        want = np.zeros(v.shape, np.int32)
        ref.maximum_path_c(want, v.copy(), t_ys, t_xs)
        assert np.array_equal(oracle_lib.mas_maximum_path(v, t_ys, t_xs), want), v.shape


def test_constants(oracle_lib, oracle_default):
    import ctypes

    g = golden("consts")
    L = oracle_lib.lib
    L.vitsref_debug_istft_basis.restype = ctypes.POINTER(ctypes.c_float)
    L.vitsref_debug_istft_basis.argtypes = [ctypes.c_void_p]
    L.vitsref_debug_pqmf_filter.restype = ctypes.POINTER(ctypes.c_float)
    L.vitsref_debug_pqmf_filter.argtypes = [ctypes.c_void_p]
    basis = np.ctypeslib.as_array(L.vitsref_debug_istft_basis(oracle_default._h), shape=(18, 16))
    filt = np.ctypeslib.as_array(L.vitsref_debug_pqmf_filter(oracle_default._h), shape=(4, 63))
    assert_close("istft basis", g["istft_inverse_basis"], basis, 1e-6)
    assert_close("pqmf filter", g["pqmf_synthesis_filter"], filt, 1e-6)


def test_algorithmic_flops_match_survey(oracle_default):
    """14.40 MFLOP/token (+4608*T_x) and 162.4 MFLOP/frame (+3072*T_y)."""
    tok = oracle_default.algorithmic_flops(1, 1, 0)
    frame = oracle_default.algorithmic_flops(1, 0, 1)
    assert abs(tok - 14.40e6) / 14.40e6 < 0.01
    assert abs(frame - 162.4e6) / 162.4e6 < 0.01
    c2 = oracle_default.algorithmic_flops(1, 50, 150)
    assert abs(c2 - 25.2e9) / 25.2e9 < 0.02


def test_error_paths(oracle_lib, oracle_default, default_blob):
    from vosk_tts_amd.capi import VitsError

    with pytest.raises(VitsError):  # token id out of range
        oracle_default.text_encoder(np.array([[999]]), np.array([1]), np.array([0]))
    with pytest.raises(VitsError):  # speaker out of range
        oracle_default.text_encoder(np.array([[1]]), np.array([1]), np.array([1000]))
    with pytest.raises(VitsError):  # truncated blob
        oracle_lib.create(default_blob[:4096])
    with pytest.raises(VitsError):
        oracle_lib.create(b"NOTABLOB" + default_blob[8:])


def test_bench_cpu_baseline_leg_runs_without_a_gpu(tiny_blob):
    """bench.py's cpu_baseline leg (the oracle timed on the host, thread count picked by a short scan) is plain host code: run it
    here on the tiny model so that a slip in it cannot take the driver's default bench line down."""
    import importlib.util
    import os

    from vosk_tts_amd import weights as W

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    hp = W.tiny_hparams()
    ids, lengths, dur = bench.make_workload("c1", np.random.default_rng(1234))
    ids = np.minimum(ids, hp.n_vocab - 1)
    out = bench.vits_cpu_baseline(tiny_blob, hp, ids, lengths, dur, "c1", 0.2)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1 and out["cores"] <= out["threads_available"]
    assert str(out["cores"]) in out["seconds_per_forward_by_threads"]
    import json

    json.dumps(out)  # the bench prints it as part of its one JSON line


def test_bench_workload_names_and_shapes():
    """bench.py --workload: the BASELINE configs, the serving-shaped batches (s8 / s16) and u<N> (one utterance of N tokens, the
    kernel-selection / program-length sweeps of round 4); durations pinned to 3 frames per valid token."""
    import argparse
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name, B, Tx in (("c2", 1, 50), ("s8", 8, 47), ("s16", 16, 47), ("u120", 1, 120), ("u512", 1, 512), ("c5", 1, 2000)):
        assert bench._workload_name(name) == name
        ids, lengths, dur = bench.make_workload(name, np.random.default_rng(1234))
        assert ids.shape == (B, Tx) and lengths.shape == (B,) and int(lengths.max()) == Tx
        assert np.array_equal(dur.sum(1), 3 * lengths) and ids.min() >= 1
    ids, lengths, dur = bench.make_workload("c3", np.random.default_rng(1234))
    assert ids.shape[0] == 32 and 20 <= lengths.min() and lengths.max() <= 200
    for bad in ("u0", "u", "x3", "u99999"):
        with pytest.raises(argparse.ArgumentTypeError):
            bench._workload_name(bad)
