"""GPU parity tests of the StableTTS / Matcha ("multistream") path through the C ABI of include/stts_mi355.h:
HIP vs the reference goldens (tests/golden/stts_*.npz) and vs the CPU oracle on other seeded inputs."""
import os

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
STAGE_TOL = 1e-4
E2E_TOL = 5e-4  # north_star budget: 1e-3


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def stts_pair(hip_lib, oracle_lib):
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi_stts import SttsModel

    vblob = W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234)
    blob = S.synthetic_blob(S.default_hparams(40, 7), 1234)
    hip = SttsModel(hip_lib, blob, hip_lib.create(vblob, 0))
    ref = SttsModel(oracle_lib, blob, oracle_lib.create(vblob))
    yield hip, ref
    hip.close()


@pytest.mark.parametrize("name", ["stts_b1", "stts_nobert"])
def test_stts_stages_vs_golden(stts_pair, name):
    hip, _ = stts_pair
    g = golden(name)
    x, mu = hip.encoder(g["ids"], g["lengths"], g["sid"], g["bert"])
    assert_close("enc_x", g["enc_x"], x, STAGE_TOL)
    assert_close("mu_dp", g["mu_dp"], mu, STAGE_TOL)
    pde = g["phone_duration_extra"] if int(g["has_pde"]) else None
    d, yl = hip.durations(g["mu_dp"], float(g["scales"][1]), pde)
    assert np.array_equal(d, g["durations"]) and yl.tolist() == g["y_lengths"].tolist()
    ylen = [int(g["y_lengths"][0])]
    assert_close("estimator", g["est_out"], hip.estimator(g["est_x"], g["est_mu"], ylen, float(g["est_t"]), g["est_c"]), STAGE_TOL)
    assert_close("estimator(cfg)", g["est_fake_out"],
                 hip.estimator(g["est_x"], g["est_fake_mu"], ylen, float(g["est_t"]), g["est_fake_c"]), STAGE_TOL)


@pytest.mark.parametrize("name", ["stts_b1", "stts_nobert"])
def test_stts_synthesise_vs_golden(stts_pair, name):
    hip, _ = stts_pair
    g = golden(name)
    pde = g["phone_duration_extra"][0] if int(g["has_pde"]) else None
    audio, mel = hip.synthesize(g["ids"][0], g["scales"], int(g["sid"][0]), g["bert"][0], pde, noise=g["noise"][0])
    assert mel.shape == g["mel"][0].shape and audio.shape == g["audio"][0].shape
    assert_close("mel", g["mel"][0], mel, E2E_TOL)
    assert_close("audio", g["audio"][0], audio, E2E_TOL)


def test_stts_batched_estimator_and_cfm_vs_oracle(stts_pair):
    """estimator with B=3 ragged lengths, and the whole Euler/CFG loop (stage_cfm) on another size, against the oracle"""
    hip, ref = stts_pair
    rng = np.random.default_rng(17)
    B, T = 3, 44
    x = rng.standard_normal((B, 80, T)).astype(np.float32)
    mu = rng.standard_normal((B, 256, T)).astype(np.float32)
    c = rng.standard_normal((B, 128)).astype(np.float32)
    yl = np.array([44, 30, 17], np.int64)
    assert_close("estimator B=3", ref.estimator(x, mu, yl, 0.37, c), hip.estimator(x, mu, yl, 0.37, c), STAGE_TOL)
    T = 52
    mu_y = rng.standard_normal((256, T)).astype(np.float32)
    mu_y[:, 50:] = 0
    noise = rng.standard_normal((80, T)).astype(np.float32)
    want = ref.cfm(mu_y, 50, 2, noise, 0.8, 3)
    got = hip.cfm(mu_y, 50, 2, noise, 0.8, 3)
    assert_close("cfm", want[:, :50], got[:, :50], E2E_TOL)


def test_stts_seeded_noise_and_errors(stts_pair):
    """library Philox noise is the same stream in HIP and oracle; bad ids / speaker are refused; mel-only call works"""
    from vosk_tts_amd.capi import VitsError

    hip, ref = stts_pair
    g = golden("stts_nobert")
    a_ref, m_ref = ref.synthesize(g["ids"][0], g["scales"], 2, None, None, seed=11, n_timesteps=2)
    a_hip, m_hip = hip.synthesize(g["ids"][0], g["scales"], 2, None, None, seed=11, n_timesteps=2)
    assert_close("mel(seeded)", m_ref, m_hip, E2E_TOL)
    assert_close("audio(seeded)", a_ref, a_hip, E2E_TOL)
    a, m = hip.synthesize(g["ids"][0], g["scales"], 2, seed=11, n_timesteps=2, want_audio=False)
    assert a is None and np.array_equal(m, m_hip)
    bad = g["ids"][0].copy(); bad[0, 0] = 12345
    with pytest.raises(VitsError, match="token id"):
        hip.synthesize(bad, g["scales"], 0)
    with pytest.raises(VitsError, match="speaker id"):
        hip.synthesize(g["ids"][0], g["scales"], 77)


def test_stts_medium_utterance_vs_oracle_and_long_form_properties(stts_pair):
    """60 symbols -> 300 frames against the oracle (2 Euler steps keep the CPU side short), then a 400-symbol / 1600-frame
    utterance without an oracle run: finite, deterministic, length = 256 * sum(durations), mel-only call returns the
    mel the vocoder consumed."""
    hip, ref = stts_pair
    rng = np.random.default_rng(23)
    ids = rng.integers(1, 40, size=(5, 60)).astype(np.int64)
    pde = np.full(60, 5.0, np.float32)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    a_ref, m_ref = ref.synthesize(ids, sc, 4, None, pde, seed=2, n_timesteps=2)
    a_hip, m_hip = hip.synthesize(ids, sc, 4, None, pde, seed=2, n_timesteps=2)
    assert m_hip.shape == (80, 300) and a_hip.shape == (300 * 256,)
    assert_close("mel", m_ref, m_hip, E2E_TOL)
    assert_close("audio", a_ref, a_hip, E2E_TOL)
    ids = rng.integers(1, 40, size=(5, 400)).astype(np.int64)
    pde = np.full(400, 4.0, np.float32)
    a1, m1 = hip.synthesize(ids, sc, 1, None, pde, seed=9)
    a2, m2 = hip.synthesize(ids, sc, 1, None, pde, seed=9)
    assert a1.shape == (1600 * 256,) and np.isfinite(a1).all() and np.abs(a1).max() <= 1.0
    assert np.array_equal(a1, a2) and np.array_equal(m1, m2)
    _, m3 = hip.synthesize(ids, sc, 1, None, pde, seed=9, want_audio=False)
    assert np.array_equal(m1, m3)


def test_stts_batch_with_split_bf16_vocoder(hip_lib, oracle_lib):
    """hparams.conv_precision = 1 on the vocoder-only model: the HiFi-GAN V1 ResBlock convs of the 256- / 128-channel stages run
    as split-bf16 at batch size.  A batch of 8 utterances against the same batch on the fp32 vocoder (same acoustic model, same
    seeds): fp32-class agreement."""
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi_stts import SttsModel

    blob = S.synthetic_blob(S.default_hparams(40, 7), 1234)
    vhp = W.hifigan_v1_vocoder_hparams()
    v32 = hip_lib.create(W.synthetic_blob(vhp, 1234), 0)
    vhp.conv_precision = 1
    vbf = hip_lib.create(W.synthetic_blob(vhp, 1234), 0)
    m32, mbf = SttsModel(hip_lib, blob, v32), SttsModel(hip_lib, blob, vbf)
    rng = np.random.default_rng(5)
    B, Tx = 8, 60
    lengths = rng.integers(30, Tx + 1, size=B).astype(np.int64)
    ids = rng.integers(1, 40, size=(B, 5, Tx)).astype(np.int64)
    pde = np.full((B, Tx), 4.0, np.float32)
    sid = rng.integers(0, 7, size=B).astype(np.int64)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    try:
        a32, l32 = m32.synthesize_batch(ids, lengths, sc, sid, None, pde, seed=9, n_timesteps=2)
        abf, lbf = mbf.synthesize_batch(ids, lengths, sc, sid, None, pde, seed=9, n_timesteps=2)
        assert np.array_equal(l32, lbf) and np.isfinite(abf).all()
        assert not np.array_equal(a32, abf)  # the variant really ran
        assert_close("batch audio: split-bf16 vocoder vs fp32 vocoder", a32, abf, 1e-4)
    finally:
        m32.close(); mbf.close()


def test_stts_fast_path_equals_eager(hip_lib, stts_pair):
    """stts_synthesize replays two captured graphs per call (shape buckets: T_x to 8, frames to 32; scalars and inputs through
    a device block) -- against the eager path (vits_debug_fast_path(0)) on the same seeds: lengths inside and on bucket borders,
    with / without BERT rows and forced pauses, different seeds / temperatures / speakers / step counts on a cached bucket,
    mel-only and audio-only calls, and an invalid token id (error, then the next call is clean)."""
    hip, _ = stts_pair
    rng = np.random.default_rng(101)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    cases = []
    for Tx, use_bert, use_pde, seed, sid, n in [(13, True, True, 5, 2, 2), (16, False, True, 6, 0, 2), (11, True, False, 7, 3, 3),
                                                (13, False, True, 8, 1, 2), (40, True, True, 9, 6, 2)]:
        ids = rng.integers(1, 40, size=(5, Tx)).astype(np.int64)
        bert = rng.standard_normal((768, Tx)).astype(np.float32) if use_bert else None
        pde = rng.integers(2, 7, size=Tx).astype(np.float32) if use_pde else None
        s2 = sc.copy()
        s2[0] = 0.5 + 0.1 * (seed % 4)
        cases.append((ids, s2, sid, bert, pde, seed, n))
    try:
        outs = {}
        for on in (1, 0, 1):  # the second fast pass runs entirely on cached graphs
            hip_lib.lib.vits_debug_fast_path(on)
            outs[on] = [hip.synthesize(i, s2, sid, b, p, seed=sd, n_timesteps=n) for (i, s2, sid, b, p, sd, n) in cases]
        for k, ((a1, m1), (a0, m0)) in enumerate(zip(outs[1], outs[0])):
            assert a1.shape == a0.shape and m1.shape == m0.shape
            assert_close(f"mel, case {k}", m0, m1, 2e-5)
            assert_close(f"audio, case {k}", a0, a1, 5e-5)
        hip_lib.lib.vits_debug_fast_path(1)
        i, s2, sid, b, p, sd, n = cases[0]
        a, m = outs[1][0]
        _, m_only = hip.synthesize(i, s2, sid, b, p, seed=sd, n_timesteps=n, want_audio=False)
        a_only, _ = hip.synthesize(i, s2, sid, b, p, seed=sd, n_timesteps=n, want_mel=False)
        assert np.array_equal(m_only, m) and np.array_equal(a_only, a)
        a_other, _ = hip.synthesize(i, s2, sid, b, p, seed=sd + 1, n_timesteps=n)
        assert a_other.shape == a.shape and not np.array_equal(a_other, a)
        bad = i.copy()
        bad[0, 3] = 10 ** 6
        with pytest.raises(Exception):
            hip.synthesize(bad, s2, sid, b, p, seed=sd, n_timesteps=n)
        a_again, _ = hip.synthesize(i, s2, sid, b, p, seed=sd, n_timesteps=n)
        assert np.array_equal(a_again, a)
    finally:
        hip_lib.lib.vits_debug_fast_path(1)


@pytest.mark.parametrize("name,hidden", [("bert_small", 128), ("bert_768", 768)])
def test_bert_encoder_vs_transformers_golden(hip_lib, name, hidden):
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd.capi_stts import BertEncoder

    g = golden(name)
    enc = BertEncoder(hip_lib, BW.synthetic_blob(BW.small_hparams(120, hidden, 4), 1234))
    assert_close("hidden_states[-3]", g["hidden"], enc.encode(g["ids"], g["types"]), STAGE_TOL)
    enc.close()


def test_bert_encoder_graph_replay_equals_eager(hip_lib):
    """stts_bert_encode replays one captured forward per token-count bucket (multiples of 8, [PAD] columns beyond the sentence):
    same rows as the exact-size eager launches for every length of a bucket, on the capturing call and on replays, with and
    without token types, and from two threads at once (a busy bucket falls back to the eager form)."""
    import threading

    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd.capi_stts import BertEncoder

    enc = BertEncoder(hip_lib, BW.synthetic_blob(BW.small_hparams(120, 768, 4), 1234))
    rng = np.random.default_rng(12)
    try:
        for T in (1, 5, 7, 8, 9, 30, 31):
            ids = rng.integers(0, 120, size=T)
            types = rng.integers(0, 2, size=T) if T % 2 else None
            a = enc.encode(ids, types)
            a2 = enc.encode(ids, types)
            hip_lib.lib.vits_debug_fast_path(0)
            try:
                e = enc.encode(ids, types)
            finally:
                hip_lib.lib.vits_debug_fast_path(1)
            assert a.shape == (T, 768) and np.array_equal(a, a2)
            assert_close(f"BERT rows, T = {T}: replayed vs eager", e, a, 1e-6)
        ids = rng.integers(0, 120, size=14)
        want = enc.encode(ids)
        got, errs = [None] * 6, []

        def work(k):
            try:
                for _ in range(20):
                    got[k] = enc.encode(ids)
            except Exception as ex:  # noqa: BLE001
                errs.append(ex)

        th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs
        for k in range(6):
            assert_close(f"thread {k}", want, got[k], 1e-6)
    finally:
        enc.close()


def test_bert_base_geometry_vs_oracle(hip_lib, oracle_lib):
    """rubert-base geometry (12 x 768, 12 heads, 3072; 10 layers run for hidden_states[-3]) at 60 tokens, HIP vs oracle"""
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd.capi import VitsError
    from vosk_tts_amd.capi_stts import BertEncoder

    blob = BW.synthetic_blob(BW.base_hparams(300), 7)
    hip, ref = BertEncoder(hip_lib, blob), BertEncoder(oracle_lib, blob)
    ids = np.random.default_rng(3).integers(0, 300, size=60)
    assert_close("bert-base", ref.encode(ids), hip.encode(ids), 2 * STAGE_TOL)
    with pytest.raises(VitsError, match="token id"):
        hip.encode(np.array([1, 2, 9999]))
    hip.close()


def test_bert_ffn_k_sliced_launch_equals_the_single_launch(hip_lib):
    """Sentence-sized BERT calls run the 3072 -> 768 FFN matrix as three K-slices in one grouped conv_wp launch whose partial tensors the
    LayerNorm sums (stts.hip.h bert_forward); with the kernel choice forced (vits_debug_force_tile) the same layer runs as one launch
    of the K-split kernel.  Same rows up to the summation order, for one and for two column tiles."""
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd.capi_stts import BertEncoder

    enc = BertEncoder(hip_lib, BW.synthetic_blob(BW.base_hparams(300), 11))
    rng = np.random.default_rng(5)
    try:
        for T in (6, 16, 41):
            ids = rng.integers(0, 300, size=T)
            hip_lib.lib.vits_debug_fast_path(0)  # (graphs captured under one choice would replay it)
            try:
                sliced = enc.encode(ids)
                hip_lib.lib.vits_debug_force_tile(2)
                single = enc.encode(ids)
            finally:
                hip_lib.lib.vits_debug_force_tile(0)
                hip_lib.lib.vits_debug_fast_path(1)
            assert_close(f"T = {T}: K-sliced vs single launch", single, sliced, 2e-6)
            assert_close(f"T = {T}: replayed", sliced, enc.encode(ids), 2e-6)
    finally:
        enc.close()


def test_stts_batch_items_equal_their_single_utterance_calls(stts_pair):
    """stts_synthesize_batch: B = 4 ragged utterances (different lengths, speakers, BERT vectors, forced pauses) in one
    pass; every item must equal its own single-utterance call with seed + b (same kernels, per-item masks / zero padding
    / solo vocoding), and the first item is also checked against the oracle."""
    hip, ref = stts_pair
    rng = np.random.default_rng(41)
    B, Tx = 4, 22
    lengths = np.array([22, 9, 15, 4], np.int64)
    ids = rng.integers(1, 40, size=(B, 5, Tx)).astype(np.int64)
    bert = rng.standard_normal((B, 768, Tx)).astype(np.float32)
    pde = np.zeros((B, Tx), np.float32)
    pde[0, 3] = 6.0; pde[2, 1] = 9.0
    sid = np.array([0, 3, 6, 1], np.int64)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    audio, olen = hip.synthesize_batch(ids, lengths, sc, sid, bert, pde, seed=30, n_timesteps=3)
    assert audio.shape[0] == B and audio.shape[1] == olen.max() and np.all(olen % 256 == 0) and len(set(olen.tolist())) > 1
    for b in range(B):
        L = int(lengths[b])
        one, _ = hip.synthesize(ids[b][:, :L], sc, int(sid[b]), bert[b][:, :L], pde[b][:L], seed=30 + b, n_timesteps=3, want_mel=False)
        assert one.shape[0] == olen[b]
        # not bit-for-bit: the batch is big enough for the big-tile conv kernel where the single call takes the K-split one
        assert_close(f"item {b}", one, audio[b, :olen[b]], 2e-5)
        assert not audio[b, olen[b]:].any()
    want, wlen = ref.synthesize_batch(ids, lengths, sc, sid, bert, pde, seed=30, n_timesteps=3)
    assert np.array_equal(wlen, olen) and want.shape == audio.shape
    assert_close("batch vs oracle", want, audio, E2E_TOL)


def test_stts_single_speaker_model_and_custom_steps(hip_lib, oracle_lib):
    """n_spks = 1: no speaker tables in the blob, the speaker vectors are zeros (matcha_tts.py:136-139 skips the embedding);
    guidance and the Euler loop still run; 1 and 7 steps against the oracle."""
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi_stts import SttsModel

    hp = S.default_hparams(30, 1)
    blob = S.synthetic_blob(hp, 99)
    assert "spk_emb.weight" in S.make_synthetic_weights(hp, 99)  # the spec still carries the (unused) table of one row
    vblob = W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234)
    hip = SttsModel(hip_lib, blob, hip_lib.create(vblob, 0))
    ref = SttsModel(oracle_lib, blob, oracle_lib.create(vblob))
    ids = np.random.default_rng(5).integers(1, 30, size=(5, 11)).astype(np.int64)
    pde = np.full(11, 4.0, np.float32)
    sc = np.array([0.6, 0.9, 0.8], np.float32)
    for steps in (1, 7):
        _, m_ref = ref.synthesize(ids, sc, 0, None, pde, seed=1, n_timesteps=steps, want_audio=False)
        _, m_hip = hip.synthesize(ids, sc, 0, None, pde, seed=1, n_timesteps=steps, want_audio=False)
        assert m_hip.shape == (80, 44)
        assert_close(f"mel ({steps} steps)", m_ref, m_hip, E2E_TOL)
    hip.close()


def test_stts_is_reentrant_from_threads(stts_pair):
    """the gRPC server shares one Synth across a thread pool (server/tts_server.py:39-40,57): concurrent
    stts_synthesize / stts_synthesize_batch calls on one model give exactly the results of sequential calls"""
    import threading

    hip, _ = stts_pair
    rng = np.random.default_rng(61)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    jobs = []
    for k in range(6):
        T = int(rng.integers(6, 30))
        jobs.append((rng.integers(1, 40, size=(5, T)).astype(np.int64), np.full(T, 3.0, np.float32), int(rng.integers(0, 7)), 100 + k))
    want = [hip.synthesize(i, sc, s, None, p, seed=sd, n_timesteps=2, want_mel=False)[0] for (i, p, s, sd) in jobs]
    got = [None] * len(jobs)
    errs = []

    def work(k):
        try:
            for _ in range(3):
                i, p, s, sd = jobs[k]
                got[k] = hip.synthesize(i, sc, s, None, p, seed=sd, n_timesteps=2, want_mel=False)[0]
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for k in range(len(jobs)):
        assert want[k].shape == got[k].shape, (k, want[k].shape, got[k].shape)
        d = np.abs(want[k] - got[k])
        assert np.array_equal(want[k], got[k]), (k, int((d > 0).sum()), want[k].size, float(np.nanmax(d)), int(np.isnan(got[k]).sum()),
                                                 np.argwhere(d > 0)[[0, -1]].ravel().tolist())


def test_stts_fresh_sessions_on_recycled_poisoned_memory(stts_pair, hip_lib):
    """a new model's first calls, issued from 8 threads at once (every thread creates its sessions), on device memory that was
    just released full of 0xFF bytes and with NaN-poisoned workspaces: nothing may depend on what a fresh allocation holds
    (regression: the sessions' error word was cleared on the null stream, unordered with the non-blocking session stream)"""
    import threading

    import torch
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi_stts import SttsModel

    hip, _ = stts_pair
    rng = np.random.default_rng(67)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    jobs = [(rng.integers(1, 40, size=(5, T)).astype(np.int64), np.full(T, 2.0, np.float32), k % 7, 200 + k) for k, T in enumerate((5, 9, 14, 20, 27, 33, 11, 3))]
    want = [hip.synthesize(i, sc, s, None, p, seed=sd, n_timesteps=2) for (i, p, s, sd) in jobs]
    for _ in range(3):
        junk = [torch.full((n,), -1, dtype=torch.int32, device="cuda") for n in (1, 64, 1 << 10, 1 << 16, 1 << 22, 1 << 26)]
        torch.cuda.synchronize()
        del junk
        torch.cuda.empty_cache()
        hip_lib.lib.vits_debug_poison_workspace(1)
        try:
            fresh = SttsModel(hip_lib, S.synthetic_blob(S.default_hparams(40, 7), 1234), hip_lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0))
            got, errs = [None] * len(jobs), []

            def work(k):
                try:
                    i, p, s, sd = jobs[k]
                    got[k] = fresh.synthesize(i, sc, s, None, p, seed=sd, n_timesteps=2)
                except Exception as e:  # noqa: BLE001
                    errs.append(e)

            ths = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
            [t.start() for t in ths]
            [t.join() for t in ths]
            fresh.close()
        finally:
            hip_lib.lib.vits_debug_poison_workspace(0)
        assert not errs, errs
        for k in range(len(jobs)):
            assert np.array_equal(want[k][0], got[k][0]) and np.array_equal(want[k][1], got[k][1]), k


def test_stts_tiny_utterances_vs_oracle(stts_pair):
    """1..5 symbols, 1..2 frames each (T_y as small as 1, padded to 4): every tile / mask edge of the path"""
    hip, ref = stts_pair
    rng = np.random.default_rng(71)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    for Tx, d in ((1, 1.0), (2, 1.0), (3, 2.0), (5, 1.0)):
        ids = rng.integers(1, 40, size=(5, Tx)).astype(np.int64)
        pde = np.full(Tx, d, np.float32)
        a_ref, m_ref = ref.synthesize(ids, sc, 3, None, pde, seed=8, n_timesteps=2)
        a_hip, m_hip = hip.synthesize(ids, sc, 3, None, pde, seed=8, n_timesteps=2)
        assert m_hip.shape == (80, int(Tx * d)) and a_hip.shape == (int(Tx * d) * 256,)
        assert_close(f"mel Tx={Tx}", m_ref, m_hip, E2E_TOL)
        assert_close(f"audio Tx={Tx}", a_ref, a_hip, E2E_TOL)


def test_stts_streamed_chunks_equal_the_one_shot_call(hip_lib):
    """stts_stream_open / SttsSession.run_stream: the vocoder streamed over the mel in windows (first chunk alone, then 8-chunk
    windows) gives exactly the one-shot audio, including the clamp, for chunk sizes that do and do not divide T_y and for an
    utterance shorter than one window."""
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.session_stts import SttsSession

    sess = SttsSession(S.synthetic_blob(S.default_hparams(40, 7), 1234), W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), lib=hip_lib)
    rng = np.random.default_rng(5)
    for T, per, chunks in ((400, 4.0, (64, 100)), (12, 3.0, (64,))):
        feed = {"input": rng.integers(1, 40, size=(1, 5, T)).astype(np.int64), "input_lengths": np.array([T], np.int64),
                "scales": np.array([0.8, 1.0, 0.8], np.float32), "sid": np.array([2], np.int64),
                "phone_duration_extra": np.full((1, T), per, np.float32), "vits.seed": 11}
        wav, n = sess.run(None, feed)
        assert n[0] == int(T * per) * 256
        for cf in chunks:
            parts = list(sess.run_stream(None, feed, chunk_frames=cf))
            assert all(len(p) == cf * 256 for p in parts[:-1]) and len(parts) == -(-int(T * per) // cf)
            got = np.concatenate(parts)
            assert got.shape == wav[0].shape
            assert_close(f"stream T={T} chunk={cf}", wav[0], got, 1e-5)  # windows of other widths run other tile shapes: summation order
    # the vocoder alone over a caller-held mel
    _, mel = sess._model.synthesize(feed["input"][0], feed["scales"], 2, None, feed["phone_duration_extra"][0], seed=11)
    got = np.concatenate(list(sess._vocoder.stream_latent(mel, chunk_frames=16, clamp=True)))
    assert_close("stream_latent", wav[0], got, 1e-5)
    sess.close()
