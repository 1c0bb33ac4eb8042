"""CPU tests: the StableTTS / Matcha oracle (oracle/stts_oracle.c) against fixtures produced by the reference's own
modules (oracle/gen_golden_stts.py): every stage and the full MatchaTTS.synthesise + HiFi-GAN path."""
import os

import numpy as np
import pytest

from conftest import assert_close

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5


def golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def stts_oracle(oracle_lib):
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi_stts import SttsModel

    voc = oracle_lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234))
    return SttsModel(oracle_lib, S.synthetic_blob(S.default_hparams(40, 7), 1234), voc)


def _stages(m, g, tol):
    x, mu = m.encoder(g["ids"], g["lengths"], g["sid"], g["bert"])
    assert_close("enc_x", g["enc_x"], x, tol)
    assert_close("mu_dp", g["mu_dp"], mu, tol)
    pde = g["phone_duration_extra"] if int(g["has_pde"]) else None
    d, yl = m.durations(g["mu_dp"], float(g["scales"][1]), pde)
    assert np.array_equal(d, g["durations"]) and yl.tolist() == g["y_lengths"].tolist()
    ylen = [int(g["y_lengths"][0])]
    assert_close("estimator", g["est_out"], m.estimator(g["est_x"], g["est_mu"], ylen, float(g["est_t"]), g["est_c"]), tol)
    assert_close("estimator(cfg)", g["est_fake_out"],
                 m.estimator(g["est_x"], g["est_fake_mu"], ylen, float(g["est_t"]), g["est_fake_c"]), tol)


def _e2e(m, g, tol):
    pde = g["phone_duration_extra"][0] if int(g["has_pde"]) else None
    audio, mel = m.synthesize(g["ids"][0], g["scales"], int(g["sid"][0]), g["bert"][0], pde, noise=g["noise"][0])
    assert mel.shape == g["mel"][0].shape and audio.shape == g["audio"][0].shape == (int(g["y_lengths"][0]) * 256,)
    assert_close("mel", g["mel"][0], mel, tol)
    assert_close("audio", g["audio"][0], audio, 5 * tol)


@pytest.mark.parametrize("name", ["stts_b1", "stts_nobert"])
def test_stts_stages(stts_oracle, name):
    _stages(stts_oracle, golden(name), TOL)


@pytest.mark.parametrize("name", ["stts_b1", "stts_nobert"])
def test_stts_synthesise_and_vocoder(stts_oracle, name):
    _e2e(stts_oracle, golden(name), TOL)


def test_stts_zero_bert_equals_none_and_errors(stts_oracle):
    """bert = None means zeros (the tokenizer-less multistream_v2 case, vosk_tts/synth.py:77-81); bad ids are refused."""
    from vosk_tts_amd.capi import VitsError

    g = golden("stts_nobert")
    a1, m1 = stts_oracle.synthesize(g["ids"][0], g["scales"], int(g["sid"][0]), None, None, noise=g["noise"][0], want_audio=False)
    assert a1 is None
    assert_close("mel", g["mel"][0], m1, TOL)
    bad = g["ids"][0].copy(); bad[2, 1] = 999
    with pytest.raises(VitsError, match="token id"):
        stts_oracle.synthesize(bad, g["scales"], 0)
    with pytest.raises(VitsError, match="speaker id"):
        stts_oracle.synthesize(g["ids"][0], g["scales"], 99)


@pytest.mark.parametrize("name,hidden", [("bert_small", 128), ("bert_768", 768)])
def test_bert_encoder_oracle_vs_transformers(oracle_lib, name, hidden):
    """The word-embedding BERT encoder (bert/model.onnx = transformers.BertModel, output hidden_states[-3]) restated in C,
    against activations of transformers' own BertModel on the same synthetic weights (oracle/gen_golden_stts.py)."""
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd.capi_stts import BertEncoder

    g = golden(name)
    enc = BertEncoder(oracle_lib, BW.synthetic_blob(BW.small_hparams(120, hidden, 4), 1234))
    out = enc.encode(g["ids"], g["types"])
    assert out.shape == g["hidden"].shape
    assert_close("hidden_states[-3]", g["hidden"], out, TOL)
    assert_close("run()", g["hidden"], enc.run(None, {"input_ids": [g["ids"]], "attention_mask": [np.ones_like(g["ids"])], "token_type_ids": [g["types"]]})[0], TOL)
