#!/usr/bin/env python3
"""Generate tests/golden/stts_*.npz by running the REFERENCE's own StableTTS / Matcha modules
(training/stabletts/matcha, imported through oracle/refimport_stts.py) on build-owned synthetic weights.
TEST INFRASTRUCTURE, container-only.   python oracle/gen_golden_stts.py

  stts_b1   B=1, T_x=14, 5-stream ids, random bert, phone_duration_extra with two forced pauses,
            MatchaTTS.synthesise (n_timesteps=5, Euler, guidance 0.5) with the noise draw captured; stage tensors:
            encoder concat x, mu_dp, durations, y_lengths, mu_y, one estimator call (real and CFG branch), decoder
            output before/after the pause fill, denormalised mel, and the waveform of the bundled HiFi-GAN V1.
  stts_nobert  the `multistream_v2` without tokenizer case of vosk_tts/synth.py:77-81: bert = zeros, no
            phone_duration_extra (None -> zeros inside synthesise).
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refimport_stts as R  # noqa: E402
from vosk_tts_amd import weights as W  # noqa: E402
from vosk_tts_amd import weights_stts as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1234
N_VOCAB, N_SPKS = 40, 7


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def build():
    hp = S.default_hparams(N_VOCAB, N_SPKS)
    tens = S.make_synthetic_weights(hp, SEED)
    net = R.build_reference_model(N_VOCAB, N_SPKS)
    sd = net.state_dict()
    mine = set(tens)
    theirs = {k for k in sd if not k.startswith("encoder.encoder.") and k not in ("mel_mean", "mel_std")}
    assert mine == theirs, sorted(mine ^ theirs)[:8]
    with torch.no_grad():
        for k, v in tens.items():
            assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
    assert abs(float(net.mel_mean) - hp.mel_mean) < 1e-6 and abs(float(net.mel_std) - hp.mel_std) < 1e-6
    # vocoder: the bundled HiFi-GAN V1 with the synthetic vocoder-only blob's tensors
    M = R.modules()
    vhp = W.hifigan_v1_vocoder_hparams()
    vt = W.make_synthetic_weights(vhp, SEED)
    with contextlib.redirect_stdout(io.StringIO()):
        voc = M["hifigan"].Generator(M["AttrDict"](M["hifigan_cfg"])).eval()
        voc.remove_weight_norm()
    vsd = voc.state_dict()
    with torch.no_grad():
        for k in vsd:
            vsd[k].copy_(torch.from_numpy(vt["dec." + k]))
    return hp, net, voc


def case(name, net, voc, hp, rng, Tx, bert_zero, pde_mode, sid, scales):
    ids = rng.integers(1, N_VOCAB, size=(1, 5, Tx)).astype(np.int64)
    lens = np.array([Tx], np.int64)
    bert = np.zeros((1, 768, Tx), np.float32) if bert_zero else rng.standard_normal((1, 768, Tx)).astype(np.float32)
    pde = None
    if pde_mode:
        pde = np.zeros((1, Tx), np.float32)
        pde[0, 3] = 7.0   # forced 7-frame pause tokens (synth.py g2p_multistream_scales)
        pde[0, Tx - 2] = 4.0
    temperature, length_scale, dp_temperature = scales  # scales = [noise_level, 1/speech_rate, duration_noise]
    x_t, l_t, s_t, b_t = torch.from_numpy(ids), torch.from_numpy(lens), torch.tensor([sid]), torch.from_numpy(bert)
    p_t = None if pde is None else torch.from_numpy(pde)
    out = {}
    with torch.no_grad():
        spk = net.spk_emb(s_t)
        dspk = net.dur_spk_emb(s_t)
        x, x_mel, mu_mel, x_dp, mu_dp, x_mask = net.encoder(x_t, l_t, spk, dspk, b_t)
        out["enc_x"], out["mu_dp"] = x.numpy(), mu_dp.numpy()
    captured = {}
    orig_randn = torch.randn

    def fake_randn(*size, **kw):
        shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
        z = torch.from_numpy(rng.standard_normal(tuple(int(v) for v in shape)).astype(np.float32))
        captured["z"] = z.numpy().copy()
        return z

    # record every estimator call of the Euler loop
    calls = []
    est = net.decoder.estimator
    orig_forward = est.forward

    def rec_forward(xx, mask, mu, t, c):
        r = orig_forward(xx, mask, mu, t, c)
        calls.append((xx.numpy().copy(), mu.numpy().copy(), float(t), c.numpy().copy(), r.numpy().copy()))
        return r

    est.forward = rec_forward
    torch.randn = fake_randn
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            res = net.synthesise(x_t, l_t, n_timesteps=hp.n_timesteps, temperature=temperature, dp_temperature=dp_temperature,
                                 spks=s_t, bert=b_t, length_scale=length_scale, phone_duration_extra=p_t)
    finally:
        torch.randn = orig_randn
        est.forward = orig_forward
    assert len(calls) == 2 * hp.n_timesteps
    y_len = int(res["mel_lengths"][0])
    with torch.no_grad():
        wav = voc(res["mel"]).clamp(-1, 1)
    out.update(
        ids=ids, lengths=lens, sid=np.array([sid], np.int64), bert=bert, scales=np.asarray(scales, np.float32),
        phone_duration_extra=np.zeros((1, Tx), np.float32) if pde is None else pde, has_pde=np.int32(pde is not None),
        noise=captured["z"], y_lengths=np.array([y_len], np.int64),
        durations=res["attn"][0, 0].sum(-1).numpy().astype(np.int32)[None],
        est_x=calls[0][0], est_mu=calls[0][1], est_t=np.float32(calls[0][2]), est_c=calls[0][3], est_out=calls[0][4],
        est_fake_mu=calls[1][1], est_fake_c=calls[1][3], est_fake_out=calls[1][4],
        est_last_t=np.float32(calls[-2][2]), decoder_outputs=res["decoder_outputs"].numpy(), mel=res["mel"].numpy(),
        audio=wav.numpy()[:, 0])
    save(name, **out)


def frontend_golden():
    """vosk_tts.Synth.g2p_multistream (vosk_tts/synth.py:273-347) on fixed sentences, with the toy dictionary and id map
    of vosk_tts_amd.toymodel as the model data.  synth.py is loaded by file path with onnxruntime stubbed out."""
    import importlib.util
    import types

    from vosk_tts_amd.toymodel import multistream_phoneme_id_map

    sys.modules.setdefault("onnxruntime", types.ModuleType("onnxruntime"))
    pkg = types.ModuleType("vosk_tts_ref")
    pkg.__path__ = [os.path.join(R.REF_ROOT, "vosk_tts")]
    sys.modules["vosk_tts_ref"] = pkg
    for name in ("g2p", "synth"):
        spec = importlib.util.spec_from_file_location(f"vosk_tts_ref.{name}", os.path.join(R.REF_ROOT, "vosk_tts", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"vosk_tts_ref.{name}"] = mod
        spec.loader.exec_module(mod)
    Synth = sys.modules["vosk_tts_ref.synth"].Synth
    model = types.SimpleNamespace(dic={"привет": "p rj i0 vj e1 t", "мир": "mj i1 r"},
                                  config={"phoneme_id_map": multistream_phoneme_id_map()}, tokenizer=None)
    synth = Synth(model)
    sentences = ['Прив+ет, "м+ир" - да... Нет!', "м+ир", "Да? Нет. (Мож+ет б+ыть): хорош+о; ладно - пок+а!", '"Прив+ет" сказ+ал +он, и уш+ёл...',
                 "од+ин -  дв+а -тр+и", "чт+о... чт+о?! д+а."]
    flat, offs = [], [0]
    for sent in sentences:
        for wp in (False, True):
            with contextlib.redirect_stdout(io.StringIO()):
                ids, _ = synth.g2p_multistream(sent, None, word_pos=wp) if wp else synth.g2p_multistream(sent, None)
            flat.extend(ids)
            offs.append(len(flat))
    # multistream_v3 front-end (g2p_multistream_scales, synth.py:360-456) and get_word_bert's row selection (synth.py:25-44)
    # with the toy WordPiece vocabulary and a stand-in encoder whose row i is filled with the value i
    import tempfile

    from tokenizers import BertWordPieceTokenizer

    from vosk_tts_amd.toymodel import BERT_VOCAB

    v3_sentences = ["прив+ет _ м+ир.", '"Да" _ - нет... ладно _', "од+ин, дв+а _ тр+и!"]
    v3_ids, v3_pde, v3_offs = [], [], [0]
    for sent in v3_sentences:
        with contextlib.redirect_stdout(io.StringIO()):
            ids, _, pde = synth.g2p_multistream_scales(sent, None)
        v3_ids.extend(ids); v3_pde.extend(pde); v3_offs.append(len(v3_ids))
    with tempfile.TemporaryDirectory() as d:
        vp = os.path.join(d, "vocab.txt")
        with open(vp, "w", encoding="utf-8") as f:
            f.write("\n".join(BERT_VOCAB) + "\n")
        model.tokenizer = BertWordPieceTokenizer(vocab=vp, unk_token="[UNK]", lowercase=True)
        model.bert_onnx = types.SimpleNamespace(run=lambda names, feed: [np.arange(len(feed["input_ids"][0]), dtype=np.float32)[:, None].repeat(4, 1)])
        wb_rows, wb_offs = [], [0]
        for sent in sentences + v3_sentences:
            for nopunc in (False, True):
                rows = synth.get_word_bert(sent.lower(), nopunc=nopunc)[:, 0].astype(np.int64)
                wb_rows.extend(rows.tolist()); wb_offs.append(len(wb_rows))
    np.savez_compressed(os.path.join(OUT, "stts_frontend.npz"), sentences=np.array(sentences), ids=np.array(flat, np.int64),
                        offsets=np.array(offs, np.int64), v3_sentences=np.array(v3_sentences), v3_ids=np.array(v3_ids, np.int64),
                        v3_pde=np.array(v3_pde, np.float32), v3_offsets=np.array(v3_offs, np.int64),
                        wb_rows=np.array(wb_rows, np.int64), wb_offsets=np.array(wb_offs, np.int64))
    print("  stts_frontend.npz")


def bert_golden():
    """transformers.BertModel (the model behind bert/model.onnx, onnx/bert-export.py:5-13: output hidden_states[-3]) on
    build-owned synthetic weights: a small geometry (hidden 128, 4 layers) and a 768-wide one (4 layers)."""
    from transformers import BertConfig, BertModel

    from vosk_tts_amd import weights_bert as BW

    for name, hp in (("bert_small", BW.small_hparams(120, 128, 4)), ("bert_768", BW.small_hparams(120, 768, 4))):
        cfg = BertConfig(vocab_size=hp.vocab_size, hidden_size=hp.hidden, num_hidden_layers=hp.n_layers, num_attention_heads=hp.n_heads,
                         intermediate_size=hp.intermediate, max_position_embeddings=hp.max_position, type_vocab_size=hp.type_vocab,
                         layer_norm_eps=hp.ln_eps, hidden_act="gelu", attn_implementation="eager")
        net = BertModel(cfg).eval()
        tens = BW.make_synthetic_weights(hp, SEED)
        sd = net.state_dict()
        with torch.no_grad():
            for k, v in tens.items():
                assert tuple(sd[k].shape) == v.shape, k
                sd[k].copy_(torch.from_numpy(v))
        rng = np.random.default_rng(55)
        T = 17
        ids = rng.integers(0, hp.vocab_size, size=(1, T)).astype(np.int64)
        types = np.zeros((1, T), np.int64)
        with torch.no_grad():
            out = net(input_ids=torch.from_numpy(ids), attention_mask=torch.ones(1, T, dtype=torch.long), token_type_ids=torch.from_numpy(types),
                      output_hidden_states=True)
        hs = torch.cat(out["hidden_states"][-3:-2], -1).squeeze(0)  # bert-export.py:11
        save(name, ids=ids[0], types=types[0], hidden=hs.numpy())


def main():
    bert_golden()  # before frontend_golden(): its onnxruntime stub module would confuse transformers' import probing
    frontend_golden()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    hp, net, voc = build()
    rng = np.random.default_rng(4321)
    case("stts_b1", net, voc, hp, rng, 14, False, True, 3, [0.8, 1.1, 0.8])
    case("stts_nobert", net, voc, hp, rng, 9, True, False, 1, [0.667, 1.0, 0.8])


if __name__ == "__main__":
    main()
