#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own PyTorch modules
(imported from /root/reference/training/vits2, see oracle/refimport.py) on
build-owned synthetic weights.  TEST INFRASTRUCTURE, container-only: the
reference cannot travel to the GPU box, these small vectors do.

    python oracle/gen_golden.py            # rewrites tests/golden/

Fixtures are DATA ONLY: inputs (ids, lengths, sid, scales, captured noise,
durations) and the reference's outputs per stage.  Weights are NOT stored; they
are regenerated from (hparams, seed) by vosk_tts_amd.weights on any machine.

Cases
  full_c1      default config, B=1, T_x=10, durations pinned to 3 (BASELINE configs[0] shape)
  full_b2      default config, B=2 ragged (12, 9), random durations 1..4 incl. a zero
  free_c1      default config, B=1, T_x=16, real SynthesizerTrn.infer() call with
               torch.randn / randn_like captured (free-running durations, ceil path)
  tails        default config, duration predictor with noise_scale_w=6 (spline linear tails, |z|>5)
  enc_T{1,3,4,5,9}  text encoder at the relative-attention edge lengths (window 4)
  tiny_b3      scaled-down config (hidden 64, 3 layers, 5 speakers), B=3 ragged
  consts       OnnxSTFT.inverse_basis and PQMF.synthesis_filter buffers
  edge_b3      default config, B=3 with lengths (6, 0, 1): an empty item (y_length clamps to 1 frame, models.py:1691), a
               one-token item, and zero durations at both ends of the first item
  plain_b2     the plain HiFi-GAN `Generator` decoder variant (models.py:845-898) with speaker conditioning,
               ups [8,8,2,2]: reference Generator module alone, B=2
  hifigan_v1   the HiFi-GAN V1 generator bundled with StableTTS (training/stabletts/matcha/hifigan/models.py:148-199,
               config.py v1; the `vocoder.decode(mel)` of matcha/onnx/export.py:28-32): 80-channel mel -> waveform, B=2
  mas          monotonic_align.maximum_path_c (the reference's Cython core, compiled here from its own .pyx):
               ragged batch incl. t_x == t_y, t_x == 1, ties, noise-scaled scores
  g2p          known answers of vosk_tts/g2p.py:convert (examples at g2p.py:5-11 + extra words)
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refimport  # noqa: E402
from vosk_tts_amd import weights as W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def ref_for(hp, tensors):
    cfg = refimport.ref_config()
    cfg["model"]["hidden_channels"] = hp.hidden_channels
    cfg["model"]["inter_channels"] = hp.inter_channels
    cfg["model"]["filter_channels"] = hp.filter_channels
    cfg["model"]["n_layers"] = hp.n_layers
    cfg["model"]["gin_channels"] = hp.gin_channels
    cfg["model"]["upsample_initial_channel"] = hp.dec_initial_channel
    cfg["data"]["n_speakers"] = hp.n_speakers
    net = refimport.build_reference_model(n_vocab=hp.n_vocab, cfg=cfg)
    refimport.load_into_reference(net, tensors)
    return net


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  {name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def full_case(net, hp, name, ids, lengths, sid, scales, durations, rng):
    B, Tx = ids.shape
    nd = rng.standard_normal((B, 2, Tx)).astype(np.float32)
    r = refimport.run_reference_stages(
        net, ids, lengths, sid, scales, nd,
        lambda s: torch.from_numpy(rng.standard_normal(s).astype(np.float32)),
        forced_durations=durations)
    taps = np.stack(r["flow_taps"])  # [2*n_flows, B, I, T_y] in execution order (Flip, L3, Flip, L2, ...)
    save(name, ids=ids.astype(np.int64), lengths=lengths.astype(np.int64), sid=sid.astype(np.int64),
         scales=np.asarray(scales, np.float32), noise_dp=nd, noise_prior=r["noise_prior"],
         forced_durations=durations.astype(np.int32),
         x=r["x"], m_p_tok=r["m_p_tok"], logs_p_tok=r["logs_p_tok"], logw=r["logw"][:, 0],
         w_ceil_free=r["w_ceil_free"][:, 0].astype(np.int32), y_lengths=r["y_lengths"].astype(np.int64),
         z_p=r["z_p"], flow_taps=taps, z=r["z"], audio_mb=r["audio_mb"], audio=r["audio"][:, 0])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    hp = W.default_hparams()
    tens = W.make_synthetic_weights(hp, SEED)
    net = ref_for(hp, tens)
    print("default config:")
    rng = np.random.default_rng(1234)
    # ---- full_c1
    ids = rng.integers(1, hp.n_vocab, size=(1, 10))
    full_case(net, hp, "full_c1", ids, np.array([10]), np.array([2]), [0.667, 1.0, 0.8], np.full((1, 10), 3), rng)
    # ---- full_b2 (ragged batch, zero duration inside)
    ids = rng.integers(1, hp.n_vocab, size=(2, 12))
    dur = rng.integers(1, 5, size=(2, 12))
    dur[0, 4] = 0
    full_case(net, hp, "full_b2", ids, np.array([12, 9]), np.array([2, 5]), [0.667, 1.1, 0.8], dur, rng)
    # ---- free_c1: the real infer() with both randn draws captured
    ids = rng.integers(1, hp.n_vocab, size=(1, 16))
    nd = rng.standard_normal((1, 2, 16)).astype(np.float32)
    captured = {}
    orig_randn, orig_randn_like = torch.randn, torch.randn_like

    def fake_randn(*a, **k):
        return torch.from_numpy(nd.copy())

    def fake_randn_like(t, **k):
        e = torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32))
        captured["prior"] = e.numpy().copy()
        return e

    torch.randn, torch.randn_like = fake_randn, fake_randn_like
    try:
        with torch.no_grad():
            o, o_mb, attn, y_mask, (z, z_p, m_p, logs_p) = net.infer(
                torch.from_numpy(ids), torch.tensor([16]), sid=torch.tensor([2]),
                noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8)
    finally:
        torch.randn, torch.randn_like = orig_randn, orig_randn_like
    durations = attn[0, 0].sum(0).numpy().astype(np.int32)[None]  # [1,T_x]
    save("free_c1", ids=ids.astype(np.int64), lengths=np.array([16], np.int64), sid=np.array([2], np.int64),
         scales=np.array([0.667, 1.0, 0.8], np.float32), noise_dp=nd, noise_prior=captured["prior"],
         durations=durations, y_lengths=np.array([z.shape[2]], np.int64), z=z.numpy(), audio=o.numpy()[:, 0])
    # ---- spline tails
    ids = rng.integers(1, hp.n_vocab, size=(1, 24))
    nd = rng.standard_normal((1, 2, 24)).astype(np.float32)
    r = refimport.run_reference_stages(net, ids, np.array([24]), np.array([7]), [0.667, 1.0, 6.0], nd,
                                       lambda s: torch.zeros(s), forced_durations=np.ones((1, 24)))
    save("tails", ids=ids.astype(np.int64), lengths=np.array([24], np.int64), sid=np.array([7], np.int64),
         noise_dp=nd, noise_scale_w=np.float32(6.0), x=r["x"], logw=r["logw"][:, 0])
    # ---- encoder edge lengths
    for T in (1, 3, 4, 5, 9):
        ids = rng.integers(1, hp.n_vocab, size=(1, T))
        with torch.no_grad():
            g = net.emb_g(torch.tensor([3])).unsqueeze(-1)
            x, m_p, logs_p, _ = net.enc_p(torch.from_numpy(ids), torch.tensor([T]), g=g)
        save(f"enc_T{T}", ids=ids.astype(np.int64), lengths=np.array([T], np.int64), sid=np.array([3], np.int64),
             x=x.numpy(), m_p_tok=m_p.numpy(), logs_p_tok=logs_p.numpy())
    # ---- constants
    pq = refimport.ref_modules()["pqmf"].PQMF("cpu")
    save("consts", istft_inverse_basis=net.dec.stft.inverse_basis.numpy()[:, 0, :],
         pqmf_synthesis_filter=pq.synthesis_filter.numpy()[0])
    # ---- tiny config
    print("tiny config:")
    thp = W.tiny_hparams()
    ttens = W.make_synthetic_weights(thp, SEED)
    tnet = ref_for(thp, ttens)
    ids = rng.integers(1, thp.n_vocab, size=(3, 20))
    dur = rng.integers(0, 4, size=(3, 20))
    full_case(tnet, thp, "tiny_b3", ids, np.array([20, 7, 13]), np.array([0, 4, 2]), [0.5, 0.9, 0.7], dur, rng)
    # ---- edge cases: empty item, single token, zero durations at the boundaries (separate rng: keeps the older fixtures stable)
    erng = np.random.default_rng(77)
    ids = erng.integers(1, hp.n_vocab, size=(3, 6))
    dur = erng.integers(1, 4, size=(3, 6))
    dur[0, 0] = 0
    dur[0, 5] = 0
    full_case(net, hp, "edge_b3", ids, np.array([6, 0, 1]), np.array([1, 2, 3]), [0.667, 1.0, 0.8], dur, erng)
    # ---- plain Generator variant: the reference module alone
    print("plain Generator:")
    php = W.plain_hparams()
    ptens = W.make_synthetic_weights(php, SEED)
    models = refimport.ref_modules()["models"]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        gen = models.Generator(php.inter_channels, "1", [3, 7, 11], [[1, 3, 5]] * 3, [8, 8, 2, 2], php.dec_initial_channel,
                               [16, 16, 4, 4], gin_channels=php.gin_channels).eval()
        gen.remove_weight_norm()
    sd = gen.state_dict()
    with torch.no_grad():
        for k in sd:
            sd[k].copy_(torch.from_numpy(ptens["dec." + k]))
    z = rng.standard_normal((2, php.inter_channels, 12)).astype(np.float32)
    sid = np.array([1, 3], np.int64)
    with torch.no_grad():
        g = torch.from_numpy(ptens["emb_g.weight"][sid]).unsqueeze(-1)
        audio = gen(torch.from_numpy(z), g=g)
    save("plain_b2", z=z, sid=sid, audio=audio.numpy()[:, 0])
    # ---- StableTTS' bundled HiFi-GAN V1 vocoder (vocoder stage only)
    sys.path.insert(0, os.path.join(refimport.REF_ROOT, "training", "stabletts"))
    from matcha.hifigan.config import v1 as hifigan_v1_cfg  # noqa: E402
    from matcha.hifigan.env import AttrDict  # noqa: E402
    from matcha.hifigan.models import Generator as HifiGenerator  # noqa: E402

    vhp = W.hifigan_v1_vocoder_hparams()
    vtens = W.make_synthetic_weights(vhp, SEED)
    with contextlib.redirect_stdout(io.StringIO()):
        hg = HifiGenerator(AttrDict(hifigan_v1_cfg)).eval()
        hg.remove_weight_norm()
    sd = hg.state_dict()
    assert set("dec." + k for k in sd) == set(vtens), sorted(set("dec." + k for k in sd) ^ set(vtens))[:5]
    with torch.no_grad():
        for k in sd:
            sd[k].copy_(torch.from_numpy(vtens["dec." + k]))
    vrng = np.random.default_rng(91)
    mel = vrng.standard_normal((2, 80, 10)).astype(np.float32)
    with torch.no_grad():
        wav = hg(torch.from_numpy(mel))
    save("hifigan_v1", mel=mel, audio=wav.numpy()[:, 0])
    # ---- monotonic alignment search: the reference's compiled Cython core
    mas = refimport.build_reference_mas()
    B, Ty, Tx = 7, 96, 40
    values = (rng.standard_normal((B, Ty, Tx)) * 3.0).astype(np.float32)
    values[5] = np.round(values[5])          # many exact ties -> exercises the strict `<` rule
    t_ys = np.array([96, 50, 40, 7, 96, 64, 1], np.int32)
    t_xs = np.array([40, 23, 40, 1, 17, 33, 1], np.int32)   # item 2: t_x == t_y, item 3: one token, item 6: 1x1
    paths = np.zeros((B, Ty, Tx), np.int32)
    mas.maximum_path_c(paths, values.copy(), t_ys, t_xs)     # the routine accumulates into its `values` argument
    save("mas", values=values, t_ys=t_ys, t_xs=t_xs, paths=paths.astype(np.int8))
    # ---- g2p known answers (vosk_tts/g2p.py; imported by file path: the package itself needs onnxruntime)
    spec = importlib.util.spec_from_file_location("ref_g2p", os.path.join(refimport.REF_ROOT, "vosk_tts", "g2p.py"))
    g2p = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g2p)
    words = ["прив+ет", "м+ир", "абстракцион+истов", "+я", "сво+бодный", "об+ъект", "друзь+я", "ё+лка", "подъ+езд",
             "счастл+ивый", "чт+о", "сег+одня", "пожалуйста", "жизнь", "цирк", "щука", "вьюга", "йогурт", "мя+у",
             "абстр+акция", "белор+усский", "+эхо", "по+эт", "съ+ёмка", "бульон"]
    answers = [g2p.convert(w) for w in words]
    np.savez_compressed(os.path.join(OUT, "g2p.npz"), words=np.array(words), phonemes=np.array(answers))
    print("  g2p.npz")


if __name__ == "__main__":
    main()
