#!/usr/bin/env python3
"""tests/golden/full_heavy.npz: the full_c1 case on HEAVY-TAILED weights (vosk_tts_amd.weights.make_synthetic_weights(...,
heavy_sigma=HEAVY_SIGMA): per-row log-normal scales on every weight matrix), run through the REFERENCE's own PyTorch modules.
TEST INFRASTRUCTURE, container-only (needs /root/reference); the committed fixture is data only.

Why: fan-in-scaled uniform weights keep every activation O(1); a weight-normed trained voice does not -- its per-channel scales
differ by an order of magnitude, the WaveNet gates and the exp() of the iSTFT heads (models.py:1043) see heavy tails.  The fp32
(2e-6) and split-bf16 (5e-5) error figures measured on benign weights are re-checked on this fixture.

    python oracle/gen_golden_heavy.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import gen_golden  # noqa: E402
from vosk_tts_amd import weights as W  # noqa: E402

HEAVY_SIGMA = 1.0


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    hp = W.default_hparams()
    net = gen_golden.ref_for(hp, W.make_synthetic_weights(hp, gen_golden.SEED, heavy_sigma=HEAVY_SIGMA))
    rng = np.random.default_rng(777)
    ids = rng.integers(1, hp.n_vocab, size=(1, 24))
    dur = rng.integers(1, 5, size=(1, 24))
    gen_golden.full_case(net, hp, "full_heavy", ids, np.array([24]), np.array([3]), [0.667, 1.0, 0.8], dur, rng)
    g = np.load(os.path.join(gen_golden.OUT, "full_heavy.npz"))
    a = g["audio"]
    print("  audio: max |a| = %.3g, rms = %.3g, finite = %s;  audio_mb max = %.3g;  z max = %.3g" %
          (np.abs(a).max(), np.sqrt((a ** 2).mean()), np.isfinite(a).all(), np.abs(g["audio_mb"]).max(), np.abs(g["z"]).max()))


if __name__ == "__main__":
    main()
