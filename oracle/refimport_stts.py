"""Import the reference's StableTTS / Matcha acoustic model (training/stabletts/matcha) in THIS container.
TEST INFRASTRUCTURE, container-only (like refimport.py): used by gen_golden_stts.py to produce fixtures.

The package drags in lightning / hydra / rich / gdown / torchdiffeq at import time although inference needs none of
them.  Small sys.modules shims stand in for those third-party packages and for the three matcha.utils submodules
that import them; every module that defines arithmetic (models/, utils/model.py, hifigan/) is the reference's own
file, loaded from where it lies under /root/reference.
"""
import importlib
import logging
import os
import sys
import types

REF_ROOT = os.environ.get("VOSK_TTS_REFERENCE", "/root/reference")
REF_STTS = os.path.join(REF_ROOT, "training", "stabletts")

_mods = {}


def _shim(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_shims():
    import torch

    class LightningModule(torch.nn.Module):  # only what MatchaTTS.__init__ touches
        def save_hyperparameters(self, *a, **k):
            pass

    lp_util = _shim("lightning.pytorch.utilities", grad_norm=lambda *a, **k: {}, rank_zero_only=lambda f: f)
    lp = _shim("lightning.pytorch", utilities=lp_util)
    _shim("lightning", LightningModule=LightningModule, pytorch=lp)
    _shim("torchdiffeq", odeint=None)
    # matcha.utils: keep the real directory as the package path, replace the __init__ that imports hydra/rich
    pkg = _shim("matcha.utils", get_pylogger=lambda name=__name__: logging.getLogger(name))
    pkg.__path__ = [os.path.join(REF_STTS, "matcha", "utils")]
    _shim("matcha.utils.pylogger", get_pylogger=pkg.get_pylogger)
    _shim("matcha.utils.utils", plot_tensor=lambda *a, **k: None)
    _shim("matcha.utils.monotonic_align", maximum_path=None)  # training only


def modules():
    """{'matcha_tts', 'hifigan', 'hifigan_cfg', 'AttrDict'} from the reference tree."""
    if _mods:
        return _mods
    if not os.path.isdir(REF_STTS):
        raise RuntimeError("/root/reference is not present (GPU box?) - reference import is container-only")
    if REF_STTS not in sys.path:
        sys.path.insert(0, REF_STTS)
    _install_shims()
    _mods["matcha_tts"] = importlib.import_module("matcha.models.matcha_tts")
    _mods["hifigan"] = importlib.import_module("matcha.hifigan.models")
    _mods["hifigan_cfg"] = importlib.import_module("matcha.hifigan.config").v1
    _mods["AttrDict"] = importlib.import_module("matcha.hifigan.env").AttrDict
    return _mods


def build_reference_model(n_vocab, n_spks, quiet=True):
    """MatchaTTS as training/stabletts/configs/model/matcha.yaml builds it (n_feats 80, spk_emb_dim 128);
    every other size is hard-coded in the reference's own constructors (text_encoder.py:73-95, flow_matching.py:300)."""
    import contextlib
    import io

    M = modules()["matcha_tts"]
    enc = types.SimpleNamespace(encoder_type="RoPE Encoder", encoder_params=types.SimpleNamespace(n_feats=80, n_channels=192))
    cfm = types.SimpleNamespace(name="CFM", solver="euler", sigma_min=1e-4)
    dp = types.SimpleNamespace(name="deterministic")
    with contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext():
        net = M.MatchaTTS(n_vocab=n_vocab, n_spks=n_spks, spk_emb_dim=128, n_feats=80, encoder=enc, duration_predictor=dp,
                          decoder=None, cfm=cfm, data_statistics={"mel_mean": -5.5, "mel_std": 2.1}, out_size=None)
    return net.eval()
