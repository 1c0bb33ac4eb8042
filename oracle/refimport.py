"""Import the reference's VITS2 model definition (container-only test infrastructure).

TEST INFRASTRUCTURE — not product code.  This module only works where
/root/reference exists (the build container).  It never travels to the GPU box
and nothing under vosk_tts_amd/ imports it.  It is used by
oracle/gen_golden.py to produce the committed fixtures under tests/golden/ and
by the container-only tests that pin the C oracle against the reference.

The reference's training/vits2/models.py imports three packages that are absent
here and are OFF the inference path:
  * librosa.util.{pad_center,tiny,normalize}  (stft.py:32-33; pad_center is
    called in OnnxSTFT.__init__, stft.py:206, with size == win_length == 16,
    where a real centre-pad is the identity)
  * torchaudio.transforms.Spectrogram          (models.py:1291, discriminators)
  * monotonic_align                            (models.py:10, training-only MAS)
They are replaced by minimal sys.modules shims.  No reference source is copied.
"""
import json
import os
import sys
import types

REF_ROOT = "/root/reference"
REF_VITS2 = os.path.join(REF_ROOT, "training", "vits2")
REF_CONFIG = os.path.join(REF_VITS2, "configs", "mb_istft_vits2_multi.json")


def have_reference():
    return os.path.isdir(REF_VITS2)


def _install_shims():
    import numpy as np

    if "librosa" not in sys.modules:
        librosa = types.ModuleType("librosa")
        util = types.ModuleType("librosa.util")

        def pad_center(data, size, axis=-1, **kw):
            n = data.shape[axis]
            lpad = int((size - n) // 2)
            lengths = [(0, 0)] * data.ndim
            lengths[axis] = (lpad, int(size - n - lpad))
            if lpad < 0:
                raise ValueError("target size smaller than input")
            return np.pad(data, lengths, **kw)

        def tiny(x):
            return np.finfo(np.float32).tiny

        def normalize(S, **kw):
            return S

        util.pad_center = pad_center
        util.tiny = tiny
        util.normalize = normalize
        librosa.util = util
        sys.modules["librosa"] = librosa
        sys.modules["librosa.util"] = util
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        tr = types.ModuleType("torchaudio.transforms")

        class Spectrogram:  # discriminators only; never constructed on the inference path
            def __init__(self, *a, **k):
                raise RuntimeError("torchaudio shim: Spectrogram is not available")

        tr.Spectrogram = Spectrogram
        ta.transforms = tr
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = tr
    if "monotonic_align" not in sys.modules:
        ma = types.ModuleType("monotonic_align")

        def maximum_path(*a, **k):
            raise RuntimeError("monotonic_align shim: training-only")

        ma.maximum_path = maximum_path
        sys.modules["monotonic_align"] = ma


_ref = {}


def ref_modules():
    """Returns dict of imported reference modules (models, modules, attentions, ...)."""
    if _ref:
        return _ref
    if not have_reference():
        raise RuntimeError("/root/reference is not present (GPU box?) — reference import is container-only")
    _install_shims()
    if REF_VITS2 not in sys.path:
        sys.path.insert(0, REF_VITS2)
    import importlib

    for name in ("commons", "transforms", "modules", "attentions", "pqmf", "stft", "models"):
        _ref[name] = importlib.import_module(name)
    return _ref


def ref_config():
    with open(REF_CONFIG) as f:
        return json.load(f)


def build_reference_model(n_vocab=62, cfg=None, quiet=True):
    """SynthesizerTrn as the exporter builds it (onnx_export.py:47-53,77-80):
    is_onnx=True, eval(), weight-norm removed from dec and flow."""
    import contextlib
    import io

    import torch

    m = ref_modules()
    cfg = cfg or ref_config()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf) if quiet else contextlib.nullcontext():
        net = m["models"].SynthesizerTrn(
            n_vocab,
            80,
            cfg["train"]["segment_size"] // cfg["data"]["hop_length"],
            n_speakers=cfg["data"]["n_speakers"],
            is_onnx=True,
            **cfg["model"],
        )
        net.eval()
        with torch.no_grad():
            net.dec.remove_weight_norm()
            net.flow.remove_weight_norm()
    return net


def load_into_reference(net, tensors):
    """Push build-owned synthetic tensors (vosk_tts_amd.weights) into the reference
    module by state_dict name; every inference-path parameter must be covered."""
    import torch

    sd = net.state_dict()
    with torch.no_grad():
        for name, arr in tensors.items():
            if name not in sd:
                raise KeyError(f"reference has no tensor {name}")
            if tuple(sd[name].shape) != tuple(arr.shape):
                raise ValueError(f"{name}: ref {tuple(sd[name].shape)} vs ours {arr.shape}")
            sd[name].copy_(torch.from_numpy(arr))
    skip_prefix = ("enc_q.", "dp.post_", "dp.flows.1.", "dec.stft.")
    missing = [k for k in sd if k not in tensors and not k.startswith(skip_prefix)]
    if missing:
        raise KeyError(f"inference tensors not covered by the blob: {missing[:5]} ...")


def run_reference_stages(net, ids, lengths, sid, scales, noise_dp, noise_prior_fn, forced_durations=None):
    """SynthesizerTrn.infer (models.py:1679-1704) executed stage by stage on the
    reference's own sub-modules so intermediate tensors can be captured and the
    two randn draws (models.py:96, :1700) / w_ceil (:1690) can be injected.

    noise_dp: [B,2,T_x] unit normal; noise_prior_fn(shape)->tensor unit normal.
    Returns dict of numpy arrays.
    """
    import torch

    m = ref_modules()
    commons = m["commons"]
    noise_scale, length_scale, noise_scale_w = [float(s) for s in scales]
    out = {}
    with torch.no_grad():
        x_ids = torch.as_tensor(ids, dtype=torch.long)
        x_lengths = torch.as_tensor(lengths, dtype=torch.long)
        sid_t = torch.as_tensor(sid, dtype=torch.long)
        g = net.emb_g(sid_t).unsqueeze(-1)                                   # :1680-1681
        x, m_p, logs_p, x_mask = net.enc_p(x_ids, x_lengths, g=g)            # :1684
        out["x"], out["m_p_tok"], out["logs_p_tok"] = x.numpy(), m_p.numpy(), logs_p.numpy()
        # dp reverse with injected noise: mirror models.py:56-63,93-101
        dp = net.dp
        orig_randn = torch.randn
        nd = torch.as_tensor(noise_dp, dtype=torch.float32)

        def fake_randn(*a, **k):
            return nd.clone()

        torch.randn = fake_randn
        try:
            logw = dp(x, x_mask, g=g, reverse=True, noise_scale=noise_scale_w)  # :1686
        finally:
            torch.randn = orig_randn
        out["logw"] = logw.numpy()
        w = torch.exp(logw) * x_mask * length_scale                          # :1689
        w_ceil = torch.ceil(w)                                               # :1690
        out["w_ceil_free"] = w_ceil.numpy()
        if forced_durations is not None:
            w_ceil = torch.as_tensor(forced_durations, dtype=torch.float32).view_as(w_ceil) * x_mask
        y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()     # :1691
        y_mask = torch.unsqueeze(commons.sequence_mask(y_lengths, None), 1).to(x_mask.dtype)
        attn_mask = torch.unsqueeze(x_mask, 2) * torch.unsqueeze(y_mask, -1)
        attn = commons.generate_path(w_ceil, attn_mask)                      # :1694
        m_p = torch.matmul(attn.squeeze(1), m_p.transpose(1, 2)).transpose(1, 2)
        logs_p = torch.matmul(attn.squeeze(1), logs_p.transpose(1, 2)).transpose(1, 2)
        eps = noise_prior_fn(tuple(m_p.shape))
        z_p = m_p + eps * torch.exp(logs_p) * noise_scale                    # :1700
        out["durations"] = w_ceil.squeeze(1).numpy().astype("int32")
        out["y_lengths"] = y_lengths.numpy()
        out["m_p"], out["logs_p"], out["z_p"] = m_p.numpy(), logs_p.numpy(), z_p.numpy()
        out["noise_prior"] = eps.numpy()
        # flow reverse, layer by layer (models.py:755-756)
        z = z_p
        taps = []
        for flow in reversed(net.flow.flows):
            z = flow(z, y_mask, g=g, reverse=True)
            taps.append(z.numpy().copy())
        out["flow_taps"] = taps
        out["z"] = z.numpy()
        zin = (z * y_mask)                                                   # :1703
        o, o_mb = net.dec(zin, g=g)
        out["audio"], out["audio_mb"] = o.numpy(), o_mb.numpy()
        out["y_mask"] = y_mask.numpy()
    return out


def build_reference_mas():
    """Compiles the reference's own Cython MAS core (training/vits2/monotonic_align/core.pyx) from where it lies
    under /root/reference into oracle/_ref/mas/ (git-ignored; the built module travels to the GPU box with the snapshot like the
    other prebuilt .so files) with the installed Cython + gcc, and returns the module (exposes maximum_path_c).  Building is
    container-only (needs /root/reference); used by gen_golden.py and, where the module exists, by the MAS tests."""
    import importlib.util
    import subprocess
    import sysconfig

    if not have_reference():
        raise RuntimeError("/root/reference is not present")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "mas")
    os.makedirs(out, exist_ok=True)
    pyx = os.path.join(REF_VITS2, "monotonic_align", "core.pyx")
    so = os.path.join(out, "core" + sysconfig.get_config_var("EXT_SUFFIX"))
    stale = os.path.join(out, "core.c")  # (earlier rounds left Cython's C output here: it quotes the .pyx text in comments)
    if os.path.exists(stale):
        os.remove(stale)
    if not os.path.exists(so):
        import tempfile

        # only the compiled module is kept under oracle/_ref: Cython's generated C quotes the reference's .pyx source as comments,
        # so it lives in a temporary directory for the length of the gcc call
        with tempfile.TemporaryDirectory() as td:
            c_file = os.path.join(td, "core.c")
            subprocess.check_call([sys.executable, "-m", "cython", "-3", pyx, "-o", c_file])
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-fopenmp", "-I" + sysconfig.get_paths()["include"], c_file, "-o", so])
    spec = importlib.util.spec_from_file_location("core", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
