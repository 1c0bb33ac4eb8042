import os
import subprocess
import sys

import numpy as np
import pytest

# torch bundles its own HIP runtime; import it BEFORE the product library is dlopen'ed so that one
# process holds exactly one libamdhip64 (bench.py does the same: torch is the plumbing for device
# tensors and torch.distributed).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "libvits_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "nightly: long GPU tests (minutes of CPU oracle time); they skip unless VITS_NIGHTLY=1")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from vosk_tts_amd.capi import VitsLib

    src = os.path.join(ROOT, "oracle", "vits_oracle.c")
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    return VitsLib(ORACLE_SO, "vitsref_")


@pytest.fixture(scope="session")
def default_blob():
    from vosk_tts_amd import weights as W

    return W.synthetic_blob(W.default_hparams(), 1234)


@pytest.fixture(scope="session")
def tiny_blob():
    from vosk_tts_amd import weights as W

    return W.synthetic_blob(W.tiny_hparams(), 1234)


@pytest.fixture(scope="session")
def oracle_default(oracle_lib, default_blob):
    return oracle_lib.create(default_blob)


@pytest.fixture(scope="session")
def oracle_tiny(oracle_lib, tiny_blob):
    return oracle_lib.create(tiny_blob)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library (HIP kernels).  Fails loudly if it is not built."""
    from vosk_tts_amd.capi import VitsLib

    return VitsLib()


@pytest.fixture(scope="session")
def hip_default(hip_lib, default_blob):
    return hip_lib.create(default_blob, 0)


@pytest.fixture(scope="session")
def hip_tiny(hip_lib, tiny_blob):
    return hip_lib.create(tiny_blob, 0)


def assert_close(name, ref, got, rtol, atol_frac=None):
    """max|ref-got| <= rtol * max|ref| (the north_star's 'relative fp32 tolerance' is on
    the tensor scale: per-element relative error is meaningless at zero crossings of audio)."""
    ref = np.asarray(ref, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64).reshape(ref.shape)
    assert np.isfinite(got).all(), f"{name}: non-finite values"
    scale = np.abs(ref).max()
    err = np.abs(ref - got).max()
    assert err <= rtol * max(scale, 1e-30), f"{name}: max abs err {err:.3e} vs scale {scale:.3e} (rel {err / max(scale, 1e-30):.3e} > {rtol})"
    return err / max(scale, 1e-30)


def load_reference_mas():
    """the reference's own compiled MAS core (oracle/_ref/mas/core*.so, built by __graft_entry__.build() in the container, where
    /root/reference exists; git-ignored, travels to the GPU box with the snapshot like every prebuilt .so) or None where it was
    never built -- the committed fixture tests/golden/mas.npz pins the kernel either way"""
    import glob
    import importlib.util

    so = glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "mas", "core*.so"))
    if not so:
        return None
    spec = importlib.util.spec_from_file_location("core", so[0])
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError:
        return None
    return mod


def random_mas_cases(seed):
    """[(values float32 [B,Ty,Tx], t_ys int32 [B], t_xs int32 [B])]: ragged extents, t_x == t_y, one token, quantised scores (ties)"""
    import numpy as np

    rng = np.random.default_rng(seed)
    cases = []
    for B, Ty, Tx in ((1, 1, 1), (3, 17, 5), (4, 64, 23), (2, 130, 130), (5, 257, 40)):
        v = rng.standard_normal((B, Ty, Tx)).astype(np.float32)
        if B == 4:
            v = np.round(v * 2) / 2  # exact ties
        t_xs = rng.integers(1, Tx + 1, size=B).astype(np.int32)
        t_ys = np.array([rng.integers(tx, Ty + 1) for tx in t_xs], np.int32)
        t_xs[0], t_ys[0] = Tx, Ty
        cases.append((v, t_ys, t_xs))
    return cases
