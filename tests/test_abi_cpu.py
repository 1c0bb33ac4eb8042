"""CPU-side checks of the drop-in boundary: the product library loads, exports every symbol the
header declares, and refuses to run without a GPU (no silent CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(header="vits_mi355.h", prefix="vits_"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_the_expected_surface():
    names = _header_functions()
    for must in ("vits_create", "vits_destroy", "vits_synthesize", "vits_free_output", "vits_last_error",
                 "vits_session_create", "vits_session_synthesize_device", "vits_stage_text_encoder",
                 "vits_stage_duration", "vits_stage_regulate", "vits_stage_flow", "vits_stage_decoder", "vits_op_conv1d"):
        assert must in names


def test_product_library_exports_every_declared_symbol(hip_lib):
    missing = [n for n in _header_functions() if not hasattr(hip_lib.lib, n)]
    assert not missing, f"libvits_mi355.so lacks {missing}"
    assert hip_lib.is_device


def test_debug_hooks_live_in_their_own_header_and_are_exported(hip_lib):
    """Round 6: the installed header carries no test hook; include/vits_mi355_debug.h (tests / tools / bench only) declares them all
    and the library exports every one."""
    assert not [n for n in _header_functions() if n.startswith("vits_debug_")]
    hooks = _header_functions("vits_mi355_debug.h")
    assert len([n for n in hooks if n.startswith("vits_debug_")]) >= 17 and "vits_debug_clock_probe" in hooks
    missing = [n for n in hooks if not hasattr(hip_lib.lib, n)]
    assert not missing, f"libvits_mi355.so lacks {missing}"


def test_stts_header_surface_is_exported_by_product_library_and_oracle(hip_lib, oracle_lib):
    """include/stts_mi355.h (StableTTS / Matcha family): every declared entry point exists in libvits_mi355.so, and the
    oracle exports the same surface under sttsref_; the hparams mirror has the header's size."""
    names = _header_functions("stts_mi355.h", "stts_")
    for must in ("stts_create", "stts_destroy", "stts_synthesize", "stts_last_error", "stts_get_hparams", "stts_stage_encoder",
                 "stts_stage_durations", "stts_stage_estimator", "stts_stage_cfm"):
        assert must in names
    assert not [n for n in names if not hasattr(hip_lib.lib, n)]
    # streaming is an extension of the product library: its expected output is the oracle's one-shot audio
    assert not [n for n in names if n != "stts_stream_open" and not hasattr(oracle_lib.lib, "sttsref_" + n[len("stts_"):])]
    from vosk_tts_amd.weights_stts import SttsHParams

    assert ctypes.sizeof(SttsHParams) == 26 * 4


def test_oracle_exports_the_stage_abi(oracle_lib):
    for n in ("create", "destroy", "synthesize", "free_output", "stage_text_encoder", "stage_duration", "stage_regulate",
              "stage_flow", "stage_decoder", "op_conv1d", "algorithmic_flops"):
        assert hasattr(oracle_lib.lib, "vitsref_" + n)
    assert not oracle_lib.is_device


def test_hparams_struct_matches_header():
    from vosk_tts_amd.weights import BlobEntry, HParams

    # 24 scalar ints + 4+4 ups + 1 + 4 resk + 1 + 16 resd + 4 ints + 3 floats + 2 ints + 8 reserved = 71 words
    assert ctypes.sizeof(HParams) == 71 * 4
    assert ctypes.sizeof(BlobEntry) == 96 + 4 + 16 + 4 + 8 + 8


def test_hparams_offsets_of_the_round_2_fields_match_the_header():
    """bert_dim / conv_precision replaced reserved words: their offsets in the ctypes mirror must be where include/vits_mi355.h
    puts them (a C compile of the header's struct), and pack_blob rejects a conv_precision the engine does not know."""
    import os
    import subprocess
    import tempfile

    from vosk_tts_amd import weights as W

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ('#include <stddef.h>\n#include <stdio.h>\n#include "vits_mi355.h"\nint main(void) { printf("%zu %zu %zu\\n", '
           'offsetof(vits_hparams, bert_dim), offsetof(vits_hparams, conv_precision), sizeof(vits_hparams)); return 0; }\n')
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "off.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "off")
        subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", exe, c])
        o_bert, o_prec, size = map(int, subprocess.check_output([exe]).split())
    assert W.HParams.bert_dim.offset == o_bert and W.HParams.conv_precision.offset == o_prec and ctypes.sizeof(W.HParams) == size
    hp = W.default_hparams()
    hp.conv_precision = 2
    with pytest.raises(ValueError, match="conv_precision"):
        W.validate_hparams(hp)


def test_every_ctypes_mirror_has_the_layout_of_its_header_struct(tmp_path):
    """Size and every field offset of each ctypes.Structure the host side passes across the C ABI, against a C compile of
    include/*.h: a field added on one side only (or reordered) shows up here, not as garbage on the device."""
    import os
    import subprocess

    from vosk_tts_amd import capi, capi_stts, weights, weights_bert, weights_stts

    pairs = [(weights.HParams, "vits_hparams"), (weights.BlobEntry, "vits_blob_entry"), (capi.SynthOpts, "vits_synth_opts"), (capi.PersistInfo, "vits_persist_info"),
             (weights_stts.SttsHParams, "stts_hparams"), (capi_stts.SttsOpts, "stts_synth_opts"), (weights_bert.BertHParams, "bert_hparams")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "vits_mi355.h"', '#include "stts_mi355.h"', 'int main(void) {']
    for cls, cname in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, *_ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    c_layout = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cls, cname in pairs:
        assert ctypes.sizeof(cls) == int(c_layout[cname]), cname
        for fname, *_ in cls._fields_:
            assert getattr(cls, fname).offset == int(c_layout[f"{cname}.{fname}"]), f"{cname}.{fname}"


def _header_prototypes():
    """name -> list of argument kinds ('p' pointer, 'f' floating point, 'i' integer) of every function include/*.h declares"""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    protos = {}
    for h in ("vits_mi355.h", "vits_mi355_debug.h", "stts_mi355.h"):
        text = open(os.path.join(root, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
        for m in re.finditer(r"\b(?:int|void|double|const\s+char\s*\*)\s*((?:vits|stts|bert)_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
            args = [a.strip() for a in m.group(2).split(",")]
            if args == ["void"] or args == [""]:
                args = []
            protos[m.group(1)] = ["p" if "*" in a else ("f" if re.match(r"(const\s+)?(float|double)\b", a) else "i") for a in args]
    return protos


def _ctypes_kind(t):
    if t in (ctypes.c_float, ctypes.c_double):
        return "f"
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or issubclass(t, ctypes._Pointer):
        return "p"
    return "i"


def test_ctypes_bindings_agree_with_the_header_prototypes(hip_lib, oracle_lib):
    """Argument count and kind (pointer / integer / floating point) of every binding the Python host side declares, against the
    prototypes in include/*.h (a missing or swapped argument is undefined behaviour that no parity test is guaranteed to catch).
    The stts_* / stts_bert_* bindings are declared when a model object is built: built here on the oracle library, which exports the
    same prototypes under its own prefix."""
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.capi_stts import BertEncoder, SttsModel

    protos = _header_prototypes()
    assert len(protos) >= 40 and "vits_synthesize" in protos and "stts_synthesize" in protos
    voc = oracle_lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234))
    keep = [SttsModel(oracle_lib, S.synthetic_blob(S.default_hparams(40, 7), 1234), voc),
            BertEncoder(oracle_lib, BW.synthetic_blob(BW.small_hparams(120, 64, 2), 1234))]
    checked = set()
    for lib, rename in ((hip_lib, lambda n: n), (oracle_lib, lambda n: n.replace("vits_", "vitsref_", 1).replace("stts_", "sttsref_", 1))):
        for name, kinds in protos.items():
            try:
                fn = getattr(lib.lib, rename(name))
            except AttributeError:
                continue
            if fn.argtypes is None:
                continue
            got = [_ctypes_kind(t) for t in fn.argtypes]
            assert got == kinds, f"{rename(name)}: ctypes {got} vs header {kinds}"
            checked.add(name)
    del keep
    assert len(checked) >= 30 and {"stts_synthesize", "stts_synthesize_batch", "stts_bert_encode", "vits_synthesize_pcm16"} <= checked, sorted(checked)


def test_integration_md_ctypes_stub_is_current(tmp_path, tiny_blob):
    """The reduced binding INTEGRATION.md shows a maintainer is executed as written (library path and blob file made absolute): its
    struct and argument lists must equal the ones vosk_tts_amd/capi.py uses, and on a box without a GPU it must stop at vits_create
    with the library's own error, not crash."""
    import os
    import re

    from vosk_tts_amd import capi

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "lib.vits_synthesize.argtypes" in b)
    blob_path = tmp_path / "model.vitsw"
    blob_path.write_bytes(tiny_blob)
    stub = stub.replace('"vosk_tts_amd/csrc/libvits_mi355.so"', repr(os.path.join(root, "vosk_tts_amd", "csrc", "libvits_mi355.so")))
    stub = stub.replace('"model.vitsw"', repr(str(blob_path)))
    head, tail = stub.split("blob = open(", 1)
    ns = {}
    exec(head, ns)  # imports, struct, argtypes
    assert [(n, t) for n, t in ns["SynthOpts"]._fields_] == [(n, t) for n, t in capi.SynthOpts._fields_]
    ref = capi.VitsLib()
    assert list(ns["lib"].vits_synthesize.argtypes[:7]) == list(ref.lib.vits_synthesize.argtypes[:7]) and len(ns["lib"].vits_synthesize.argtypes) == len(ref.lib.vits_synthesize.argtypes)
    assert list(ns["lib"].vits_create.argtypes) == list(ref.lib.vits_create.argtypes)
    import torch

    if not torch.cuda.is_available():
        with pytest.raises(AssertionError, match="(?i)gpu|device|hip"):
            exec("blob = open(" + tail, ns)


def test_missing_library_fails_loudly(tmp_path):
    from vosk_tts_amd.capi import VitsLib

    with pytest.raises(ImportError, match="no CPU fallback"):
        VitsLib(str(tmp_path / "nope.so"))


def test_no_gpu_means_error_not_fallback(hip_lib, tiny_blob):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vosk_tts_amd.capi import VitsError

    with pytest.raises(VitsError, match="no HIP device"):
        hip_lib.create(tiny_blob, 0)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "vosk_tts_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "libvits_oracle" not in src or f == "capi.py", f
                assert "vits_oracle.c" not in src, f
                assert "import oracle" not in src and "from oracle" not in src, f


@pytest.mark.parametrize("which", ["tiny", "default"])
def test_decoder_ragged_limits_cover_the_oracle_decoders_dependency_cone(hip_lib, oracle_lib, tiny_blob, default_blob, which):
    """The per-layer limits a ragged batch decodes with (engine.hip decoder_needs, exported as host arithmetic by
    vits_debug_decoder_needs) against the ORACLE decoder's measured one-sided dependency: frames of z at or beyond an item's end + k
    are re-drawn and the first len * hop samples compared bit for bit.  The limits must cover every frame that changes a valid
    sample (else a ragged batch differs from the reference's padded batch) and should not cover many more (they are the tiles saved).
    Also: the limits grow monotonically from the waveform back to z, layer by layer, at every stage."""
    m = oracle_lib.create(tiny_blob if which == "tiny" else default_blob)
    hp = m.hp
    need = hip_lib.decoder_needs(hp)
    L, T = 12, 12 + need["z_frames"] + 8
    rng = np.random.default_rng(17)
    z = rng.standard_normal((1, hp.inter_channels, T)).astype(np.float32)
    sid = np.array([0], np.int64)
    base, _ = m.decoder(z, want_mb=False, sid=sid)
    valid = L * hp.hop_length

    def changes(k):  # do frames >= L + k reach the item's own samples?
        z2 = z.copy()
        z2[:, :, L + k:] = rng.standard_normal((1, hp.inter_channels, T - L - k)).astype(np.float32)
        a, _ = m.decoder(z2, want_mb=False, sid=sid)
        return not np.array_equal(a[:, :valid], base[:, :valid])

    assert changes(0) and not changes(need["z_frames"]), "frames beyond the limit reach valid samples"
    lo, hi = 0, need["z_frames"]  # changes(lo), not changes(hi)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        lo, hi = (mid, hi) if changes(mid) else (lo, mid)
    true_frames = hi  # frames L .. L + hi - 1 are needed
    assert true_frames <= need["z_frames"] <= true_frames + 6, (true_frames, need)
    # layer by layer, towards z: every limit at least what its consumer needs, scaled by the stage's rate
    prev = need["post_out"]
    assert need["post_out"] == need["tail_cols"] + 1
    for st, u in zip(reversed(need["stages"]), reversed(list(hp.up_rates[:hp.n_ups]))):
        for c2, c1 in zip(reversed(st["c2_out"]), reversed(st["c1_out"])):
            assert c2 >= prev and c1 > c2, (need, prev)
            prev = c1
        assert st["ups_q"] * u >= prev
        prev = st["ups_q"]
    assert need["pre_out"] > prev and need["z_frames"] == need["pre_out"] + 3
