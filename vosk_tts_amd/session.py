"""Drop-in replacement for the onnxruntime.InferenceSession the reference builds at
vosk_tts/model.py:46 and calls at vosk_tts/synth.py:123-126:

    audio = self.model.onnx.run(None, args)[0]

`VitsSession.run(output_names, input_feed)` accepts the same feed dict — "input" int64 [B,T_x],
"input_lengths" int64 [B], "scales" float32 [3], "sid" int64 [B], and None-valued "bert" /
"phone_duration_extra" (synth.py:113-120; onnxruntime skips None feeds) — and returns
[float32 [B,1,1,S]] like the exported graph (training/vits2/onnx_export.py:65-72).
The arithmetic runs in hand-written HIP kernels behind the C ABI (include/vits_mi355.h); there
is no CPU path.  Thread-safe: the gRPC server shares one Synth across worker threads
(server/tts_server.py:39-40) and ctypes releases the GIL during the call.
"""
import itertools
import threading

import numpy as np

from .capi import VitsLib

_GRAPH_INPUTS = ("input", "input_lengths", "scales", "sid")
_OPTIONAL_NONE = ("bert", "phone_duration_extra")
# extension feeds (not part of the ONNX graph) used by parity tests
_EXT = ("vits.noise_dp", "vits.noise_prior", "vits.forced_durations", "vits.seed", "vits.solo", "vits.item_seeds")


class _Arg:
    def __init__(self, name, typ, shape):
        self.name, self.type, self.shape = name, typ, shape


class VitsSession:
    def __init__(self, blob, device=0, lib=None):
        self._lib = lib or VitsLib()
        self._model = self._lib.create(blob, device)
        self.hp = self._model.hp
        self._seed = itertools.count(1)
        self._seed_lock = threading.Lock()

    # -- onnxruntime.InferenceSession surface used by the reference ------------------------
    def get_inputs(self):
        args = [_Arg("input", "tensor(int64)", ["batch_size", "phonemes"]),
                _Arg("input_lengths", "tensor(int64)", ["batch_size"]),
                _Arg("scales", "tensor(float)", [3]),
                _Arg("sid", "tensor(int64)", ["batch_size"])]
        if self.hp.bert_dim > 0:  # BERT-conditioned flavour (vosk_tts/synth.py:88-99)
            args.append(_Arg("bert", "tensor(float)", ["batch_size", self.hp.bert_dim, "phonemes"]))
        return args

    def get_outputs(self):
        return [_Arg("output", "tensor(float)", ["batch_size", 1, 1, "time"])]

    def get_providers(self):
        return ["MI355XExecutionProvider"]

    def _validated(self, output_names, input_feed):
        if output_names is not None and list(output_names) != ["output"]:
            raise ValueError(f"unknown output names {output_names}")
        feed = {k: v for k, v in input_feed.items() if v is not None}  # ORT ignores None feeds
        for k in feed:
            if k == "bert" and self.hp.bert_dim > 0:
                continue  # this graph declares the input (BERT-conditioned VITS flavour)
            if k in _OPTIONAL_NONE:
                raise NotImplementedError(
                    f"feed '{k}': this VITS graph declares no such input (bert: voices whose blob carries enc_p.bert_proj; "
                    "phone_duration_extra: multistream voices only, SURVEY.md §8f rank 2-3)")
            if k not in _GRAPH_INPUTS and k not in _EXT:
                raise ValueError(f"Invalid input name: {k}")
        for k in _GRAPH_INPUTS + (("bert",) if self.hp.bert_dim > 0 else ()):
            if k not in feed and not (k == "sid" and self.hp.n_speakers <= 1):
                raise ValueError(f"Required input {k} is missing")
        ids = np.asarray(feed["input"])
        if ids.ndim != 2:
            raise ValueError("input must be int64 [batch, phonemes] (multistream [B,5,T] flavours are out of scope)")
        B = ids.shape[0]
        sid = np.asarray(feed.get("sid", np.zeros(B, np.int64))).reshape(-1)
        if sid.shape[0] == 1 and B > 1:
            sid = np.repeat(sid, B)
        seed = feed.get("vits.seed")
        if seed is None:
            with self._seed_lock:
                seed = next(self._seed)
        return feed, ids, sid, int(seed)

    def run(self, output_names, input_feed, run_options=None):
        feed, ids, sid, seed = self._validated(output_names, input_feed)
        audio, lengths = self._model.synthesize(
            ids, np.asarray(feed["input_lengths"]).reshape(-1), np.asarray(feed["scales"], np.float32).reshape(-1), sid,
            noise_dp=feed.get("vits.noise_dp"), noise_prior=feed.get("vits.noise_prior"),
            forced_durations=feed.get("vits.forced_durations"), seed=int(seed), solo=bool(feed.get("vits.solo", False)),
            item_seeds=feed.get("vits.item_seeds"), bert=feed.get("bert"))
        self.last_lengths = lengths
        return [audio[:, None, None, :]]

    def run_pcm16(self, input_feed, scale=1.0, return_lengths=False):
        """run() followed by `audio.squeeze() * scale` and Synth.audio_float_to_int16 (vosk_tts/synth.py:127-130), with the
        conversion done on the device (vits_synthesize_pcm16): returns int16 [B, S] -- bit-identical to converting run()'s
        float output with numpy, half the bytes over PCIe.  return_lengths: also return the per-item sample counts
        (concurrent callers must take them from the call, not from the shared `last_lengths` attribute)."""
        feed, ids, sid, seed = self._validated(None, input_feed)
        pcm, lengths = self._model.synthesize_pcm16(
            ids, np.asarray(feed["input_lengths"]).reshape(-1), np.asarray(feed["scales"], np.float32).reshape(-1), sid,
            pcm_scale=float(scale), noise_dp=feed.get("vits.noise_dp"), noise_prior=feed.get("vits.noise_prior"),
            forced_durations=feed.get("vits.forced_durations"), seed=int(seed), solo=bool(feed.get("vits.solo", False)),
            item_seeds=feed.get("vits.item_seeds"), bert=feed.get("bert"))
        self.last_lengths = lengths
        return (pcm, lengths) if return_lengths else pcm

    def run_stream(self, output_names, input_feed, chunk_frames=64):
        """Streaming form of run() for ONE utterance (extension; the reference's transport is already
        `stream AudioChunk`, server/tts_service.proto:46-54): yields float32 [n] chunks of chunk_frames*256
        samples whose concatenation equals run(...)[0].squeeze() for the same feed (same "vits.seed")."""
        feed, ids, sid, seed = self._validated(output_names, input_feed)
        if ids.shape[0] != 1:
            raise ValueError("run_stream takes one utterance")
        if "bert" in feed:
            raise NotImplementedError("streaming of BERT-conditioned voices: vits_stream_open takes no bert feed yet")
        n = int(np.asarray(feed["input_lengths"]).reshape(-1)[0])
        fd = feed.get("vits.forced_durations")
        nd = feed.get("vits.noise_dp")
        return self._model.stream(
            ids[:, :n], np.asarray(feed["scales"], np.float32).reshape(-1), int(sid[0]), chunk_frames=chunk_frames,
            noise_dp=None if nd is None else np.asarray(nd)[:, :, :n], noise_prior=feed.get("vits.noise_prior"),
            forced_durations=None if fd is None else np.asarray(fd)[:, :n], seed=seed)

    def close(self):
        self._model.close()
