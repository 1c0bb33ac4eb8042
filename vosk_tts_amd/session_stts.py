"""Drop-in for the onnxruntime session of a `multistream_v*` voice (vosk_tts/model.py:46, synth.py:113-126):
`SttsSession.run(None, feed)` takes the feed of the StableTTS export (training/stabletts/matcha/onnx/export.py:64-98)
    input [1,5,T] int64, input_lengths [1], scales [3] = [noise_level, 1/speech_rate, duration_noise_level],
    sid [1], bert [1,768,T] or None (-> zeros, synth.py:79), phone_duration_extra [1,T] or None
and returns [wav float32 [1,S], wav_lengths int64 [1]] like the exported graph (export.py:21-32,58-59).
The arithmetic runs in HIP kernels behind include/stts_mi355.h; there is no CPU path."""
import itertools
import threading

import numpy as np

from .capi import VitsLib
from .capi_stts import SttsModel

_INPUTS = ("input", "input_lengths", "scales", "sid", "bert", "phone_duration_extra")
_EXT = ("vits.noise", "vits.seed", "vits.n_timesteps")


class SttsSession:
    def __init__(self, blob, vocoder_blob, device=0, lib=None):
        self._lib = lib or VitsLib()
        self._vocoder = self._lib.create(vocoder_blob, device)
        self._model = SttsModel(self._lib, blob, self._vocoder, device)
        self.hp = self._model.hp
        self._seed = itertools.count(1)
        self._seed_lock = threading.Lock()

    def get_providers(self):
        return ["MI355XExecutionProvider"]

    def _parse(self, output_names, input_feed):
        if output_names is not None and not set(output_names) <= {"wav", "wav_lengths"}:
            raise ValueError(f"unknown output names {output_names}")
        feed = {k: v for k, v in input_feed.items() if v is not None}
        for k in feed:
            if k not in _INPUTS and k not in _EXT:
                raise ValueError(f"Invalid input name: {k}")
        for k in ("input", "input_lengths", "scales"):
            if k not in feed:
                raise ValueError(f"Required input {k} is missing")
        ids = np.asarray(feed["input"])
        if ids.ndim != 3 or ids.shape[0] != 1 or ids.shape[1] != 5:
            raise ValueError("input must be int64 [1, 5, T] (one utterance, five streams)")
        T = ids.shape[2]
        if int(np.asarray(feed["input_lengths"]).reshape(-1)[0]) != T:
            raise ValueError("input_lengths must equal T (the graph is driven with B = 1, synth.py:69-70)")
        sid = int(np.asarray(feed.get("sid", [0])).reshape(-1)[0])
        bert = feed.get("bert")
        if bert is not None:
            bert = np.asarray(bert, np.float32).reshape(self.hp.bert_dim, T)
        pde = feed.get("phone_duration_extra")
        if pde is not None:
            pde = np.asarray(pde, np.float32).reshape(T)
        seed = feed.get("vits.seed")
        if seed is None:
            with self._seed_lock:
                seed = next(self._seed)
        return feed, ids[0], np.asarray(feed["scales"], np.float32).reshape(-1), sid, bert, pde, int(seed)

    def run(self, output_names, input_feed, run_options=None):
        feed, ids, scales, sid, bert, pde, seed = self._parse(output_names, input_feed)
        audio, _ = self._model.synthesize(ids, scales, sid, bert, pde, noise=feed.get("vits.noise"), seed=seed,
                                          n_timesteps=int(feed.get("vits.n_timesteps", 0)), want_mel=False)
        outs = {"wav": audio[None, :], "wav_lengths": np.array([audio.shape[0]], np.int64)}
        return [outs[n] for n in (output_names or ["wav", "wav_lengths"])]

    def run_stream(self, output_names, input_feed, chunk_frames=64):
        """Streaming form of run() (extension; the reference's transport is already `stream AudioChunk`,
        server/tts_service.proto:46-54): yields float32 [n] chunks of chunk_frames * hop samples whose concatenation equals
        run(...)[0].squeeze() for the same feed (same "vits.seed").  The acoustic model runs once, the vocoder is streamed."""
        feed, ids, scales, sid, bert, pde, seed = self._parse(output_names, input_feed)
        if "vits.noise" in feed:
            raise NotImplementedError("run_stream draws the CFM noise on the device (vits.seed)")
        return self._model.stream(ids, scales, sid, bert, pde, seed=seed, n_timesteps=int(feed.get("vits.n_timesteps", 0)),
                                  chunk_frames=chunk_frames)

    def close(self):
        self._model.close()
        self._vocoder.close()
