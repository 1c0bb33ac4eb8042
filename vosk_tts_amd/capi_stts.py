"""ctypes binding of include/stts_mi355.h (StableTTS / Matcha "multistream" path).  Like capi.VitsLib the class
takes an explicit (library, prefix) so the tests can point it at the CPU oracle; the product default is the HIP
library and there is no CPU fallback."""
import ctypes

import numpy as np

from .capi import VitsError, VitsLib, _f32, _i32, _i64, _p, c_f32p, c_i32p, c_i64p
from .weights_stts import SttsHParams


STTS_FLAG_ITEM_SEEDS = 1  # include/stts_mi355.h


class SttsOpts(ctypes.Structure):
    _fields_ = [("noise", c_f32p), ("noise_stride", ctypes.c_int64), ("seed", ctypes.c_uint64),
                ("n_timesteps", ctypes.c_int32), ("flags", ctypes.c_int32), ("item_seeds", ctypes.POINTER(ctypes.c_uint64))]


class SttsModel:
    """Owns one stts_model* ; `vocoder` is a capi.VitsModel created from a vocoder-only blob (kept alive here)."""

    def __init__(self, vlib: VitsLib, blob, vocoder=None, device=0, prefix=None):
        self.vlib = vlib
        self.prefix = prefix or ("stts_" if vlib.prefix == "vits_" else "sttsref_")
        L = vlib.lib
        f = self._fn
        vp = ctypes.c_void_p
        f("create").argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.POINTER(vp)]
        f("destroy").argtypes = [vp]
        f("destroy").restype = None
        f("last_error").restype = ctypes.c_char_p
        f("get_hparams").argtypes = [vp, ctypes.POINTER(SttsHParams)]
        f("synthesize").argtypes = [vp, c_i64p, ctypes.c_int32, c_f32p, ctypes.c_int64, c_f32p, c_f32p, ctypes.POINTER(SttsOpts),
                                    ctypes.POINTER(c_f32p), c_i64p, ctypes.POINTER(c_f32p), c_i64p]
        f("synthesize_batch").argtypes = [vp, c_i64p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_f32p, c_i64p, c_f32p, c_f32p,
                                          ctypes.POINTER(SttsOpts), ctypes.POINTER(c_f32p), c_i64p, c_i64p]
        if hasattr(L, self.prefix + "stream_open"):
            f("stream_open").argtypes = [vp, c_i64p, ctypes.c_int32, c_f32p, ctypes.c_int64, c_f32p, c_f32p, ctypes.POINTER(SttsOpts),
                                         ctypes.c_int32, ctypes.POINTER(vp), c_i64p]
        f("stage_encoder").argtypes = [vp, c_i64p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_i64p, c_f32p, c_f32p, c_f32p]
        f("stage_durations").argtypes = [vp, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_f32p, c_i32p, c_i64p]
        f("stage_estimator").argtypes = [vp, c_f32p, c_f32p, c_i64p, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_f32p, c_f32p]
        f("stage_cfm").argtypes = [vp, c_f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, c_f32p, ctypes.c_float, ctypes.c_int32,
                                   c_f32p]
        self._vocoder = vocoder
        self._h = vp()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        self.check(f("create")(ctypes.cast(buf, vp), len(blob), vocoder._h if vocoder is not None else None, device,
                               ctypes.byref(self._h)))
        self.hp = SttsHParams()
        self.check(f("get_hparams")(self._h, ctypes.byref(self.hp)))

    def _fn(self, name):
        return getattr(self.vlib.lib, self.prefix + name)

    def check(self, rc):
        if rc != 0:
            raise VitsError(rc, self._fn("last_error")().decode(errors="replace"))

    def close(self):
        if self._h:
            self._fn("destroy")(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- hot path ----------------------------------------------------------------------------
    def synthesize(self, ids, scales, sid, bert=None, phone_duration_extra=None, noise=None, seed=0, n_timesteps=0,
                   want_audio=True, want_mel=True):
        """One utterance.  ids int64 [5,T]; returns (audio float32 [S] or None, mel float32 [n_feats,T_y] or None)."""
        ids = _i64(ids)
        if ids.ndim != 2 or ids.shape[0] != 5:
            raise ValueError("ids must be [5, T]")
        T = ids.shape[1]
        scales = _f32(scales)
        keep = []
        opts = SttsOpts()
        if noise is not None:
            a = _f32(noise); keep.append(a)
            if a.ndim != 2 or a.shape[0] != self.hp.n_feats:
                raise ValueError("noise must be [n_feats, T]")
            opts.noise = _p(a, c_f32p)
            opts.noise_stride = a.shape[1]
        opts.seed = seed
        opts.n_timesteps = n_timesteps
        b = None if bert is None else _f32(bert)
        if b is not None and b.shape != (self.hp.bert_dim, T):
            raise ValueError("bert must be [768, T]")
        p = None if phone_duration_extra is None else _f32(phone_duration_extra)
        if p is not None and p.shape != (T,):
            raise ValueError("phone_duration_extra must be [T]")
        au, mel = c_f32p(), c_f32p()
        ns, nf = ctypes.c_int64(), ctypes.c_int64()
        self.check(self._fn("synthesize")(
            self._h, _p(ids, c_i64p), T, _p(scales, c_f32p), int(sid), None if b is None else _p(b, c_f32p),
            None if p is None else _p(p, c_f32p), ctypes.byref(opts), ctypes.byref(au) if want_audio else None,
            ctypes.byref(ns) if want_audio else None, ctypes.byref(mel) if want_mel else None, ctypes.byref(nf) if want_mel else None))
        free = self.vlib._fn("free_output")
        audio = melo = None
        if want_audio:
            audio = np.ctypeslib.as_array(au, shape=(ns.value,)).copy(); free(au)
        if want_mel:
            melo = np.ctypeslib.as_array(mel, shape=(self.hp.n_feats, nf.value)).copy(); free(mel)
        return audio, melo

    def stream(self, ids, scales, sid, bert=None, phone_duration_extra=None, seed=0, n_timesteps=0, chunk_frames=64):
        """Streaming form of synthesize() (stts_stream_open): a generator of float32 chunks of chunk_frames*hop samples whose
        concatenation equals synthesize()'s audio for the same arguments."""
        if self._vocoder is None:
            raise ValueError("no vocoder attached")
        ids = _i64(ids)
        if ids.ndim != 2 or ids.shape[0] != 5:
            raise ValueError("ids must be [5, T]")
        T = ids.shape[1]
        scales = _f32(scales)
        opts = SttsOpts(); opts.seed = seed; opts.n_timesteps = n_timesteps
        b = None if bert is None else _f32(bert)
        if b is not None and b.shape != (self.hp.bert_dim, T):
            raise ValueError("bert must be [768, T]")
        p = None if phone_duration_extra is None else _f32(phone_duration_extra)
        if p is not None and p.shape != (T,):
            raise ValueError("phone_duration_extra must be [T]")
        st = ctypes.c_void_p()
        total = ctypes.c_int64()
        self.check(self._fn("stream_open")(self._h, _p(ids, c_i64p), T, _p(scales, c_f32p), int(sid), None if b is None else _p(b, c_f32p),
                                           None if p is None else _p(p, c_f32p), ctypes.byref(opts), int(chunk_frames),
                                           ctypes.byref(st), ctypes.byref(total)))
        return self._vocoder._drain(st, chunk_frames)

    def synthesize_batch(self, ids, lengths, scales, sid, bert=None, phone_duration_extra=None, seed=0, n_timesteps=0, item_seeds=None):
        """B independent utterances in one pass (stts_synthesize_batch): ids [B,5,T], lengths [B], sid [B];
        returns (audio float32 [B,S] zero-padded, out_lengths int64 [B] in samples).  Item b equals
        synthesize(ids[b][:, :lengths[b]], ..., seed=item_seeds[b]) (seed + b without item seeds)."""
        ids = _i64(ids); lengths = _i64(lengths); sid = _i64(sid); scales = _f32(scales)
        B, five, T = ids.shape
        b = None if bert is None else _f32(bert)
        p = None if phone_duration_extra is None else _f32(phone_duration_extra)
        opts = SttsOpts(); opts.seed = seed; opts.n_timesteps = n_timesteps
        if item_seeds is not None:
            sd = np.ascontiguousarray(item_seeds, dtype=np.uint64)
            if sd.shape != (B,):
                raise ValueError("item_seeds must be [B]")
            opts.item_seeds = sd.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
            opts.flags |= STTS_FLAG_ITEM_SEEDS  # (the trailing field is only read under this flag: include/stts_mi355.h)
        au = c_f32p(); ns = ctypes.c_int64(); ol = np.zeros(B, np.int64)
        self.check(self._fn("synthesize_batch")(self._h, _p(ids, c_i64p), _p(lengths, c_i64p), B, T, _p(scales, c_f32p), _p(sid, c_i64p),
                                                None if b is None else _p(b, c_f32p), None if p is None else _p(p, c_f32p),
                                                ctypes.byref(opts), ctypes.byref(au), ctypes.byref(ns), _p(ol, c_i64p)))
        audio = np.ctypeslib.as_array(au, shape=(B, ns.value)).copy()
        self.vlib._fn("free_output")(au)
        return audio, ol

    # ---- stages --------------------------------------------------------------------------------
    def encoder(self, ids, lengths, sid, bert=None):
        ids = _i64(ids); lengths = _i64(lengths); sid = _i64(sid)
        B, five, T = ids.shape
        b = None if bert is None else _f32(bert)
        x = np.empty((B, self.hp.enc_hidden, T), np.float32); mu = np.empty((B, self.hp.dp_out, T), np.float32)
        self.check(self._fn("stage_encoder")(self._h, _p(ids, c_i64p), _p(lengths, c_i64p), B, T, _p(sid, c_i64p),
                                             None if b is None else _p(b, c_f32p), _p(x, c_f32p), _p(mu, c_f32p)))
        return x, mu

    def durations(self, mu_dp, length_scale, phone_duration_extra=None):
        mu_dp = _f32(mu_dp)
        B, K, T = mu_dp.shape
        p = None if phone_duration_extra is None else _f32(phone_duration_extra)
        d = np.empty((B, T), np.int32); yl = np.empty(B, np.int64)
        self.check(self._fn("stage_durations")(self._h, _p(mu_dp, c_f32p), B, T, float(length_scale),
                                               None if p is None else _p(p, c_f32p), _p(d, c_i32p), _p(yl, c_i64p)))
        return d, yl

    def estimator(self, x, mu, y_lengths, t, c):
        x = _f32(x); mu = _f32(mu); c = _f32(c); yl = _i64(y_lengths)
        B, NF, T = x.shape
        out = np.empty_like(x)
        self.check(self._fn("stage_estimator")(self._h, _p(x, c_f32p), _p(mu, c_f32p), _p(yl, c_i64p), B, T, float(t), _p(c, c_f32p),
                                               _p(out, c_f32p)))
        return out

    def cfm(self, mu_y, y_length, sid, noise, temperature, n_timesteps=0):
        mu_y = _f32(mu_y); noise = _f32(noise)
        T = mu_y.shape[1]
        out = np.empty((self.hp.n_feats, T), np.float32)
        self.check(self._fn("stage_cfm")(self._h, _p(mu_y, c_f32p), int(y_length), T, int(sid), _p(noise, c_f32p), float(temperature),
                                         int(n_timesteps), _p(out, c_f32p)))
        return out


class BertEncoder:
    """stts_bert_* : the word-embedding BERT encoder (`bert/model.onnx` of the reference, vosk_tts/model.py:61).
    `run(None, {"input_ids": [ids], "attention_mask": [...], "token_type_ids": [...]})[0]` -> float32 [T, hidden]
    mirrors the onnxruntime call at synth.py:27-34."""

    def __init__(self, vlib: VitsLib, blob, device=0, prefix=None):
        from .weights_bert import BertHParams

        self.vlib = vlib
        self.prefix = prefix or ("stts_bert_" if vlib.prefix == "vits_" else "sttsref_bert_")
        vp = ctypes.c_void_p
        f = self._fn
        f("create").argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(vp)]
        f("destroy").argtypes = [vp]
        f("destroy").restype = None
        f("get_hparams").argtypes = [vp, ctypes.POINTER(BertHParams)]
        f("encode").argtypes = [vp, c_i64p, c_i64p, ctypes.c_int32, c_f32p]
        self._err = getattr(vlib.lib, "stts_last_error" if vlib.prefix == "vits_" else "sttsref_last_error")
        self._err.restype = ctypes.c_char_p
        self._h = vp()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        self._check(f("create")(ctypes.cast(buf, vp), len(blob), device, ctypes.byref(self._h)))
        self.hp = BertHParams()
        self._check(f("get_hparams")(self._h, ctypes.byref(self.hp)))

    def _fn(self, name):
        return getattr(self.vlib.lib, self.prefix + name)

    def _check(self, rc):
        if rc != 0:
            raise VitsError(rc, self._err().decode(errors="replace"))

    def encode(self, input_ids, token_type_ids=None):
        ids = _i64(input_ids).reshape(-1)
        ty = None if token_type_ids is None else _i64(token_type_ids).reshape(-1)
        out = np.empty((ids.shape[0], self.hp.hidden), np.float32)
        self._check(self._fn("encode")(self._h, _p(ids, c_i64p), None if ty is None else _p(ty, c_i64p), ids.shape[0], _p(out, c_f32p)))
        return out

    def run(self, output_names, feed):
        mask = np.asarray(feed.get("attention_mask", [[1]]))
        if not np.all(mask == 1):
            raise NotImplementedError("padded BERT batches are not part of the path (synth.py:27-34 encodes one sentence)")
        ids = np.asarray(feed["input_ids"])
        if ids.ndim != 2 or ids.shape[0] != 1:
            raise ValueError("input_ids must be [1, T]")
        ty = feed.get("token_type_ids")
        return [self.encode(ids[0], None if ty is None else np.asarray(ty)[0])]

    def close(self):
        if self._h:
            self._fn("destroy")(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
