"""ctypes binding of the C ABI declared in include/vits_mi355.h.

`VitsLib()` loads the product library (vosk_tts_amd/csrc/libvits_mi355.so, the
hand-written HIP kernels) and fails loudly when it is missing — there is no CPU
fallback in the product path.  The class takes an explicit (path, prefix) only
so that tests can drive the CPU oracle (oracle/libvits_oracle.so, prefix
"vitsref_") through the very same wrapper and compare results.
"""
import ctypes
import os

import numpy as np

from .weights import HParams

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libvits_mi355.so")

c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f32p = ctypes.POINTER(ctypes.c_float)


class SynthOpts(ctypes.Structure):
    """mirror of `struct vits_synth_opts`"""

    _fields_ = [
        ("noise_dp", c_f32p),
        ("noise_prior", c_f32p),
        ("noise_prior_stride", ctypes.c_int64),
        ("forced_durations", c_i32p),
        ("seed", ctypes.c_uint64),
        ("max_frames", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("item_seeds", ctypes.POINTER(ctypes.c_uint64)),
        ("bert", c_f32p),
    ]


class PersistInfo(ctypes.Structure):
    """mirror of `struct vits_persist_info` (vits_persist_state)"""

    _fields_ = [(n, ctypes.c_int32) for n in ("configured_mask", "active_mask", "off_for_ms", "timeouts", "rearms", "launches",
                                               "process_owns_device", "reserved")]


class VitsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vits error {code}: {msg}")
        self.code = code


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


EXPORTED = [
    "create", "destroy", "last_error", "get_hparams", "is_device_backend", "synthesize", "free_output",
    "stage_text_encoder", "stage_duration", "stage_regulate", "stage_flow", "stage_decoder", "op_conv1d",
    "algorithmic_flops",
]
DEVICE_ONLY = ["session_create", "session_destroy", "session_synthesize_device", "session_last_ms"]


class VitsLib:
    def __init__(self, path=None, prefix="vits_"):
        path = path or os.environ.get("VITS_MI355_LIB", DEFAULT_LIB)
        if not os.path.exists(path):
            raise ImportError(
                f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback in the product path."
            )
        self.path = path
        self.prefix = prefix
        self.lib = ctypes.CDLL(path)
        L = self.lib
        f = self._fn
        f("create").argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        f("destroy").argtypes = [ctypes.c_void_p]
        f("destroy").restype = None
        f("last_error").restype = ctypes.c_char_p
        f("get_hparams").argtypes = [ctypes.c_void_p, ctypes.POINTER(HParams)]
        f("synthesize").argtypes = [ctypes.c_void_p, c_i64p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_f32p, c_i64p,
                                    ctypes.POINTER(SynthOpts), ctypes.POINTER(c_f32p), c_i64p, c_i64p]
        f("free_output").argtypes = [c_f32p]
        f("free_output").restype = None
        f("stage_text_encoder").argtypes = [ctypes.c_void_p, c_i64p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_i64p,
                                            c_f32p, c_f32p, c_f32p]
        f("stage_duration").argtypes = [ctypes.c_void_p, c_f32p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_i64p, c_f32p,
                                        ctypes.c_float, c_f32p]
        f("stage_regulate").argtypes = [ctypes.c_void_p, c_f32p, c_i32p, c_i64p, ctypes.c_int32, ctypes.c_int32,
                                        ctypes.c_float, c_f32p, c_f32p, c_f32p, ctypes.c_float, ctypes.c_int32, c_i32p,
                                        c_i64p, c_f32p]
        f("stage_flow").argtypes = [ctypes.c_void_p, c_f32p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_i64p, c_f32p]
        f("stage_decoder").argtypes = [ctypes.c_void_p, c_f32p, ctypes.c_int32, ctypes.c_int32, c_i64p, c_f32p, c_f32p]
        f("op_conv1d").argtypes = [ctypes.c_int, c_f32p, c_f32p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_f32p]
        f("algorithmic_flops").argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        f("algorithmic_flops").restype = ctypes.c_double
        f("mas_maximum_path").argtypes = [ctypes.c_int, c_f32p, c_i32p, c_i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          c_i32p]
        self.is_device = bool(f("is_device_backend")())
        if self.is_device:
            c_i16p = ctypes.POINTER(ctypes.c_int16)
            f("synthesize_pcm16").argtypes = [ctypes.c_void_p, c_i64p, c_i64p, ctypes.c_int32, ctypes.c_int32, c_f32p, c_i64p,
                                              ctypes.POINTER(SynthOpts), ctypes.c_float, ctypes.POINTER(c_i16p), c_i64p, c_i64p]
            f("free_pcm16").argtypes = [c_i16p]
            f("free_pcm16").restype = None
            f("session_set_sdp_always").argtypes = [ctypes.c_void_p, ctypes.c_int]
            f("session_create").argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                            ctypes.POINTER(ctypes.c_void_p)]
            f("session_destroy").argtypes = [ctypes.c_void_p]
            f("session_destroy").restype = None
            f("session_synthesize_device").argtypes = [
                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, c_f32p,
                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64,
                ctypes.c_void_p]
            f("session_last_ms").argtypes = [ctypes.c_void_p, c_f32p]
            f("debug_decoder_needs").argtypes = [ctypes.POINTER(HParams), c_i32p, ctypes.c_int32]
            f("stream_open").argtypes = [ctypes.c_void_p, c_i64p, ctypes.c_int32, c_f32p, ctypes.c_int64,
                                         ctypes.POINTER(SynthOpts), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), c_i64p]
            f("stream_open_latent").argtypes = [ctypes.c_void_p, c_f32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32,
                                                ctypes.POINTER(ctypes.c_void_p), c_i64p]
            f("stream_next").argtypes = [ctypes.c_void_p, c_f32p, ctypes.c_int64, c_i64p]
            f("stream_close").argtypes = [ctypes.c_void_p]
            f("stream_close").restype = None
            f("persist_state").argtypes = [ctypes.c_void_p, ctypes.POINTER(PersistInfo)]
            if self.has("debug_clock_probe"):
                f("debug_clock_probe").argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.c_int32]

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def device_count(self):
        """HIP devices visible to this process (0 for the CPU oracle)"""
        if not self.is_device or not self.has("device_count"):
            return 0
        self._fn("device_count").restype = ctypes.c_int
        return int(self._fn("device_count")())

    def has(self, name):
        return hasattr(self.lib, self.prefix + name)

    def check(self, rc):
        if rc != 0:
            raise VitsError(rc, self._fn("last_error")().decode(errors="replace"))

    def create(self, blob, device=0):
        return VitsModel(self, blob, device)

    def clock_probe(self, device=0, duration_us=20000, n=64):
        """vits_debug_clock_probe (include/vits_mi355_debug.h): shader clock in GHz seen by `n` one-wave workgroups that sit on the device
        for `duration_us` NEXT to whatever the caller has in flight on other streams -> sorted list.  Blocks for the duration."""
        out = (ctypes.c_double * n)()
        rc = self._fn("debug_clock_probe")(device, duration_us, out, n)
        if rc < 0:
            raise VitsError(-rc, self._fn("last_error")().decode(errors="replace"))
        return sorted(float(v) for v in out[:rc])

    def decoder_needs(self, hp):
        """vits_debug_decoder_needs (host arithmetic, no device): the decoder's per-layer limits in a ragged batch for the hparams struct
        `hp` -> dict(z_frames, pre_out, post_out, tail_cols, stages=[dict(ups_q, c1_out=[...], c2_out=[...])])"""
        buf = np.zeros(4 + 4 * 9, np.int32)
        n = self._fn("debug_decoder_needs")(ctypes.byref(hp), _p(buf, c_i32p), buf.shape[0])
        if n < 0:
            raise VitsError(-n, self._fn("last_error")().decode(errors="replace"))
        v = buf[:n].tolist()
        out = dict(z_frames=v[0], pre_out=v[1], post_out=v[2], tail_cols=v[3], stages=[])
        k, nd = 4, hp.n_resd
        for _ in range(hp.n_ups):
            out["stages"].append(dict(ups_q=v[k], c1_out=v[k + 1:k + 1 + nd], c2_out=v[k + 1 + nd:k + 1 + 2 * nd]))
            k += 1 + 2 * nd
        return out

    def mas_maximum_path(self, values, t_ys, t_xs, device=0):
        """monotonic_align.maximum_path_c (core.pyx:35-42): values float32 [B,T_y,T_x] -> paths int32 [B,T_y,T_x]."""
        values = _f32(values)
        if values.ndim != 3:
            raise ValueError("values must be [B,T_y,T_x]")
        B, Ty, Tx = values.shape
        t_ys = _i32(t_ys).reshape(-1); t_xs = _i32(t_xs).reshape(-1)
        if t_ys.shape != (B,) or t_xs.shape != (B,):
            raise ValueError("t_ys/t_xs must be [B]")
        paths = np.empty((B, Ty, Tx), np.int32)
        self.check(self._fn("mas_maximum_path")(device, _p(values, c_f32p), _p(t_ys, c_i32p), _p(t_xs, c_i32p), B, Ty, Tx,
                                                _p(paths, c_i32p)))
        return paths


class VitsModel:
    """Owns one vits_model* (weights resident on one device)."""

    def __init__(self, lib, blob, device=0):
        self.lib = lib
        self._h = ctypes.c_void_p()
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob)
        lib.check(lib._fn("create")(ctypes.cast(buf, ctypes.c_void_p), len(blob), device, ctypes.byref(self._h)))
        self.hp = HParams()
        lib.check(lib._fn("get_hparams")(self._h, ctypes.byref(self.hp)))
        self.device = device

    def persist_state(self, with_device=True):
        """vits_persist_state: dict of the persistent-program state of this process (masks, timeouts, re-arms, completed launches).
        with_device=False: process-wide counters only (no device synchronisation; `launches` / `process_owns_device` are -1)."""
        info = PersistInfo()
        self.lib.check(self.lib._fn("persist_state")(self._h if with_device else None, ctypes.byref(info)))
        return {n: int(getattr(info, n)) for n, _ in PersistInfo._fields_ if n != "reserved"}

    def close(self):
        if self._h:
            self.lib._fn("destroy")(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _opts(self, B, Tx, noise_dp, noise_prior, forced_durations, seed, max_frames, solo=False, item_seeds=None, bert=None):
        opts = SynthOpts()
        keep = []
        if bert is not None:
            a = _f32(bert); keep.append(a)
            if a.shape != (B, self.hp.bert_dim, Tx):
                raise ValueError(f"bert must be [B, {self.hp.bert_dim}, T_x] (got {a.shape})")
            opts.bert = _p(a, c_f32p)
        if item_seeds is not None:
            a = np.ascontiguousarray(item_seeds, dtype=np.uint64); keep.append(a)
            if a.shape != (B,) or not solo:
                raise ValueError("item_seeds must be [B] and needs solo=True")
            opts.item_seeds = a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        if noise_dp is not None:
            a = _f32(noise_dp); keep.append(a)
            if a.shape != (B, 2, Tx):
                raise ValueError("noise_dp must be [B,2,T_x]")
            opts.noise_dp = _p(a, c_f32p)
        if noise_prior is not None:
            a = _f32(noise_prior); keep.append(a)
            if a.ndim != 3 or a.shape[0] != B or a.shape[1] != self.hp.inter_channels:
                raise ValueError("noise_prior must be [B,inter,T]")
            opts.noise_prior = _p(a, c_f32p)
            opts.noise_prior_stride = a.shape[2]
        if forced_durations is not None:
            a = _i32(forced_durations); keep.append(a)
            if a.shape != (B, Tx):
                raise ValueError("forced_durations must be [B,T_x]")
            opts.forced_durations = _p(a, c_i32p)
        opts.seed = seed
        opts.max_frames = max_frames
        opts.flags = 1 if solo else 0  # VITS_FLAG_SOLO_BATCH
        return opts, keep

    # ---- the hot path -----------------------------------------------------
    def synthesize(self, ids, lengths, scales, sid, noise_dp=None, noise_prior=None, forced_durations=None, seed=0,
                   max_frames=0, solo=False, item_seeds=None, bert=None):
        """One .run(): returns (audio float32 [B,S], out_lengths int64 [B]).  solo=True (VITS_FLAG_SOLO_BATCH): every
        item equals its own single-utterance call with seed + b instead of the reference's padded-batch result."""
        ids = _i64(ids)
        B, Tx = ids.shape
        lengths = _i64(lengths)
        sid = _i64(sid)
        scales = _f32(scales)
        if lengths.shape != (B,) or sid.shape != (B,) or scales.shape != (3,):
            raise ValueError("bad feed shapes")
        opts, keep = self._opts(B, Tx, noise_dp, noise_prior, forced_durations, seed, max_frames, solo, item_seeds, bert)
        out = c_f32p()
        ns = ctypes.c_int64()
        olen = np.zeros(B, dtype=np.int64)
        self.lib.check(self.lib._fn("synthesize")(self._h, _p(ids, c_i64p), _p(lengths, c_i64p), B, Tx, _p(scales, c_f32p),
                                                  _p(sid, c_i64p), ctypes.byref(opts), ctypes.byref(out),
                                                  ctypes.byref(ns), _p(olen, c_i64p)))
        try:
            audio = np.ctypeslib.as_array(out, shape=(B, ns.value)).copy()
        finally:
            self.lib._fn("free_output")(out)
        return audio, olen

    def synthesize_pcm16(self, ids, lengths, scales, sid, pcm_scale=1.0, noise_dp=None, noise_prior=None, forced_durations=None,
                         seed=0, max_frames=0, solo=False, item_seeds=None, bert=None):
        """synthesize() with Synth.synth_audio's `* scale` and audio_float_to_int16 (vosk_tts/synth.py:127-130) done on the
        device: returns (pcm int16 [B,S], out_lengths int64 [B])."""
        ids = _i64(ids)
        B, Tx = ids.shape
        lengths = _i64(lengths); sid = _i64(sid); scales = _f32(scales)
        if lengths.shape != (B,) or sid.shape != (B,) or scales.shape != (3,):
            raise ValueError("bad feed shapes")
        opts, keep = self._opts(B, Tx, noise_dp, noise_prior, forced_durations, seed, max_frames, solo, item_seeds, bert)
        out = ctypes.POINTER(ctypes.c_int16)()
        ns = ctypes.c_int64()
        olen = np.zeros(B, dtype=np.int64)
        self.lib.check(self.lib._fn("synthesize_pcm16")(self._h, _p(ids, c_i64p), _p(lengths, c_i64p), B, Tx, _p(scales, c_f32p),
                                                        _p(sid, c_i64p), ctypes.byref(opts), float(pcm_scale), ctypes.byref(out),
                                                        ctypes.byref(ns), _p(olen, c_i64p)))
        try:
            pcm = np.ctypeslib.as_array(out, shape=(B, ns.value)).copy()
        finally:
            self.lib._fn("free_pcm16")(out)
        return pcm, olen

    def stream(self, ids, scales, sid, chunk_frames=64, noise_dp=None, noise_prior=None, forced_durations=None, seed=0, bert=None):
        """Streaming synthesis of ONE utterance (vits_stream_*): a generator of float32 chunks of
        chunk_frames*hop_length samples (the last one shorter); their concatenation equals synthesize()."""
        if not self.lib.has("stream_open"):
            raise VitsError(-1, "this backend has no streaming entry points")
        ids = _i64(ids).reshape(1, -1)
        Tx = ids.shape[1]
        scales = _f32(scales)
        opts, keep = self._opts(1, Tx, noise_dp, noise_prior, forced_durations, seed, 0, bert=bert)
        L = self.lib
        st = ctypes.c_void_p()
        total = ctypes.c_int64()
        L.check(L._fn("stream_open")(self._h, _p(ids, c_i64p), Tx, _p(scales, c_f32p), int(sid), ctypes.byref(opts),
                                     int(chunk_frames), ctypes.byref(st), ctypes.byref(total)))
        return self._drain(st, chunk_frames)

    def stream_latent(self, z, chunk_frames=64, clamp=False):
        """Streams the decoder over a latent the caller holds (vits_stream_open_latent): z float32 [inter_channels, T_y]."""
        z = _f32(z)
        if z.ndim != 2 or z.shape[0] != self.hp.inter_channels:
            raise ValueError("z must be [inter_channels, T_y]")
        L = self.lib
        st = ctypes.c_void_p()
        total = ctypes.c_int64()
        L.check(L._fn("stream_open_latent")(self._h, _p(z, c_f32p), z.shape[1], int(chunk_frames), 1 if clamp else 0,
                                            ctypes.byref(st), ctypes.byref(total)))
        return self._drain(st, chunk_frames)

    def _drain(self, st, chunk_frames):
        """generator over an open vits_stream; closes it when exhausted or dropped"""
        L = self.lib
        cap = int(chunk_frames) * self.hp.hop_length
        n = ctypes.c_int64()
        try:
            while True:
                buf = np.empty(cap, np.float32)
                L.check(L._fn("stream_next")(st, _p(buf, c_f32p), cap, ctypes.byref(n)))
                if n.value == 0:
                    break
                yield buf[:n.value]
        finally:
            L._fn("stream_close")(st)

    # ---- stage-level entry points (parity tests) ---------------------------
    def text_encoder(self, ids, lengths, sid):
        ids = _i64(ids); lengths = _i64(lengths); sid = _i64(sid)
        B, T = ids.shape
        H, I = self.hp.hidden_channels, self.hp.inter_channels
        x = np.empty((B, H, T), np.float32); m_p = np.empty((B, I, T), np.float32); logs_p = np.empty((B, I, T), np.float32)
        self.lib.check(self.lib._fn("stage_text_encoder")(self._h, _p(ids, c_i64p), _p(lengths, c_i64p), B, T,
                                                          _p(sid, c_i64p), _p(x, c_f32p), _p(m_p, c_f32p), _p(logs_p, c_f32p)))
        return x, m_p, logs_p

    def duration(self, x, lengths, sid, noise, noise_scale_w):
        x = _f32(x); lengths = _i64(lengths); sid = _i64(sid); noise = _f32(noise)
        B, _, T = x.shape
        logw = np.empty((B, T), np.float32)
        self.lib.check(self.lib._fn("stage_duration")(self._h, _p(x, c_f32p), _p(lengths, c_i64p), B, T, _p(sid, c_i64p),
                                                      _p(noise, c_f32p), noise_scale_w, _p(logw, c_f32p)))
        return logw

    def regulate(self, logw, forced, lengths, length_scale, m_p, logs_p, noise, noise_scale, T_cap):
        lengths = _i64(lengths)
        B = lengths.shape[0]
        logw = _f32(logw) if logw is not None else None
        forced = _i32(forced) if forced is not None else None
        T = (logw if logw is not None else forced).shape[-1]
        m_p = _f32(m_p); logs_p = _f32(logs_p)
        noise = _f32(noise) if noise is not None else None
        I = self.hp.inter_channels
        dur = np.zeros((B, T), np.int32); ylen = np.zeros(B, np.int64)
        z_p = np.zeros((B, I, T_cap), np.float32)
        self.lib.check(self.lib._fn("stage_regulate")(self._h, _p(logw, c_f32p), _p(forced, c_i32p), _p(lengths, c_i64p), B, T,
                                                      length_scale, _p(m_p, c_f32p), _p(logs_p, c_f32p), _p(noise, c_f32p),
                                                      noise_scale, T_cap, _p(dur, c_i32p), _p(ylen, c_i64p), _p(z_p, c_f32p)))
        return dur, ylen, z_p

    def flow(self, z_p, y_lengths, sid):
        z_p = _f32(z_p); y_lengths = _i64(y_lengths); sid = _i64(sid)
        B, _, T = z_p.shape
        z = np.empty_like(z_p)
        self.lib.check(self.lib._fn("stage_flow")(self._h, _p(z_p, c_f32p), _p(y_lengths, c_i64p), B, T, _p(sid, c_i64p),
                                                  _p(z, c_f32p)))
        return z

    def decoder(self, z, want_mb=True, sid=None):
        z = _f32(z)
        B, _, T = z.shape
        sid = _i64(sid) if sid is not None else None
        hop = self.hp.hop_length
        audio = np.empty((B, T * hop), np.float32)
        mb = None
        if want_mb and self.hp.dec_type == 0:
            mb = np.empty((B, self.hp.subbands, T * hop // self.hp.subbands), np.float32)
        self.lib.check(self.lib._fn("stage_decoder")(self._h, _p(z, c_f32p), B, T, _p(sid, c_i64p), _p(audio, c_f32p), _p(mb, c_f32p)))
        return audio, mb

    def algorithmic_flops(self, B, Tx, Ty):
        return float(self.lib._fn("algorithmic_flops")(self._h, B, Tx, Ty))


def op_conv1d(lib, x, w, bias, dilation=1, lrelu_slope=1.0, device=0):
    x = _f32(x); w = _f32(w)
    bias = _f32(bias) if bias is not None else None
    B, Cin, T = x.shape
    Cout, Cin2, K = w.shape
    assert Cin == Cin2
    y = np.empty((B, Cout, T), np.float32)
    lib.check(lib._fn("op_conv1d")(device, _p(x, c_f32p), _p(w, c_f32p), _p(bias, c_f32p), B, Cin, Cout, T, K, dilation,
                                   lrelu_slope, _p(y, c_f32p)))
    return y


class VitsDeviceSession:
    """Device-resident serving loop (vits_session_*): inputs and outputs are raw HBM pointers
    (e.g. torch tensors' data_ptr()); the forward runs asynchronously on the session's own HIP
    stream and is replayed as a cached hipGraph."""

    def __init__(self, model, max_B, max_Tx, max_Ty):
        self.model = model
        self.lib = model.lib
        if not self.lib.is_device:
            raise RuntimeError("device sessions exist only in the HIP library")
        self._h = ctypes.c_void_p()
        self.lib.check(self.lib._fn("session_create")(model._h, max_B, max_Tx, max_Ty, ctypes.byref(self._h)))
        L = self.lib
        L._fn("session_sync").argtypes = [ctypes.c_void_p]
        L._fn("session_set_options").argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L._fn("session_profile_report").argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]

    def close(self):
        if self._h:
            self.lib._fn("session_destroy")(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synthesize_device(self, d_ids, d_lengths, B, Tx, scales, d_sid, d_forced, Ty, seed, d_audio, audio_capacity):
        scales = _f32(scales)
        self.lib.check(self.lib._fn("session_synthesize_device")(
            self._h, ctypes.c_void_p(d_ids), ctypes.c_void_p(d_lengths), B, Tx, _p(scales, c_f32p),
            ctypes.c_void_p(d_sid) if d_sid else None, ctypes.c_void_p(d_forced) if d_forced else None, Ty, seed,
            ctypes.c_void_p(d_audio), audio_capacity, None))

    def last_ms(self):
        ms = ctypes.c_float()
        self.lib.check(self.lib._fn("session_last_ms")(self._h, ctypes.byref(ms)))
        return ms.value

    def sync(self):
        self.lib.check(self.lib._fn("session_sync")(self._h))

    def set_options(self, use_graph=True, profile=False):
        self.lib.check(self.lib._fn("session_set_options")(self._h, int(use_graph), int(profile)))

    def graph_nodes(self):
        """kernel launches (graph nodes) of the captured forward, 0 before the first graph capture"""
        fn = self.lib._fn("session_graph_nodes")
        fn.argtypes = [ctypes.c_void_p]
        return int(fn(self._h))

    def set_sdp_always(self, on=True):
        """run the duration predictor even when durations are forced (fixed-work benchmark of the whole infer())"""
        self.lib.check(self.lib._fn("session_set_sdp_always")(self._h, int(on)))

    def profile_report(self):
        """-> {(op_name, kernel_instantiation): (launches, total_ms, flops)}"""
        buf = ctypes.create_string_buffer(1 << 16)
        self.lib.check(self.lib._fn("session_profile_report")(self._h, buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            name, kern, n, ms, fl = line.split()
            out[(name, kern)] = (int(n), float(ms), float(fl))
        return out
