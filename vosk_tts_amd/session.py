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
import logging
import os
import threading

import numpy as np

from .capi import VitsLib

log = logging.getLogger(__name__)


class _Pending:
    """one request waiting in the coalescer: its owner sleeps on `event` until a result (or an exception) is posted, or until it
    is handed a batch to lead"""
    __slots__ = ("key", "ids", "sid", "seed", "event", "result", "error", "lead")

    def __init__(self, key, ids, sid, seed):
        self.key, self.ids, self.sid, self.seed = key, ids, sid, seed
        self.event = threading.Event()
        self.result = self.error = self.lead = None


class RequestCoalescer:
    """Merges concurrent single-utterance requests into solo batches (server shape: one Synth shared by a thread pool,
    server/tts_server.py:35-57 -- N threads each calling .run() for ONE utterance).

    Natural batching: a request that finds fewer than `max_inflight` engine calls running runs at once (the first of them on the
    persistent single-utterance path; nothing added to a lone client's latency).  Requests that arrive while `max_inflight` calls are in
    flight queue up; when one of those calls returns, its thread hands the queue's compatible requests (same scales / output kind /
    conversion scale, at most `max_batch`) to the first waiter, which runs them as ONE padded VITS_FLAG_SOLO_BATCH call with per-request
    item seeds -- every item of a solo batch is its own single-utterance synthesis (own Philox streams, zeros beyond its own end), so
    what a request gets does not depend on what it was batched with -- and scatters the results.

    Gathering (`gather_us` > 0): a pool of N closed-loop clients otherwise settles into two alternating half-size batches (the clients of
    the batch that just returned are still on their way back when the next one is launched), and a batch of 8 costs 0.39 ms per request on
    the device where a batch of 16 costs 0.33.  So a leader that is about to launch FEWER requests than were recently present at the same
    time (`_peak`: the largest number of requests in flight + queued during the last 50 ms) waits up to `gather_us` for the stragglers.
    A lone client never waits (peak 1), a burst waits at most once per batch."""

    def __init__(self, run_batch, max_batch=32, max_inflight=1, gather_us=0):
        self._run_batch = run_batch  # (key, [(ids, sid, seed), ...]) -> list of per-request results
        self.max_batch = int(max_batch)
        self.max_inflight = max(1, int(max_inflight))  # engine calls that may run at the same time (the GPU overlaps a few small forwards)
        self.gather_us = float(gather_us)
        self._lock = threading.Lock()
        self._cond = threading.Condition(self._lock)  # a gathering leader sleeps here; arrivals notify
        self._busy = 0
        self._queue = []
        self._in_calls = 0     # requests inside engine calls right now
        self._gathering = False  # a leader is waiting for stragglers: arrivals join its queue instead of starting calls of their own
        self._in_key = {}      # ... per key
        self._peaks = {}       # key -> (most requests of that key present at once recently, when that was last seen)
        self.calls = 0        # engine calls issued
        self.requests = 0     # requests served
        self.largest = 0      # largest batch so far
        self.split_retries = 0  # merged batches that failed and were re-run member by member
        self.gathered = 0     # launches that waited for stragglers

    # (lock held) requests of `key` present at once (inside engine calls + queued): a leader only ever takes its own key, so the
    # stragglers it may wait for are counted per key
    def _note_presence(self, now, key):
        present = self._in_key.get(key, 0) + sum(1 for p in self._queue if p.key == key)
        peak, t = self._peaks.get(key, (1, 0.0))
        if present >= peak or now - t > 0.05:
            self._peaks[key] = (max(present, 1), now)
        if len(self._peaks) > 64:  # (keys are (kind, scales, scale) tuples: bounded in practice, bounded here for good)
            self._peaks = {key: self._peaks[key]}

    def _gather(self, me, batch):
        """(lock held) `batch` (containing `me`) is about to be launched: wait for stragglers while it is smaller than the recent peak.
        Returns the batch to launch."""
        if self.gather_us <= 0:
            return batch
        import time

        now = time.perf_counter()
        self._note_presence(now, me.key)
        target = min(self.max_batch, self._peaks[me.key][0] - self._in_key.get(me.key, 0))
        if len(batch) >= target:
            return batch
        deadline = now + self.gather_us * 1e-6
        self.gathered += 1
        self._gathering = True
        try:
            return self._gather_wait(me, batch, target, deadline)
        finally:
            self._gathering = False

    def _gather_wait(self, me, batch, target, deadline):
        import time

        while True:
            more = [p for p in self._queue if p.key == me.key][:max(target, 1) - len(batch)]
            if more:
                taken = set(map(id, more))
                self._queue = [p for p in self._queue if id(p) not in taken]
                batch = batch + more
            now = time.perf_counter()
            if len(batch) >= target or now >= deadline:
                return batch
            self._cond.wait(deadline - now)

    def _promote(self):
        """(lock held) a leader has just taken its batch: requests it left in the queue (beyond its target, or of another key) must not
        wait for a call to finish when a call slot is free -- the queue's head becomes a leader of its own (and gathers in turn)."""
        if not self._queue or self._busy >= self.max_inflight or self._gathering:
            return None
        head = self._queue.pop(0)
        self._busy += 1
        head.lead = [head]
        return head

    def submit(self, key, ids, sid, seed):
        import time

        me = _Pending(key, ids, sid, seed)
        promote = None
        with self._lock:
            if self._busy >= self.max_inflight or self._gathering:
                self._queue.append(me)
                if self.gather_us > 0:
                    self._note_presence(time.perf_counter(), key)
                self._cond.notify_all()
                batch = None
            else:
                self._busy += 1
                batch = self._gather(me, [me])
                self._in_calls += len(batch)
                if self.gather_us > 0:
                    self._in_key[key] = self._in_key.get(key, 0) + len(batch)
                promote = self._promote()
        if batch is not None and promote is not None:
            promote.event.set()
        if batch is None:
            me.event.wait()
            if me.lead is None:  # somebody else ran it
                if me.error is not None:
                    raise me.error
                return me.result
            with self._lock:
                batch = self._gather(me, me.lead)
                self._in_calls += len(batch)
                if self.gather_us > 0:
                    self._in_key[key] = self._in_key.get(key, 0) + len(batch)
                promote = self._promote()
            if promote is not None:
                promote.event.set()
        # this thread leads `batch` (which contains its own request)
        try:
            try:
                outs = self._run_batch(batch[0].key, [(p.ids, p.sid, p.seed) for p in batch])
                for p, o in zip(batch, outs):
                    p.result = o
            except Exception as e:
                if len(batch) == 1:
                    me.error = e
                else:
                    # a merged batch failed: one bad request (token / speaker id out of range, ...) must not fail the unrelated
                    # requests it was merged with -- run the members one by one, each gets its own result or its own exception
                    self.split_retries += 1
                    try:
                        for p in batch:
                            try:
                                p.result = self._run_batch(p.key, [(p.ids, p.sid, p.seed)])[0]
                            except Exception as e1:
                                p.error = e1
                    except BaseException as e2:  # interrupted in the middle of the retries (the sibling handler below does not cover this one)
                        for p in batch:
                            if p.result is None and p.error is None:
                                p.error = RuntimeError(f"batch leader interrupted: {e2!r}") if p is not me else e2
            except BaseException as e:  # KeyboardInterrupt / SystemExit in the leader: nobody may be left waiting
                for p in batch:
                    p.error = RuntimeError(f"batch leader interrupted: {e!r}") if p is not me else e
        finally:
            with self._lock:
                self.calls += 1
                self.requests += len(batch)
                self.largest = max(self.largest, len(batch))
                self._in_calls -= len(batch)
                if self.gather_us > 0:  # (the per-key presence count only feeds the gather window)
                    left = self._in_key.get(batch[0].key, 0) - len(batch)
                    if left > 0:
                        self._in_key[batch[0].key] = left
                    else:
                        self._in_key.pop(batch[0].key, None)  # a server that forwards client-chosen speech rates sees unboundedly many keys
                for p in batch:
                    if p.result is None and p.error is None:  # whatever happened above: a member never wakes up with nothing
                        p.error = RuntimeError("coalesced batch ended without a result for this request")
                nxt = None
                if self._queue:
                    head = self._queue[0]
                    take = [p for p in self._queue if p.key == head.key][:self.max_batch]
                    taken = set(map(id, take))
                    self._queue = [p for p in self._queue if id(p) not in taken]
                    head.lead = take
                    nxt = head
                else:
                    self._busy -= 1
            for p in batch:
                if p is not me:
                    p.event.set()
            if nxt is not None:
                nxt.event.set()
        if me.error is not None:
            raise me.error
        return me.result

# frame-bucket ("back") contexts the engine keeps per T_x bucket before it evicts the least recently used one (csrc/engine_fastpath.hip.h back_get)
BACK_SESSIONS_PER_FRONT = 6

_GRAPH_INPUTS = ("input", "input_lengths", "scales", "sid")
_OPTIONAL_NONE = ("bert", "phone_duration_extra")
# extension feeds (not part of the ONNX graph) used by parity tests
_EXT = ("vits.noise_dp", "vits.noise_prior", "vits.forced_durations", "vits.seed", "vits.solo", "vits.item_seeds")


class _Arg:
    def __init__(self, name, typ, shape):
        self.name, self.type, self.shape = name, typ, shape


class VitsSession:
    def __init__(self, blob, device=0, lib=None, coalesce=True, max_batch=32, max_inflight=None):
        self._lib = lib or VitsLib()
        self._model = self._lib.create(blob, device)
        self.hp = self._model.hp
        self._seed = itertools.count(1)
        self._seed_lock = threading.Lock()
        # concurrent single-utterance run() / run_pcm16() calls are merged into solo batches (RequestCoalescer); coalesce=False:
        # every call goes to the engine on its own, as in rounds 1-3
        if max_inflight is None:
            # measured on MI355X (16 threads, 47-token requests, bench.py host_api.concurrent): 1 call in flight 1520 requests/s, 2-4: ~2000,
            # 8: 2200, no coalescing: 2110 -- the device overlaps a handful of single-utterance forwards (one of them on the persistent
            # programs); merging only starts beyond that, which also bounds the workspaces and streams a burst can pin
            max_inflight = int(os.environ.get("VITS_COALESCE_INFLIGHT", "8"))
        gather_us = float(os.environ.get("VITS_COALESCE_GATHER_US", "0"))
        self.coalescer = RequestCoalescer(self._run_solo_batch, max_batch, max_inflight, gather_us) if coalesce else None
        self._ps_timeouts = 0  # persistent-program timeouts already reported (persist_state)

    def persist_state(self, with_device=True):
        """State of the single-utterance persistent programs in this process (vits_persist_state): a poll timeout -- the device shared
        with another process, a transient -- puts single utterances on the ~1.5x slower launch path for a bounded interval, after which
        the programs are re-armed.  Logged at WARNING whenever the timeout count has grown since the last look."""
        st = self._model.persist_state(with_device)
        if st["timeouts"] > self._ps_timeouts:
            # process_owns_device is the outcome of the LAST call's lock lease (1 got it, -1 was refused, 0 not asked yet) and is only
            # known with a model handle: the periodic watcher (with_device=False) says nothing about it
            lock = (", last call " + {1: "got", -1: "was refused", 0: "has not asked for"}.get(st["process_owns_device"], "?") +
                    " the device's program lock") if with_device else ""
            log.warning("persistent programs timed out %d time(s) so far (re-armed %d time(s)); launch path for another %d ms%s",
                        st["timeouts"], st["rearms"], st["off_for_ms"], lock)
            self._ps_timeouts = st["timeouts"]
        return st

    def _watch(self):
        # cheap (process-wide counters, no device access): read through the C ABI every 256 requests
        self._n_req = getattr(self, "_n_req", 0) + 1
        if self._n_req & 255 == 0 and self._lib.is_device:
            try:
                self.persist_state(with_device=False)
            except Exception:  # diagnostics must never cost a request
                pass

    def _coalescable(self, feed, ids):
        """plain single-utterance requests only: no injected tensors, no pinned durations, no caller-chosen batch semantics.
        Returns the request's token count (> 0) when it may be merged, 0 otherwise -- a length outside (0, T] is NOT sliced here:
        such a request takes the direct call, whose C-side check answers VITS_ERR_ARG as it always did."""
        if not (self.coalescer is not None and ids.shape[0] == 1 and self.hp.bert_dim == 0 and
                not any(k in feed for k in ("vits.noise_dp", "vits.noise_prior", "vits.forced_durations", "vits.solo", "vits.item_seeds", "bert"))):
            return 0
        lens = np.asarray(feed["input_lengths"]).reshape(-1)
        if lens.shape[0] != 1:
            return 0
        n = int(lens[0])
        return n if 0 < n <= ids.shape[1] else 0

    def _run_solo_batch(self, key, reqs):
        """reqs: [(ids [1,T] int64, sid, seed)] -> per request (audio-or-pcm [1,S_b], lengths [1]); one request: the plain call"""
        kind, scales, scale = key
        scales = np.array(scales, np.float32)
        if len(reqs) == 1:
            ids, sid, seed = reqs[0]
            lens = np.array([ids.shape[1]], np.int64)
            if kind == "pcm":
                return [self._model.synthesize_pcm16(ids, lens, scales, np.array([sid], np.int64), pcm_scale=scale, seed=seed)]
            return [self._model.synthesize(ids, lens, scales, np.array([sid], np.int64), seed=seed)]
        # batch sizes come in powers of two (filled up with one-token dummies, whose padding tiles the ragged solo batch never
        # computes): every (batch size, length bucket) is a workspace and two captured graphs inside the engine, and a thread pool
        # produces every batch size between 1 and its own size
        n_real = len(reqs)
        n_pad = 1
        while n_pad < n_real:
            n_pad *= 2
        reqs = list(reqs) + [(np.ones((1, 1), np.int64), 0, 0)] * (n_pad - n_real)
        lens = np.array([r[0].shape[1] for r in reqs], np.int64)
        batch = np.zeros((len(reqs), int(lens.max())), np.int64)
        for b, r in enumerate(reqs):
            batch[b, :lens[b]] = r[0][0]
        sids = np.array([r[1] for r in reqs], np.int64)
        seeds = np.array([r[2] for r in reqs], np.uint64)
        if kind == "pcm":
            out, ol = self._model.synthesize_pcm16(batch, lens, scales, sids, pcm_scale=scale, seed=int(seeds[0]), solo=True, item_seeds=seeds)
        else:
            out, ol = self._model.synthesize(batch, lens, scales, sids, seed=int(seeds[0]), solo=True, item_seeds=seeds)
        return [(out[b:b + 1, :int(ol[b])].copy(), ol[b:b + 1].copy()) for b in range(n_real)]

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
        self._watch()
        n = self._coalescable(feed, ids)
        if n:
            key = ("f32", tuple(float(v) for v in np.asarray(feed["scales"], np.float32).reshape(-1)), 1.0)
            audio, lengths = self.coalescer.submit(key, np.ascontiguousarray(ids[:, :n], np.int64), int(sid[0]), int(seed))
            self.last_lengths = lengths
            return [audio[:, None, None, :]]
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
        self._watch()
        n = self._coalescable(feed, ids)
        if n:
            key = ("pcm", tuple(float(v) for v in np.asarray(feed["scales"], np.float32).reshape(-1)), float(scale))
            pcm, lengths = self.coalescer.submit(key, np.ascontiguousarray(ids[:, :n], np.int64), int(sid[0]), int(seed))
            self.last_lengths = lengths
            return (pcm, lengths) if return_lengths else pcm
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
        n = int(np.asarray(feed["input_lengths"]).reshape(-1)[0])
        fd = feed.get("vits.forced_durations")
        nd = feed.get("vits.noise_dp")
        bert = feed.get("bert")  # BERT-conditioned flavour (synth.py:88-99): vits_stream_open takes it through opts->bert
        return self._model.stream(
            ids[:, :n], np.asarray(feed["scales"], np.float32).reshape(-1), int(sid[0]), chunk_frames=chunk_frames,
            noise_dp=None if nd is None else np.asarray(nd)[:, :, :n], noise_prior=feed.get("vits.noise_prior"),
            forced_durations=None if fd is None else np.asarray(fd)[:, :n], seed=seed,
            bert=None if bert is None else np.ascontiguousarray(np.asarray(bert, np.float32)[:, :, :n]))

    def warmup(self, max_tokens=128, frames_per_token=(2.0, 5.0), speaker_id=0, freeze_gc=False, typical_frames_per_token=3.0,
               stream_chunk_frames=None):
        """Pay the one-off costs of the graph-replayed host path before the first real request does (extension; onnxruntime has the same
        need and no such call): for every T_x bucket (multiples of 8) up to `max_tokens`, the frame buckets (multiples of 32) between
        frames_per_token[0] and [1] frames per token that lie closest to `typical_frames_per_token` -- at most BACK_SESSIONS_PER_FRONT - 1
        of them, the engine's per-T_x cap less one -- are synthesized once with pinned durations -- which lays out the workspaces, builds
        the persistent programs and captures the front / back graphs of those (T_x, T_y) buckets -- plus one free-running call.
        Returns (calls, seconds).  Requests outside the warmed buckets still work; they pay their bucket's capture (tens of
        milliseconds) on first use.
        stream_chunk_frames=<n> also opens and drains a stream (run_stream's engine path) of every T_x bucket TWICE, at the typical frame
        count: streams run the eager stage path on a pooled session, and the first two opens of a size cost 15 - 40 ms instead of the
        steady 5 (workspace allocation, then one more slow open after the first graph-replayed call: tools/stream_warm_probe.py).
        freeze_gc=True additionally runs gc.collect() + gc.freeze() at the end: CPython's older-generation passes over the heap of a
        process that has loaded its models stop every thread for tens of milliseconds (25-60 ms measured beside 1 ms requests,
        profiles/r5_m2_gc.txt); frozen, that heap is no longer traversed and the collector only looks at what requests allocate.
        Process-wide, hence opt-in."""
        import time

        t0 = time.perf_counter()
        calls = 0
        scales = np.array([0.8, 1.0, 0.8], np.float32)
        sid = np.array([speaker_id], np.int64)
        lo, hi = float(frames_per_token[0]), float(frames_per_token[1])
        for tx in range(8, max(int(max_tokens), 8) + 1, 8):
            ids = np.ones((1, tx), np.int64)
            lens = np.array([tx], np.int64)
            bert = np.zeros((1, self.hp.bert_dim, tx), np.float32) if self.hp.bert_dim > 0 else None
            first = max(32, (int(lo * tx) + 31) // 32 * 32)
            last = max(first, (int(hi * tx) + 31) // 32 * 32)
            # The engine keeps at most BACK_SESSIONS_PER_FRONT frame-bucket contexts per T_x bucket and evicts the least recently used
            # (engine_fastpath.hip.h back_get): warming more than that frees the first ones again (from T_x = 64 on there are 7+
            # buckets between 2 and 5 frames per token).  So: the buckets closest to `typical_frames_per_token`, at most one fewer
            # than the cap (the free-running call below may open one more), walked from the farthest to the closest -- the
            # likeliest bucket is the most recently used one when warm-up ends.
            centre = typical_frames_per_token * tx
            buckets = sorted(range(first, last + 1, 32), key=lambda ty: abs(ty - centre))[:BACK_SESSIONS_PER_FRONT - 1]
            for ty in sorted(buckets, key=lambda ty: -abs(ty - centre)) + [None]:
                dur = None
                if ty is not None:  # exactly ty frames over the tx tokens
                    dur = np.full((1, tx), ty // tx, np.int32)
                    dur[0, :ty % tx] += 1
                self._model.synthesize_pcm16(ids, lens, scales, sid, forced_durations=dur, seed=1, bert=bert)
                calls += 1
            if stream_chunk_frames:
                ty = max(1, int(round(typical_frames_per_token * tx)))
                dur = np.full((1, tx), ty // tx, np.int32)
                dur[0, :ty % tx] += 1
                for _ in range(2):
                    for _chunk in self._model.stream(ids, scales, speaker_id, chunk_frames=int(stream_chunk_frames), forced_durations=dur, seed=1, bert=bert):
                        pass
                    calls += 1
        if freeze_gc:
            import gc

            gc.collect()
            gc.freeze()
        return calls, time.perf_counter() - t0

    def close(self):
        self._model.close()
