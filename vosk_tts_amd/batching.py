"""Host-side batching for multi-utterance synthesis: the path shards by utterance
with no exchange step, so multi-GPU means independent replicas fed with balanced shards.

`plan_shards` sorts requests by token count and deals them greedily (longest first) to the shard
with the least predicted work, so padded batches stay tight and ranks finish together;
`pad_batch` builds the [B,T_x] feed of one shard; `scatter_results` restores request order.
No collective is needed on the data path: each rank reads the same request list and keeps its
own shard; only the small int16 PCM results travel back (over the launcher's channel of choice).
"""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def predicted_cost(n_tokens, frames_per_token=3.0):
    """Relative work of one utterance: decoder+flow scale with frames, encoder with tokens
    (SURVEY.md §8a totals: 162.4 MFLOP/frame vs 14.4 MFLOP/token)."""
    n_tokens = np.asarray(n_tokens, dtype=np.float64)
    return 162.4 * frames_per_token * n_tokens + 14.4 * n_tokens


def plan_shards(lengths, n_shards, max_batch=None):
    """lengths: tokens per request.  Returns n_shards lists of request indices (each sorted by
    descending length), balanced by predicted cost; max_batch caps items per shard per round."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    cost = predicted_cost(lengths)
    load = np.zeros(n_shards)
    shards = [[] for _ in range(n_shards)]
    for i in order:
        cand = [s for s in range(n_shards) if max_batch is None or len(shards[s]) < max_batch]
        if not cand:
            raise ValueError("max_batch * n_shards < number of requests")
        s = min(cand, key=lambda k: (load[k], k))
        shards[s].append(int(i))
        load[s] += cost[i]
    return shards


def pad_batch(token_lists, indices, pad_id=0):
    """-> ids int64 [B,T_x] (padded), lengths int64 [B] for the requests `indices`."""
    lens = np.array([len(token_lists[i]) for i in indices], dtype=np.int64)
    Tx = int(lens.max()) if len(indices) else 0
    ids = np.full((len(indices), Tx), pad_id, dtype=np.int64)
    for r, i in enumerate(indices):
        ids[r, :lens[r]] = token_lists[i]
    return ids, lens


def scatter_results(n_requests, shards, shard_outputs):
    """shard_outputs[s][r] is the result of request shards[s][r]; returns them in request order."""
    out = [None] * n_requests
    for idx, res in zip(shards, shard_outputs):
        for i, r in zip(idx, res):
            out[i] = r
    if any(o is None for o in out):
        raise ValueError("some requests were not assigned to any shard")
    return out


class MultiDeviceSynth:
    """Batched multi-utterance synthesis over N device replicas in ONE process (BASELINE north_star: "Batched multi-utterance
    synthesis shards across the 8 GPUs of one node as embarrassingly-parallel data replicas"; SURVEY.md 8e: "one host thread
    per device; results gathered on host in original order").  The seam is where the reference's gRPC server shares one Synth
    across a thread pool (server/tts_server.py:37-40,56-63): here each device holds its own `Model` replica (weights resident
    once per device) and a worker thread; a list of requests is front-ended on the host (the same g2p as `Synth`), sharded by
    predicted cost (`plan_shards`), every shard runs as padded batches of at most `max_batch` through the C ABI
    (vits_synthesize_pcm16, int16 conversion on the device) and the results come back in request order.

    Every request is synthesized as an independent utterance (VITS_FLAG_SOLO_BATCH) with its own noise seed, so the samples a
    request gets do not depend on how many devices there are or on what else was in its batch.  No collective anywhere: ctypes
    drops the GIL during the call, the devices run concurrently.

        mds = MultiDeviceSynth(model_path, devices=[0, 1, 2, 3, 4, 5, 6, 7])
        pcm = mds.synth_batch(["...", "..."], speaker_ids=2)        # list of int16 arrays, request order
    """

    def __init__(self, model_path=None, devices=None, model_name=None, lang=None, max_batch=32):
        from .model import Model
        from .synth import Synth

        if devices is None:
            from .capi import VitsLib

            n = VitsLib().device_count()
            devices = list(range(max(n, 1)))
        if not devices:
            raise ValueError("no devices")
        self.devices = list(devices)
        self.max_batch = int(max_batch)
        self.models = [Model(model_path=model_path, model_name=model_name, lang=lang, device=d) for d in self.devices]
        # the three voice families of vosk_tts/synth.py:64-103 behind the same door:
        #   "vits"        plain VITS (g2p_noembed -> token ids)                              -> vits_synthesize_pcm16, solo batch
        #   "vits_bert"   BERT-conditioned VITS (get_word_bert + g2p / g2p_noblank, synth.py:88-99) -> the same with a padded `bert` feed
        #   "multistream" StableTTS / Matcha voices (five id streams + BERT rows, synth.py:64-87)  -> stts_synthesize_batch per replica
        m0 = self.models[0]
        mt = str(m0.config.get("model_type") or "")
        if mt.startswith("multistream"):
            self.family = "multistream"
        elif getattr(m0.onnx, "hp", None) is not None and m0.onnx.hp.bert_dim > 0:
            if m0.tokenizer is None:
                raise NotImplementedError("this voice is BERT-conditioned (hparams.bert_dim > 0) but has no bert/ directory next to it")
            self.family = "vits_bert"
        else:
            self.family = "vits"
        self.synths = [Synth(m) for m in self.models]
        self._replica_locks = [threading.Lock() for _ in self.models]  # one batch at a time per replica (shared pool threads)
        self._pool = ThreadPoolExecutor(max_workers=len(self.devices), thread_name_prefix="vits-dev")
        self._seed = 0
        self._lock = threading.Lock()

    def close(self):
        self._pool.shutdown(wait=True)
        for m in self.models:
            m.onnx.close()
            if getattr(m, "bert_onnx", None) is not None and hasattr(m.bert_onnx, "close"):
                m.bert_onnx.close()

    def _run_shard(self, r, token_lists, idx, sids, scales, scale, seeds):
        """the requests `idx` on replica r, in batches of <= max_batch (already sorted by descending length) -> list of int16 arrays"""
        out = []
        sess = self.models[r].onnx
        for k in range(0, len(idx), self.max_batch):
            part = idx[k:k + self.max_batch]
            ids, lens = pad_batch(token_lists, part)
            feed = {"input": ids, "input_lengths": lens, "scales": scales, "sid": np.array([sids[i] for i in part], np.int64),
                    "bert": None, "phone_duration_extra": None, "vits.solo": True,
                    "vits.item_seeds": np.array([seeds[i] for i in part], np.uint64)}
            with self._replica_locks[r]:  # concurrent synth_batch() calls do not interleave on one replica
                pcm, lengths = sess.run_pcm16(feed, scale, return_lengths=True)  # lengths of THIS call (not the shared attribute)
            out.extend(pcm[j, :int(lengths[j])].copy() for j in range(len(part)))
        return out

    def synth_batch(self, texts, speaker_ids=0, noise_level=None, speech_rate=None, duration_noise_level=None, scale=None, seeds=None):
        """texts: list of str -> list of int16 PCM arrays (22.05 kHz), one per request, in request order.  `seeds`: optional
        per-request noise seeds (default: a running counter), `speaker_ids`: one id or one per request."""
        s0 = self.synths[0]
        if self.family == "vits":
            token_lists = [s0.g2p_noembed(s0.normalize(t)) for t in texts]
            return self.synth_tokens(token_lists, speaker_ids, noise_level, speech_rate, duration_noise_level, scale, seeds)
        # BERT-conditioned families: the front end needs the replica's BERT encoder, so it runs on the replica that gets the request;
        # requests are sharded by a cheap length estimate (phoneme count, no BERT needed)
        n = len(texts)
        if n == 0:
            return []
        texts = [s0.normalize(t) for t in texts]
        scales, scale, sids, seeds = self._call_params(n, speaker_ids, noise_level, speech_rate, duration_noise_level, scale, seeds)
        est = [len(s0.phonemize(t.replace("_", " "))) for t in texts]
        shards = plan_shards(est, len(self.devices))
        run = self._run_shard_bert if self.family == "vits_bert" else self._run_shard_multistream
        futs = [self._pool.submit(run, r, texts, idx, sids, scales, scale, seeds) if idx else None for r, idx in enumerate(shards)]
        return scatter_results(n, [idx for idx in shards if idx], [f.result() for f in futs if f is not None])

    def _call_params(self, n, speaker_ids, noise_level, speech_rate, duration_noise_level, scale, seeds):
        """runtime defaults (synth.py:50-56,106) and per-request speaker ids / seeds"""
        inf = self.synths[0].model.config.get("inference", {})
        noise_level = inf.get("noise_level", 0.8) if noise_level is None else noise_level
        speech_rate = inf.get("speech_rate", 1.0) if speech_rate is None else speech_rate
        duration_noise_level = inf.get("duration_noise_level", 0.8) if duration_noise_level is None else duration_noise_level
        scale = inf.get("scale", 1.0) if scale is None else scale
        scales = np.array([noise_level, 1.0 / speech_rate, duration_noise_level], np.float32)  # synth.py:106
        sids = [speaker_ids] * n if np.isscalar(speaker_ids) or speaker_ids is None else list(speaker_ids)
        sids = [0 if v is None else int(v) for v in sids]
        if seeds is None:
            with self._lock:
                seeds = [self._seed + 1 + i for i in range(n)]
                self._seed += n
        return scales, scale, sids, list(seeds)

    def _run_shard_bert(self, r, texts, idx, sids, scales, scale, seeds):
        """BERT-conditioned VITS requests `idx` on replica r: get_word_bert + g2p / g2p_noblank per request (synth.py:88-99), then padded
        solo batches with a padded `bert` feed [B, 768, T]"""
        synth, sess = self.synths[r], self.models[r].onnx
        fe = synth.g2p_noblank if synth.model.config.get("no_blank", 0) != 0 else synth.g2p
        fronts = []
        for i in idx:
            ids, emb = fe(texts[i], synth.get_word_bert(texts[i]))
            fronts.append((np.array(ids, np.int64), np.transpose(np.array(emb, np.float32))))  # [T], [768, T]
        order = sorted(range(len(idx)), key=lambda k: -len(fronts[k][0]))
        out = [None] * len(idx)
        for k0 in range(0, len(order), self.max_batch):
            part = order[k0:k0 + self.max_batch]
            lens = np.array([len(fronts[k][0]) for k in part], np.int64)
            T = int(lens.max())
            ids = np.zeros((len(part), T), np.int64)
            bert = np.zeros((len(part), sess.hp.bert_dim, T), np.float32)
            for b, k in enumerate(part):
                ids[b, :lens[b]] = fronts[k][0]
                bert[b, :, :lens[b]] = fronts[k][1]
            feed = {"input": ids, "input_lengths": lens, "scales": scales, "sid": np.array([sids[idx[k]] for k in part], np.int64),
                    "bert": bert, "phone_duration_extra": None, "vits.solo": True,
                    "vits.item_seeds": np.array([seeds[idx[k]] for k in part], np.uint64)}
            with self._replica_locks[r]:
                pcm, lengths = sess.run_pcm16(feed, scale, return_lengths=True)
            for b, k in enumerate(part):
                out[k] = pcm[b, :int(lengths[b])].copy()
        return out

    def _run_shard_multistream(self, r, texts, idx, sids, scales, scale, seeds):
        """multistream (StableTTS / Matcha) requests `idx` on replica r: the five-stream front end of Synth._feed per request
        (synth.py:64-87), then stts_synthesize_batch with per-request seeds; float -> int16 as Synth.audio_float_to_int16"""
        synth, sess = self.synths[r], self.models[r].onnx
        fronts = []
        for i in idx:
            feed, _ = synth._feed(texts[i], sids[i], None, None, None, None)
            pde = feed["phone_duration_extra"]
            fronts.append((feed["input"][0], feed["bert"][0], None if pde is None else np.asarray(pde, np.float32).reshape(-1)))
        order = sorted(range(len(idx)), key=lambda k: -fronts[k][0].shape[1])
        out = [None] * len(idx)
        for k0 in range(0, len(order), self.max_batch):
            part = order[k0:k0 + self.max_batch]
            lens = np.array([fronts[k][0].shape[1] for k in part], np.int64)
            T = int(lens.max())
            ids = np.zeros((len(part), 5, T), np.int64)
            bert = np.zeros((len(part), sess.hp.bert_dim, T), np.float32)
            any_pde = any(fronts[k][2] is not None for k in part)
            pde = np.zeros((len(part), T), np.float32) if any_pde else None
            for b, k in enumerate(part):
                ids[b, :, :lens[b]] = fronts[k][0]
                bert[b, :, :lens[b]] = fronts[k][1]
                if fronts[k][2] is not None:
                    pde[b, :lens[b]] = fronts[k][2]
            with self._replica_locks[r]:
                audio, ol = sess._model.synthesize_batch(ids, lens, scales, np.array([sids[idx[k]] for k in part], np.int64), bert, pde,
                                                         seed=0, item_seeds=np.array([seeds[idx[k]] for k in part], np.uint64))
            for b, k in enumerate(part):
                out[k] = synth.audio_float_to_int16(audio[b, :int(ol[b])] * scale)
        return out

    def synth_tokens(self, token_lists, speaker_ids=0, noise_level=None, speech_rate=None, duration_noise_level=None, scale=None, seeds=None):
        """synth_batch() behind the front end: token id lists in, int16 PCM arrays out (request order)."""
        n = len(token_lists)
        if n == 0:
            return []
        if self.family != "vits":
            raise NotImplementedError("synth_tokens takes plain VITS token ids; BERT-conditioned and multistream voices need the text (synth_batch)")
        scales, scale, sids, seeds = self._call_params(n, speaker_ids, noise_level, speech_rate, duration_noise_level, scale, seeds)
        shards = plan_shards([len(t) for t in token_lists], len(self.devices))
        futs = [self._pool.submit(self._run_shard, r, token_lists, idx, sids, scales, scale, seeds) if idx else None
                for r, idx in enumerate(shards)]
        return scatter_results(n, [idx for idx in shards if idx], [f.result() for f in futs if f is not None])
