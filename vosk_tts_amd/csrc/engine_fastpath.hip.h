// engine_fastpath.hip.h -- the graph-replayed host entry points (vits_synthesize / vits_synthesize_pcm16 fast path): front / back contexts, capture, replay.
// Part of the ONE translation unit engine.hip (included there, in order; not a standalone header): split out in round 6 so that the
// planner / launch selection / stages / host paths can be read on their own.
#pragma once
// ---- fast path of the host entry point --------------------------------------------------------------------------
// What Synth.synth_audio brackets (vosk_tts/synth.py:122-131) is ids on the host -> waveform on the host, one request at a
// time, each with its own scales and a fresh noise draw.  Replaying that as captured hipGraphs needs three things:
//   * per-call scalars (scales, seed, pcm scale) live in a device block the kernels read (SynthDev), inputs are copied
//     through ONE pinned staging buffer by a memcpy node of the graph -> a graph depends on shapes only;
//   * shapes are bucketed: T_x up to a multiple of 8, T_y up to a multiple of 32.  Every stage up to the flow masks per
//     item (exactly the ragged-batch machinery), and the decoder of a bucketed single utterance reads zeros beyond the
//     item's own end at every stage (rag halo 0) -- the arithmetic of the exact-size run on every valid sample;
//   * T_y is only known after the duration predictor: phase 1 (text encoder .. durations) is one graph of a FRONT
//     session keyed by (B, T_x bucket); phase 2 (prior sample, flow, decoder, optional int16 conversion, D2H) one graph of
//     a BACK session per frame bucket, which reads the front's stats / cum / cond vectors / lengths in place.
// Per call: fill the pinned block, launch graph 1, wait (the one host round trip the path needs), launch graph 2, wait,
// copy out.  No hipMalloc / hipFree / re-plan in steady state.  Calls that inject noise tensors (parity tests) take the
// eager path below (vits_synthesize_eager), which is also the A/B reference of the fast path in tests.
static thread_local int g_fast_path = 1;
// cap on the device memory idle fast-path sessions may pin per model: VITS_CACHE_MB, else a quarter of what was free on the device
// when the first call asked (at most 24 GiB).  Besides the cap, an allocation failure on the request path evicts every idle
// session and retries once (fronts_evict_all).
static size_t fast_cache_cap() {
  static const size_t cap = [] {
    if (getenv("VITS_CACHE_MB")) return (size_t)atol(getenv("VITS_CACHE_MB")) << 20;
    size_t fr = 0, tot = 0;
    size_t c = (size_t)24 << 30;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr / 4 < c) c = fr / 4;
    return c;
  }();
  return cap;
}

static size_t session_device_bytes(const vits_session* s) {
  size_t n = s->arena_bytes + s->io_bytes + s->out_elems * (sizeof(float) + sizeof(int16_t));
  for (auto& kv : s->backs) n += session_device_bytes(kv.second);
  return n;
}

static int front_acquire(vits_model* m, int B, int TxB, vits_session** out) {
  {
    std::lock_guard<std::mutex> g(m->pool_mu);
    auto it = m->fronts.find(std::make_pair(B, TxB));
    if (it != m->fronts.end()) {
      *out = it->second;
      m->fronts_bytes -= it->second->cache_bytes;
      m->fronts.erase(it);
      return VITS_OK;
    }
  }
  vits_session* s = nullptr;
  TRY(session_new(m, &s));
  s->ps_roles = PERSIST_ENC | PERSIST_SDP;
  // per-call input block: [SynthDev | lengths int64 [B] | sid int64 [B] | ids int64 [B,TxB] | forced int32 [B,TxB]]
  // (allocated BEFORE the workspace is laid out: the text-encoder program of a BERT-conditioned voice is resolved against io_d + io_bert)
  s->io_len = align_up(sizeof(SynthDev), 64);
  s->io_sid = s->io_len + align_up(sizeof(int64_t) * B, 64);
  s->io_ids = s->io_sid + align_up(sizeof(int64_t) * B, 64);
  s->io_forced = s->io_ids + align_up(sizeof(int64_t) * (size_t)B * TxB, 64);
  s->io_seeds = s->io_forced + align_up(sizeof(int32_t) * (size_t)B * TxB, 64);
  s->io_bytes = s->io_seeds + align_up(sizeof(unsigned long long) * B, 64);
  if (m->hp.bert_dim > 0) {  // the `bert` feed of a BERT-conditioned voice (vosk_tts/synth.py:88-99) rides in the same block: [B, bert_dim, TxB]
    s->io_bert = s->io_bytes;
    s->io_bytes += align_up(sizeof(float) * (size_t)B * m->hp.bert_dim * TxB, 64) + 256;  // (+ slack: the program's operand window reads whole 16-column tiles)
  }
  if (hipHostMalloc((void**)&s->io_h, s->io_bytes) != hipSuccess || hipMalloc((void**)&s->io_d, s->io_bytes) != hipSuccess ||
      hipHostMalloc((void**)&s->h_ylen, sizeof(int64_t) * (B + 1)) != hipSuccess) {
    session_free(s);
    return fail(VITS_ERR_NOMEM, "fast-path staging buffers");
  }
  memset(s->io_h, 0, s->io_bytes);
  if (s->io_bert && B == 1) s->ps_bert = reinterpret_cast<const float*>(s->io_d + s->io_bert);
  const int rc = session_reserve(s, B, TxB, 1);
  if (rc != VITS_OK) { session_free(s); return rc; }
  *out = s;
  return VITS_OK;
}

static void front_release(vits_model* m, vits_session* s) {
  std::vector<vits_session*> evict;
  {
    std::lock_guard<std::mutex> g(m->pool_mu);
    s->last_use = ++m->use_clock;
    s->cache_bytes = session_device_bytes(s);
    m->fronts.emplace(std::make_pair(s->B, s->Tx), s);
    m->fronts_bytes += s->cache_bytes;
    while ((m->fronts_bytes > fast_cache_cap() && m->fronts.size() > 1) || m->fronts.size() > 48) {
      auto lru = m->fronts.begin();
      for (auto it = m->fronts.begin(); it != m->fronts.end(); ++it)
        if (it->second->last_use < lru->second->last_use) lru = it;
      m->fronts_bytes -= lru->second->cache_bytes;
      evict.push_back(lru->second);
      m->fronts.erase(lru);
    }
  }
  for (vits_session* e : evict) session_free(e);
}

// frees every idle front (and its backs): the answer to a failed allocation on the request path
static void fronts_evict_all(vits_model* m) {
  std::vector<vits_session*> evict;
  {
    std::lock_guard<std::mutex> g(m->pool_mu);
    for (auto& kv : m->fronts) evict.push_back(kv.second);
    m->fronts.clear();
    m->fronts_bytes = 0;
  }
  for (vits_session* e : evict) session_free(e);
  (void)hipGetLastError();
}

// back session of `F` for frame bucket TyB (created on first use; at most 6 buckets stay cached per front)
static int back_get(vits_session* F, int TyB, vits_session** out) {
  auto it = F->backs.find(TyB);
  if (it != F->backs.end()) { it->second->last_use = ++F->last_use; *out = it->second; return VITS_OK; }
  if (F->backs.size() >= 6) {
    auto lru = F->backs.begin();
    for (auto jt = F->backs.begin(); jt != F->backs.end(); ++jt)
      if (jt->second->last_use < lru->second->last_use) lru = jt;
    hipStreamSynchronize(F->stream);
    session_free(lru->second);
    F->backs.erase(lru);
  }
  vits_model* m = F->m;
  vits_session* s = new vits_session();
  s->m = m;
  s->stream = F->stream;
  s->own_stream = false;
  s->front = F;
  int rc = VITS_OK;
  if (hipMalloc((void**)&s->d_err, sizeof(int)) != hipSuccess || hipMemsetAsync(s->d_err, 0, sizeof(int), s->stream) != hipSuccess)
    rc = fail(VITS_ERR_NOMEM, "back session");
  s->ps_roles = PERSIST_FLOW;
  s->ps_defer = true;  // planned below, once the shared tensors point into the front
  if (rc == VITS_OK) rc = session_reserve(s, F->B, F->Tx, TyB);
  s->out_elems = (size_t)F->B * TyB * m->hp.hop_length;
  if (rc == VITS_OK && (hipMalloc((void**)&s->out_d, s->out_elems * sizeof(float)) != hipSuccess ||
                        hipMalloc((void**)&s->pcm_d, s->out_elems * sizeof(int16_t)) != hipSuccess ||
                        hipHostMalloc((void**)&s->out_h, s->out_elems * sizeof(float)) != hipSuccess ||
                        hipHostMalloc((void**)&s->h_err, 64) != hipSuccess))
    rc = fail(VITS_ERR_NOMEM, "fast-path output buffers (%zu samples)", s->out_elems);
  if (s->h_err) *s->h_err = 0;
  if (rc != VITS_OK) { s->stream = nullptr; session_free(s); return rc; }
  // phase 2 reads the front's phase-1 results in place
  s->stats = F->stats; s->cum = F->cum; s->condv = F->condv; s->len_y = F->len_y; s->len_x = F->len_x; s->ylen64 = F->ylen64;
  s->dv = reinterpret_cast<const SynthDev*>(F->io_d);
  s->item_seeds = reinterpret_cast<const unsigned long long*>(F->io_d + F->io_seeds);
  // the persistent flow program was resolved against this session's own len_y / condv: resolve it again against the front's
  persist_plan(s);
  s->last_use = ++F->last_use;
  F->backs[TyB] = s;
  *out = s;
  return VITS_OK;
}

// A capture that does not reach capture_end (an early return between Begin and End) must not leave the stream in capture mode:
// every later call on the session would fail.  The guard ends and discards it.
struct CaptureGuard {
  hipStream_t st; bool done = false;
  explicit CaptureGuard(hipStream_t s) : st(s) {}
  ~CaptureGuard() {
    if (done) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      hipGraph_t g = nullptr;
      hipStreamEndCapture(st, &g);
      if (g) hipGraphDestroy(g);
    }
    (void)hipGetLastError();
  }
};
static int capture_end(vits_session* s, hipGraphExec_t* out, CaptureGuard* guard = nullptr) {
  hipGraph_t g = nullptr;
  if (guard) guard->done = true;
  HIP_TRY(hipStreamEndCapture(s->stream, &g));
  hipError_t e = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  hipGraphDestroy(g);
  if (e != hipSuccess) return fail(VITS_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  return VITS_OK;
}

static int phase1_launch(vits_session* F, bool forced, bool solo) {
  const int gi = (persist_mask() ? 4 : 0) + (forced ? 2 : 0) + (solo ? 1 : 0);
  if (!F->g1[gi]) {
    const int B = F->B, TxB = F->Tx;
    HIP_TRY(hipStreamBeginCapture(F->stream, hipStreamCaptureModeThreadLocal));
    CaptureGuard cg(F->stream);
    hipMemcpyAsync(F->io_d, F->io_h, F->io_bytes, hipMemcpyHostToDevice, F->stream);
    F->ragged = true; F->solo = solo; F->tile_keys.clear();
    F->dv = reinterpret_cast<const SynthDev*>(F->io_d);
    F->item_seeds = reinterpret_cast<const unsigned long long*>(F->io_d + F->io_seeds);
    const int64_t* d_len = reinterpret_cast<const int64_t*>(F->io_d + F->io_len);
    const int64_t* d_sid = reinterpret_cast<const int64_t*>(F->io_d + F->io_sid);
    const int64_t* d_ids = reinterpret_cast<const int64_t*>(F->io_d + F->io_ids);
    const int32_t* d_forced = reinterpret_cast<const int32_t*>(F->io_d + F->io_forced);
    run_cond(F, d_sid, B, d_len, F->len_x, TxB);
    if (persist_mask() == (PERSIST_SDP | PERSIST_ENC | PERSIST_FLOW) && B == 1 && F->ps_front[forced ? 0 : 1].ok && (!F->io_bert || F->ps_bert)) {
      // text encoder [+ duration predictor] + durations as one persistent launch (a BERT-conditioned voice: the program reads the
      // `bert` tensor straight from the input block, two more steps)
      persist_launch(F, F->ps_front[forced ? 0 : 1], "front.persist", nullptr, 0.f, 0, d_ids, forced ? d_forced : nullptr, 1.f, 0.f);
      F->ea_pending = false;
    } else {
      run_text_encoder(F, d_ids, B, TxB, F->io_bert ? reinterpret_cast<const float*>(F->io_d + F->io_bert) : nullptr);
      if (!forced) run_duration(F, F->x, nullptr, 0.f, 0, B, TxB, true);
      run_durations(F, forced ? d_forced : nullptr, 1.f, B, TxB, 0);
    }
    hipMemcpyAsync(F->h_ylen, F->ylen64, sizeof(int64_t) * B, hipMemcpyDeviceToHost, F->stream);
    hipMemcpyAsync(F->h_ylen + B, F->d_err, sizeof(int), hipMemcpyDeviceToHost, F->stream);
    F->ragged = false; F->solo = false;
    TRY(capture_end(F, &F->g1[gi], &cg));
  }
  HIP_TRY(hipGraphLaunch(F->g1[gi], F->stream));
  return VITS_OK;
}

static int phase2_launch(vits_session* F, vits_session* Bk, bool solo, bool pcm) {
  const int gi = (persist_mask() ? 4 : 0) + (solo ? 2 : 0) + (pcm ? 1 : 0);
  if (!Bk->g2[gi]) {
    const int B = F->B, TxB = F->Tx, TyB = Bk->Ty;
    const long long stride = (long long)TyB * F->m->hp.hop_length;
    HIP_TRY(hipStreamBeginCapture(F->stream, hipStreamCaptureModeThreadLocal));
    CaptureGuard cg(F->stream);
    Bk->ragged = true; Bk->solo = solo; Bk->rag_b1 = true; Bk->tile_keys.clear();
    float* z;
    if (persist_mask() == (PERSIST_SDP | PERSIST_ENC | PERSIST_FLOW) && B == 1 && Bk->ps_back.ok) {  // prior sample + flow as one persistent launch
      persist_launch(Bk, Bk->ps_back, "back.persist");
      z = Bk->zB;
    } else {
      run_expand(Bk, nullptr, TyB, 0.f, 0, Bk->zA, B, TxB, TyB);
      z = run_flow(Bk, B, TyB);
    }
    // a lone utterance decodes as the exact-size run does (zeros beyond its end); batches keep the reference's padded-batch
    // continuation over the halo unless the caller asked for independent items
    run_decoder(Bk, z, true, B, TyB, Bk->out_d, stride, nullptr, true, (solo || B == 1) ? 0 : F->m->rag_halo);
    if (pcm) {
      hipLaunchKernelGGL(pcm16_kernel, dim3(cdiv((int)stride, 256), B), dim3(256), 0, F->stream, Bk->out_d, stride, Bk->pcm_d, stride, stride, 1.f, Bk->dv);
      hipMemcpyAsync(Bk->out_h, Bk->pcm_d, Bk->out_elems * sizeof(int16_t), hipMemcpyDeviceToHost, F->stream);
    } else {
      hipMemcpyAsync(Bk->out_h, Bk->out_d, Bk->out_elems * sizeof(float), hipMemcpyDeviceToHost, F->stream);
    }
    hipMemcpyAsync(Bk->h_err, Bk->d_err, sizeof(int), hipMemcpyDeviceToHost, F->stream);  // (the flow program's error bits)
    Bk->ragged = false; Bk->solo = false;
    TRY(capture_end(F, &Bk->g2[gi], &cg));
  }
  HIP_TRY(hipGraphLaunch(Bk->g2[gi], F->stream));
  return VITS_OK;
}

static int device_error_word(int e) {
  if (e & PS_ERR_TIMEOUT) return persist_timed_out();  // first: a timeout invalidates every bit derived from computed data (check_err)
  if (e & 1) return fail(VITS_ERR_ARG, "token id out of range");
  if (e & 2) return fail(VITS_ERR_ARG, "speaker id out of range");
  if (e & 4) return fail(VITS_ERR_ARG, "T_y exceeds frame capacity");
  return e ? fail(VITS_ERR_DEVICE, "device error word %d", e) : VITS_OK;
}

static int synth_fast(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                      const int64_t* sid, const vits_synth_opts* opts, bool pcm, float pcm_scale, void** out, int64_t* out_samples,
                      int64_t* out_lengths) {
  const vits_hparams& hp = m->hp;
  HIP_TRY(hipSetDevice(m->device));
  const int TxB = (Tx + 7) / 8 * 8;
  const bool forced = opts && opts->forced_durations, solo = opts && (opts->flags & VITS_FLAG_SOLO_BATCH);
  // (declared before the session guard: the call's last stream synchronisation happens before this scope ends)
  InFlight inflight(m->device);
  PersistScope pscope(B == 1 ? m->device : -1, inflight.before);  // a single utterance takes the persistent stages when the device is quiet enough as it starts
  vits_session* F = nullptr;
  {
    int rc = front_acquire(m, B, TxB, &F);
    if (rc == VITS_ERR_NOMEM) { fronts_evict_all(m); rc = front_acquire(m, B, TxB, &F); }
    if (rc != VITS_OK) return rc;
  }
  struct Rel { vits_model* m; vits_session* s; ~Rel() { front_release(m, s); } } rel{m, F};
  // ---- inputs -> pinned block
  SynthDev* hv = reinterpret_cast<SynthDev*>(F->io_h);
  hv->scales[0] = scales[0]; hv->scales[1] = scales[1]; hv->scales[2] = scales[2];
  hv->pcm_scale = pcm_scale;
  hv->seed = opts ? opts->seed : 0;
  int64_t* h_len = reinterpret_cast<int64_t*>(F->io_h + F->io_len);
  int64_t* h_sid = reinterpret_cast<int64_t*>(F->io_h + F->io_sid);
  int64_t* h_ids = reinterpret_cast<int64_t*>(F->io_h + F->io_ids);
  int32_t* h_forced = reinterpret_cast<int32_t*>(F->io_h + F->io_forced);
  unsigned long long* h_seeds = reinterpret_cast<unsigned long long*>(F->io_h + F->io_seeds);
  for (int b = 0; b < B; ++b) {
    h_seeds[b] = (opts && opts->item_seeds) ? opts->item_seeds[b] : hv->seed + (uint64_t)b;
    h_len[b] = lengths[b];
    h_sid[b] = sid ? sid[b] : 0;
    memcpy(h_ids + (size_t)b * TxB, ids + (size_t)b * Tx, sizeof(int64_t) * Tx);
    for (int t = Tx; t < TxB; ++t) h_ids[(size_t)b * TxB + t] = 0;
    if (forced) {
      memcpy(h_forced + (size_t)b * TxB, opts->forced_durations + (size_t)b * Tx, sizeof(int32_t) * Tx);
      for (int t = Tx; t < TxB; ++t) h_forced[(size_t)b * TxB + t] = 0;
    }
  }
  if (F->io_bert) {  // [B, bert_dim, Tx] -> [B, bert_dim, TxB], bucket columns zero
    float* h_bert = reinterpret_cast<float*>(F->io_h + F->io_bert);
    const size_t rows = (size_t)B * hp.bert_dim;
    for (size_t r = 0; r < rows; ++r) {
      memcpy(h_bert + r * TxB, opts->bert + r * Tx, sizeof(float) * Tx);
      for (int t = Tx; t < TxB; ++t) h_bert[r * TxB + t] = 0.f;
    }
  }
  // ---- phase 1 and the one host round trip
  TRY(phase1_launch(F, forced, solo));
  HIP_TRY(hipStreamSynchronize(F->stream));
  {
    int e = 0;
    memcpy(&e, F->h_ylen + B, sizeof(int));
    if (e) {
      hipMemsetAsync(F->d_err, 0, sizeof(int), F->stream);
      return device_error_word(e);
    }
  }
  int64_t Ty = 1;
  for (int b = 0; b < B; ++b) if (F->h_ylen[b] > Ty) Ty = F->h_ylen[b];
  if (opts && opts->max_frames > 0 && Ty > opts->max_frames) return fail(VITS_ERR_ARG, "T_y %lld exceeds max_frames %d", (long long)Ty, opts->max_frames);
  if (Ty > (1 << 24)) return fail(VITS_ERR_ARG, "T_y unreasonably large");
  // frame bucket: multiples of 32 for one utterance; batches round up in steps of 1/8 of the power of two below T_y (64 frames
  // at 512..1023): with free-running durations the longest item of a batch lands on a different multiple of 32 almost every
  // call, and every new bucket is a workspace + a graph capture on the request path.  The padding is not computed (ragged tile
  // maps skip dead tiles); it costs the D2H of the padded rows only.
  int ty_step = 32;
  if (B > 1) { int p2 = 32; while (p2 * 2 <= Ty) p2 *= 2; if (p2 / 8 > ty_step) ty_step = p2 / 8; }
  const int TyB = (int)((Ty + ty_step - 1) / ty_step * ty_step);
  // ---- phase 2
  vits_session* Bk = nullptr;
  {
    int rc = back_get(F, TyB, &Bk);
    if (rc == VITS_ERR_NOMEM) {  // idle fronts of other buckets and this front's other backs go first, then once more
      fronts_evict_all(m);
      hipStreamSynchronize(F->stream);
      for (auto& kv : F->backs) session_free(kv.second);
      F->backs.clear();
      rc = back_get(F, TyB, &Bk);
    }
    if (rc != VITS_OK) return rc;
  }
  TRY(phase2_launch(F, Bk, solo, pcm));
  const int64_t S = Ty * hp.hop_length, stride = (int64_t)TyB * hp.hop_length;
  const size_t esz = pcm ? sizeof(int16_t) : sizeof(float);
  char* h_out = static_cast<char*>(malloc(esz * (size_t)B * S));
  if (!h_out) { hipStreamSynchronize(F->stream); return fail(VITS_ERR_NOMEM, "host alloc failed"); }
  HIP_TRY(hipStreamSynchronize(F->stream));
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) { free(h_out); return fail(VITS_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(le)); }
  if (*Bk->h_err) {
    const int e2 = *Bk->h_err;
    *Bk->h_err = 0;
    hipMemsetAsync(Bk->d_err, 0, sizeof(int), F->stream);
    free(h_out);
    return device_error_word(e2);
  }
  for (int b = 0; b < B; ++b) memcpy(h_out + esz * (size_t)b * S, Bk->out_h + esz * (size_t)b * stride, esz * (size_t)S);
  *out = h_out;
  *out_samples = S;
  if (out_lengths) for (int b = 0; b < B; ++b) out_lengths[b] = F->h_ylen[b] * hp.hop_length;
  return VITS_OK;
}

static int synth_eager(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                       const int64_t* sid, const vits_synth_opts* opts, bool pcm, float pcm_scale, void** out, int64_t* out_samples,
                       int64_t* out_lengths) {
  const vits_hparams& hp = m->hp;
  HostStage hs(m);
  std::vector<int64_t> ylen;
  int64_t Ty = 0;
  float* z = nullptr;
  TRY(acoustic_host(hs, ids, lengths, B, Tx, scales, sid, opts, ylen, Ty, z));
  vits_session* s = hs.s;
  s->ragged = B > 1;
  s->solo = opts && (opts->flags & VITS_FLAG_SOLO_BATCH);
  struct RaggedOff { vits_session* s; ~RaggedOff() { s->ragged = false; s->solo = false; } } ragged_off{s};
  const int64_t S = Ty * hp.hop_length;
  float* d_audio = hs.dev_alloc<float>((size_t)B * S);
  if (!d_audio) return fail(VITS_ERR_NOMEM, "device alloc failed");
  run_decoder(s, z, true, B, (int)Ty, d_audio, S, nullptr, true, s->solo ? 0 : m->rag_halo);
  const size_t esz = pcm ? sizeof(int16_t) : sizeof(float);
  const void* d_src = d_audio;
  if (pcm) {
    int16_t* d_pcm = hs.dev_alloc<int16_t>((size_t)B * S);
    if (!d_pcm) return fail(VITS_ERR_NOMEM, "device alloc failed");
    hipLaunchKernelGGL(pcm16_kernel, dim3(cdiv((int)S, 256), B), dim3(256), 0, s->stream, d_audio, (long long)S, d_pcm, (long long)S, (long long)S, pcm_scale,
                       (const SynthDev*)nullptr);
    d_src = d_pcm;
  }
  void* h_out = malloc(esz * (size_t)B * S);
  if (!h_out) return fail(VITS_ERR_NOMEM, "host alloc failed");
  hipError_t e = hipMemcpyAsync(h_out, d_src, esz * (size_t)B * S, hipMemcpyDeviceToHost, s->stream);
  int rc = e == hipSuccess ? check_err(s) : fail(VITS_ERR_DEVICE, "D2H failed: %s", hipGetErrorString(e));
  if (rc != VITS_OK) { free(h_out); return rc; }
  *out = h_out;
  *out_samples = S;
  if (out_lengths) for (int b = 0; b < B; ++b) out_lengths[b] = ylen[b] * hp.hop_length;
  return VITS_OK;
}

static int synth_dispatch(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                          const int64_t* sid, const vits_synth_opts* opts, bool pcm, float pcm_scale, void** out, int64_t* out_samples,
                          int64_t* out_lengths) {
  if (!m || !ids || !lengths || !scales || !out || !out_samples || B <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  if (!m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  for (int b = 0; b < B; ++b) if (lengths[b] < 0 || lengths[b] > Tx) return fail(VITS_ERR_ARG, "length out of range");
  static const bool env_off = getenv("VITS_NO_FASTPATH") != nullptr;
  if (m->hp.bert_dim > 0 && (!opts || !opts->bert)) return fail(VITS_ERR_ARG, "this voice is BERT-conditioned: the bert feed [B,%d,T_x] is required", m->hp.bert_dim);
  if (m->hp.bert_dim == 0 && opts && opts->bert) return fail(VITS_ERR_ARG, "the bert feed was given but this voice has no BERT projection (hparams.bert_dim == 0)");
  // (round 5: the `bert` feed of a BERT-conditioned voice is an INPUT like the ids and goes through the graph-replayed path; only
  //  injected noise tensors -- parity tests -- take the eager path)
  bool injected = opts && (opts->noise_dp || opts->noise_prior);
  // a large padded batch of a BERT-conditioned voice: its bert feed ([B, 768, T_x], tens of MB) would be pinned once per shape bucket --
  // such calls keep the exact-size eager path (hipMemcpy from the caller's buffer)
  if (m->hp.bert_dim > 0 && (size_t)B * m->hp.bert_dim * ((Tx + 7) / 8 * 8) * sizeof(float) > ((size_t)8 << 20)) injected = true;
  for (int attempt = 0;; ++attempt) {
    tl_ps_timed_out = false;
    const int rc = (g_fast_path && !env_off && !injected)
                       ? synth_fast(m, ids, lengths, B, Tx, scales, sid, opts, pcm, pcm_scale, out, out_samples, out_lengths)
                       : synth_eager(m, ids, lengths, B, Tx, scales, sid, opts, pcm, pcm_scale, out, out_samples, out_lengths);
    if (rc == VITS_OK || !tl_ps_timed_out || attempt) return rc;  // a persistent program timed out: once more, on launches
  }
}

int vits_synthesize(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                    const int64_t* sid, const vits_synth_opts* opts, float** out_audio, int64_t* out_samples, int64_t* out_lengths) {
  return synth_dispatch(m, ids, lengths, B, Tx, scales, sid, opts, false, 1.f, reinterpret_cast<void**>(out_audio), out_samples, out_lengths);
}

int vits_synthesize_pcm16(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                          const int64_t* sid, const vits_synth_opts* opts, float pcm_scale, int16_t** out_pcm, int64_t* out_samples,
                          int64_t* out_lengths) {
  return synth_dispatch(m, ids, lengths, B, Tx, scales, sid, opts, true, pcm_scale, reinterpret_cast<void**>(out_pcm), out_samples, out_lengths);
}

void vits_free_pcm16(int16_t* p) { free(p); }
void vits_debug_fast_path(int on) { g_fast_path = on; }

void vits_free_output(float* p) { free(p); }

