// engine_stream.hip.h -- streaming synthesis (vits_stream_*): first chunk alone, then double-buffered eight-chunk windows.
// Part of the ONE translation unit engine.hip (included there, in order; not a standalone header): split out in round 6 so that the
// planner / launch selection / stages / host paths can be read on their own.
#pragma once
// ---- streaming synthesis (BASELINE configs[4]; the server's `stream AudioChunk`, tts_service.proto:46-54)
// The flow has global attention, so the acoustic half runs once over the whole utterance; the decoder is purely
// convolutional with a receptive field < 25 frames, so it is run on fixed-width frame windows
// [lo - halo, hi + halo) and only the samples of [lo, hi) are emitted -- identical to the one-shot decode.  Every
// window has the same width W = chunk + 2*halo (clamped to the utterance at both ends, where the true zero padding
// applies), so ONE captured hipGraph of the decoder is replayed per chunk; the next chunk is decoded while the
// caller consumes the current one.
struct vits_stream {
  vits_model* m = nullptr;
  HostStage* hs = nullptr;          // acoustic session + temporaries; z lives in its workspace
  const float* z = nullptr;
  int Ty = 0, chunk = 0, W = 0, halo = VITS_RAGGED_HALO;  // halo is set from the model's receptive field at open
  int pos = 0;                      // first frame not yet handed to the caller
  // The first chunk gets its own narrow window (time to first audio); after it ONE wide window of kmax chunks + halo is decoded
  // per slot: a single 192-frame decode is latency-bound (0.6 ms per chunk) and pays the 2 x 32-frame halo per 128 frames, a
  // wide window pays it once per kmax chunks and runs the one-shot's kernels.  Two audio slots: while the chunks of one window
  // are copied out on the session's copy stream, the next window is decoded into the other slot on the compute stream.
  struct Win { int lo = -1, hi = -1, start = -1; float* aud = nullptr; hipEvent_t done = nullptr; };
  Win win[2];
  int kmax = 8, WK = 0;
  bool clamp = false;               // audio clamped to [-1, 1] (the StableTTS export, onnx/export.py:28-31)
  float *d_win = nullptr, *h_pin = nullptr;
  hipEvent_t ev = nullptr;
  int find(int lo) const { for (int i = 0; i < 2; ++i) if (lo >= win[i].lo && lo < win[i].hi) return i; return -1; }
};

// enqueues the decode of the window that starts with chunk `lo` into slot `slot`
static int stream_launch(vits_stream* st, int lo, int slot) {
  vits_session* s = st->hs->s;
  const int I = st->m->hp.inter_channels;
  const bool wide = lo > 0 && st->WK > st->W;
  const int width = wide ? st->WK : st->W;
  int start = lo - st->halo;
  if (start > st->Ty - width) start = st->Ty - width;  // at the end the window is shifted inward: the true zero padding applies
  if (start < 0) start = 0;
  vits_stream::Win& w = st->win[slot];
  w.lo = lo;
  w.hi = wide ? lo + st->kmax * st->chunk : lo + st->chunk;
  w.start = start;
  hipLaunchKernelGGL(window_copy_kernel, dim3(cdiv(width, 256), I), dim3(256), 0, s->stream, st->z, (long long)st->Ty, start, width, st->d_win);
  const long long S = (long long)width * st->m->hp.hop_length;
  run_decoder(s, st->d_win, false, 1, width, w.aud, S, nullptr);
  if (st->clamp) hipLaunchKernelGGL(clamp_kernel, dim3(cdiv((int)S, 256)), dim3(256), 0, s->stream, w.aud, S);
  HIP_TRY(hipEventRecord(w.done, s->stream));
  return VITS_OK;
}

void vits_stream_close(vits_stream* st) {
  if (!st) return;
  hipSetDevice(st->m->device);
  if (st->hs && st->hs->s) hipStreamSynchronize(st->hs->s->stream);
  if (st->hs && st->hs->s && st->hs->s->copy_stream) hipStreamSynchronize(st->hs->s->copy_stream);
  for (auto& w : st->win) if (w.done) hipEventDestroy(w.done);
  if (st->ev) hipEventDestroy(st->ev);
  if (st->h_pin) hipHostFree(st->h_pin);
  delete st->hs;  // frees d_win/d_aud and returns the session to the pool
  delete st;
}

// second half of every stream open: window geometry, buffers, first chunk in flight.  Closes the stream on failure.
static int stream_start(vits_stream* st, const float* z, int Ty, int chunk_frames, vits_stream** out, int64_t* total_samples) {
  vits_model* m = st->m;
  int rc = VITS_OK;
  const vits_hparams& hp = m->hp;
  st->z = z;
  st->Ty = (int)Ty;
  st->halo = m->rag_halo;
  st->chunk = chunk_frames;
  st->W = chunk_frames + 2 * st->halo;
  if (st->W > st->Ty) st->W = st->Ty;
  st->WK = st->kmax * chunk_frames + 2 * st->halo;  // the wide window of the chunks after the first
  if (st->WK > st->Ty) st->WK = st->Ty;
  st->d_win = st->hs->dev_alloc<float>((size_t)hp.inter_channels * st->WK);
  for (auto& w : st->win) w.aud = st->hs->dev_alloc<float>((size_t)st->WK * hp.hop_length);
  if (!st->hs->s->copy_stream && hipStreamCreateWithFlags(&st->hs->s->copy_stream, hipStreamNonBlocking) != hipSuccess) {
    vits_stream_close(st);
    return fail(VITS_ERR_DEVICE, "stream: hipStreamCreate failed");
  }
  if (!st->d_win || !st->win[0].aud || !st->win[1].aud ||
      hipEventCreateWithFlags(&st->win[0].done, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&st->win[1].done, hipEventDisableTiming) != hipSuccess || hipHostMalloc((void**)&st->h_pin, sizeof(float) * (size_t)chunk_frames * hp.hop_length) != hipSuccess ||
      hipEventCreateWithFlags(&st->ev, hipEventDisableTiming) != hipSuccess) {
    vits_stream_close(st);
    return fail(VITS_ERR_NOMEM, "stream buffers");
  }
  // the acoustic half is done with the persistent stages: wait for them and hand the token back (the stream object lives on)
  hipStreamSynchronize(st->hs->s->stream);
  st->hs->pscope.release();
  rc = stream_launch(st, 0, 0);  // first chunk is already decoding when the caller asks for it
  if (rc != VITS_OK) { vits_stream_close(st); return rc; }
  if (total_samples) *total_samples = (int64_t)Ty * hp.hop_length;
  *out = st;
  return VITS_OK;
}


int vits_stream_open(vits_model* m, const int64_t* ids, int32_t Tx, const float* scales, int64_t sid, const vits_synth_opts* opts,
                     int32_t chunk_frames, vits_stream** out, int64_t* total_samples) {
  if (!m || !ids || !scales || !out || Tx <= 0 || chunk_frames <= 0) return fail(VITS_ERR_ARG, "bad argument");
  for (int attempt = 0;; ++attempt) {
    vits_stream* st = new vits_stream();
    st->m = m;
    st->hs = new HostStage(m);
    std::vector<int64_t> ylen;
    int64_t Ty = 0, len = Tx;
    float* z = nullptr;
    tl_ps_timed_out = false;
    const int rc = acoustic_host(*st->hs, ids, &len, 1, Tx, scales, &sid, opts, ylen, Ty, z);
    if (rc == VITS_OK) return stream_start(st, z, (int)Ty, chunk_frames, out, total_samples);
    vits_stream_close(st);
    if (!tl_ps_timed_out || attempt) return rc;  // a persistent program timed out: once more, on launches
  }
}

// Streams the decoder over a latent the caller already holds (host, [inter_channels, T_y] row-major): the vocoder half of a
// two-model voice (StableTTS mel -> vocoder, vosk_tts/synth.py:113-126) or a z produced elsewhere.  flags bit 0: clamp to [-1, 1].
int vits_stream_open_latent(vits_model* m, const float* z, int32_t Ty, int32_t chunk_frames, uint32_t flags, vits_stream** out,
                            int64_t* total_samples) {
  if (!m || !z || !out || Ty <= 0 || chunk_frames <= 0) return fail(VITS_ERR_ARG, "bad argument");
  if (Ty > (1 << 18)) return fail(VITS_ERR_ARG, "T_y unreasonably large");
  vits_stream* st = new vits_stream();
  st->m = m;
  st->hs = new HostStage(m);
  st->clamp = (flags & 1u) != 0;
  int rc = begin_stage(*st->hs, 1, 1, Ty, 0);
  if (rc != VITS_OK) { vits_stream_close(st); return rc; }
  const float* d_z = st->hs->to_dev(z, (size_t)m->hp.inter_channels * Ty);
  if (!d_z) { vits_stream_close(st); return fail(VITS_ERR_NOMEM, "device alloc failed"); }
  return stream_start(st, d_z, Ty, chunk_frames, out, total_samples);
}

int vits_stream_next(vits_stream* st, float* audio, int64_t capacity, int64_t* n_samples) {
  if (!st || !audio || !n_samples) return fail(VITS_ERR_ARG, "bad argument");
  *n_samples = 0;
  if (st->pos >= st->Ty) return VITS_OK;  // end of stream
  HIP_TRY(hipSetDevice(st->m->device));
  vits_session* s = st->hs->s;
  const int hop = st->m->hp.hop_length;
  const int lo = st->pos, hi = lo + st->chunk < st->Ty ? lo + st->chunk : st->Ty;
  const int64_t n = (int64_t)(hi - lo) * hop;
  if (capacity < n) return fail(VITS_ERR_ARG, "chunk capacity %lld < %lld samples", (long long)capacity, (long long)n);
  int slot = st->find(lo);
  if (slot < 0) { slot = 0; TRY(stream_launch(st, lo, slot)); }  // only the first call: later windows are decoded ahead
  const vits_stream::Win& w = st->win[slot];
  HIP_TRY(hipStreamWaitEvent(s->copy_stream, w.done, 0));
  HIP_TRY(hipMemcpyAsync(st->h_pin, w.aud + (size_t)(lo - w.start) * hop, sizeof(float) * n, hipMemcpyDeviceToHost, s->copy_stream));
  HIP_TRY(hipEventRecord(st->ev, s->copy_stream));
  st->pos = hi;
  // decode ahead into the other slot: every chunk of the window it held was handed over (and waited for) before this call
  if (w.hi < st->Ty && st->find(w.hi) < 0) TRY(stream_launch(st, w.hi, slot ^ 1));
  HIP_TRY(hipEventSynchronize(st->ev));
  memcpy(audio, st->h_pin, sizeof(float) * n);
  *n_samples = n;
  return VITS_OK;
}

