// engine.hip — MI355X-native VITS2 inference engine behind the C ABI of include/vits_mi355.h.
//
// Replaces onnxruntime.InferenceSession.run() at vosk_tts/synth.py:123-126 for the graph that
// training/vits2/onnx_export.py exports from SynthesizerTrn.infer (training/vits2/models.py:1679-1704).
// One process per GPU; weights resident in HBM in MFMA-fragment order; every stage is a short
// sequence of hand-written HIP kernels on one stream (no host round trip when durations are
// forced or a frame capacity is given), optionally replayed as a hipGraph.
//
// There is NO CPU fallback here: every entry point either runs the HIP kernels or returns an error.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/vits_mi355.h"
#include "../../include/vits_mi355_debug.h"
#include "conv_mfma.hip.h"
#include "conv_small.hip.h"
#include "conv_sp.hip.h"
#include "conv_w1.hip.h"
#include "conv_bf3.hip.h"
#include "kernels_misc.hip.h"
#include "persist.hip.h"
#include "conv_sk.hip.h"

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512];
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(VITS_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define TRY(expr)              \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != VITS_OK) return rc_; \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- test hooks ------------------------------------------------------------------------------------
// Kernel-selection / path switches are THREAD-LOCAL (round 6): a hook changes what the calling thread's engine calls launch, never what
// another server thread is running.  (They were process-wide statics; the engine runs every call on the caller's thread.)  The
// persistent-program switches below stay process-wide on purpose: they describe a per-device resource, and they are atomics.
// Declarations: include/vits_mi355_debug.h (not part of the installed ABI).
// 0 = size heuristic, 1 = force the big-tile kernel, 2 = force the K-split kernel (tests only)
static thread_local int g_force_tile = 0;
// 0 = fp32-MFMA flash attention (default), 1 = the VALU kernel (kept as an independent cross-check in tests)
static thread_local int g_attn_impl = 0;
// tests: fill every freshly planned workspace with NaN so stale padding can never hide as zeros
static thread_local int g_poison = 0;
// 0 = fused exp/sin + iSTFT + PQMF kernel (default), 1 = the two separate kernels (independent cross-check in tests)
static thread_local int g_tail_impl = 0;
// 1 = folded WN tail (stacked gate outputs, one skip+post conv; default), 0 = per-layer res/skip epilogue + post
static thread_local int g_wn_fold = 1;
// 1 = LayerNorm statistics of the folded encoder LayerNorms from the producer conv's epilogue (default), 0 = redone by the consumer
static thread_local int g_ln_stats = 1;
static std::atomic<int> g_ps_spin_limit{0};  // test hook (vits_debug_persist_spin): poll rounds before a persistent worker gives up; 0 = PS_SPIN_LIMIT
// single-utterance duration predictor as one persistent kernel (persist.hip.h): 1 = when eligible (default), 0 = launch path
// A persistent kernel needs ALL its workgroups resident at once (they spin on each other's cells): two of them in flight on one
// device could each hold half of the CUs and wait forever (the bounded poll loops turn that into an error, not a hang -- but it must
// not happen in normal operation).  So at most ONE caller per device owns the persistent path at a time (a token); everybody else
// takes the launch path for that call.  persist_mask() is what the stage launchers test: the owner's mask, 0 for everybody else.
#define PERSIST_OWNERS_DEFAULT 1
static std::mutex g_tok_mu;
static int g_tok_busy[64];  // owners of the device's persistent path right now (round 6: up to persist_owners() of them, was a flag)
// How many callers may run persistent programs on ONE device at the same time.  A program's 256 workgroups (512 threads, ~60 KB of LDS,
// <= 128 registers per thread) fill exactly HALF of every CU, so two programs are co-resident on the whole chip and neither waits for the
// other; a third could only be placed where one of them has finished.  VITS_PERSIST_OWNERS (1..4); measured in profiles/r6_owners.txt.
static int persist_owners() {
  static const int n = getenv("VITS_PERSIST_OWNERS") ? atoi(getenv("VITS_PERSIST_OWNERS")) : PERSIST_OWNERS_DEFAULT;
  return n < 1 ? 1 : (n > 4 ? 4 : n);
}
static thread_local int tl_persist = -1;  // >= 0: this thread's mask for the call in progress
static std::atomic<int> g_persist{getenv("VITS_NO_PERSIST") ? 0 : (getenv("VITS_PERSIST") ? atoi(getenv("VITS_PERSIST")) : 7)};  // mask: 1 duration predictor, 2 text encoder, 4 flow (environment switches: A/B runs of bench.py and tools/)
// A poll timeout (persist_timed_out) switches the programs off for a BOUNDED interval, not for the life of the process: a server that
// once lost co-residency (another process on the device, a transient) gets them back.  The interval starts at VITS_PERSIST_REARM_MS
// (default 1000) and doubles with every timeout that follows a re-arm within 10 intervals (cap: 64 x), so a device that is shared for
// good costs one failed forward per minute, not one per second.  persist_cfg() is the mask in effect now; vits_persist_state reports.
static std::atomic<long long> g_ps_rearm_base_ns{(getenv("VITS_PERSIST_REARM_MS") ? atoll(getenv("VITS_PERSIST_REARM_MS")) : 1000) * 1000000LL};
static std::atomic<long long> g_ps_off_until_ns{0};  // steady-clock ns; 0 = armed
static std::atomic<long long> g_ps_rearmed_at_ns{0};
static std::atomic<long long> g_ps_rearm_ns{0};      // current interval (0 = base)
static std::atomic<int> g_ps_timeouts{0}, g_ps_rearms{0};
static inline long long steady_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static int persist_cfg() {
  const int cfg = g_persist.load(std::memory_order_relaxed);
  if (!cfg) return 0;
  long long until = g_ps_off_until_ns.load(std::memory_order_relaxed);
  if (!until) return cfg;
  const long long now = steady_ns();
  if (now < until) return 0;
  if (g_ps_off_until_ns.compare_exchange_strong(until, 0)) { g_ps_rearms.fetch_add(1); g_ps_rearmed_at_ns.store(now); }
  return cfg;
}

#include "engine_model.hip.h"
#include "engine_session.hip.h"
#include "engine_launch.hip.h"
#include "engine_stages.hip.h"

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

int vits_is_device_backend(void) { return 1; }
int vits_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}
const char* vits_last_error(void) { return g_err; }

int vits_create(const void* blob, size_t bytes, int device, vits_model** out) {
  if (!blob || !out || bytes < 16 + sizeof(vits_hparams)) return fail(VITS_ERR_ARG, "bad blob argument");
  const unsigned char* p = static_cast<const unsigned char*>(blob);
  if (memcmp(p, "VITSW001", 8) != 0) return fail(VITS_ERR_BLOB, "bad magic");
  uint32_t hb;
  memcpy(&hb, p + 8, 4);
  if (hb != sizeof(vits_hparams)) return fail(VITS_ERR_BLOB, "hparams size %u != %zu", hb, sizeof(vits_hparams));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VITS_ERR_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(VITS_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  vits_model* m = new vits_model();
  memcpy(&m->hp, p + 12, sizeof(vits_hparams));
  m->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) m->n_cu = cus > 256 ? 256 : cus;
  }
  if (m->hp.abi_version != VITS_ABI_VERSION) { delete m; return fail(VITS_ERR_BLOB, "abi version mismatch"); }
  m->blob = p; m->blob_bytes = bytes;
  memcpy(&m->n_entries, p + 12 + hb, 4);
  m->entries = reinterpret_cast<const vits_blob_entry*>(p + 16 + hb);
  if (16 + hb + (size_t)m->n_entries * sizeof(vits_blob_entry) > bytes) { delete m; return fail(VITS_ERR_BLOB, "truncated table"); }
  for (uint32_t i = 0; i < m->n_entries; ++i) {  // overflow-safe: nelem and offset are 64-bit values from the file
    const uint64_t off = m->entries[i].offset, ne = m->entries[i].nelem;
    if (off > bytes || ne > (bytes - off) / 4) { delete m; return fail(VITS_ERR_BLOB, "truncated data"); }
  }
  int rc = load_model(m);
  if (rc == VITS_OK) {
    std::vector<float> z(4096, 0.f);
    m->zeros = upload(m, z.data(), z.size());
    if (!m->zeros) rc = VITS_ERR_NOMEM;
    m->ps_dbg = reinterpret_cast<int*>(upload(m, z.data(), 16));
    if (!m->ps_dbg) rc = VITS_ERR_NOMEM;
    else if (const int lim = g_ps_spin_limit.load(); lim > 0) hipMemcpy(m->ps_dbg, &lim, sizeof(int), hipMemcpyHostToDevice);
  }
  m->blob = nullptr; m->entries = nullptr;
  if (rc != VITS_OK) { for (void* a : m->allocs) hipFree(a); delete m; return rc; }
  hipDeviceSynchronize();
  { std::lock_guard<std::mutex> g(g_models_mu); g_models.push_back(m); }
  *out = m;
  return VITS_OK;
}

void vits_destroy(vits_model* m) {
  if (!m) return;
  bool last_on_device = true;
  {
    std::lock_guard<std::mutex> g(g_models_mu);
    g_models.erase(std::remove(g_models.begin(), g_models.end(), m), g_models.end());
    for (vits_model* o : g_models) if (o->device == m->device) last_on_device = false;
  }
  hipSetDevice(m->device);
  for (vits_session* s : m->pool) session_free(s);
  for (auto& kv : m->fronts) session_free(kv.second);
  for (void* a : m->allocs) hipFree(a);
  const int dev = m->device;
  delete m;
  if (last_on_device) { hipDeviceSynchronize(); persist_process_release(dev); }  // (nothing of this process runs a program there any more; re-checked under the token mutex)
}

int vits_get_hparams(const vits_model* m, vits_hparams* out) {
  if (!m || !out) return fail(VITS_ERR_ARG, "null argument");
  *out = m->hp;
  return VITS_OK;
}

double vits_algorithmic_flops(const vits_model* m, int32_t B, int32_t Tx, int32_t Ty) {
  const vits_hparams* hp = &m->hp;
  double H = hp->hidden_channels, I = hp->inter_channels, F = hp->filter_channels, D = hp->dp_filter_channels;
  double NW = 2 * hp->window_size + 1;
  double enc_layer = 2 * (4 * H * H) + 2 * (2 * H * F * hp->kernel_size) + 2 * 2 * NW * H;
  double tok = hp->n_layers * enc_layer + 2 * H * 2 * I;
  double dds = hp->dp_dds_layers * (2 * D * hp->dp_kernel_size + 2 * D * D);
  tok += 2 * H * D + 2 * D * D + dds + (hp->dp_n_flows - 1) * (2 * D + dds + 2 * D * (3 * hp->dp_num_bins - 1));
  double tok_quad = hp->n_layers * 4 * H;
  double K5 = hp->flow_kernel_size;
  double fl = 2 * (I / 2) * H + (2 * (4 * H * H) + 2 * (2 * H * H * K5) + 2 * 2 * NW * H);
  for (int i = 0; i < hp->flow_wn_layers; ++i) fl += 2 * H * 2 * H * K5 + 2 * H * (i < hp->flow_wn_layers - 1 ? 2 * H : H);
  fl += 2 * H * (I / 2);
  double frame = hp->flow_n_flows * fl, frame_quad = hp->flow_n_flows * 4 * H;
  double C = hp->dec_initial_channel, rate = 1, dec = 2 * I * C * 7;
  for (int i = 0; i < hp->n_ups; ++i) {
    dec += rate * 2 * C * (C / 2) * hp->up_kernels[i];
    rate *= hp->up_rates[i];
    C /= 2;
    for (int j = 0; j < hp->n_resk; ++j) dec += rate * hp->n_resd * 2 * (2 * C * C * hp->res_kernels[j]);
  }
  if (hp->dec_type == 0) {
    double P = hp->subbands * (hp->istft_n_fft + 2);
    dec += rate * 2 * C * P * 7;
    dec += rate * hp->subbands * 2 * (hp->istft_n_fft + 2) * hp->istft_n_fft;
    dec += rate * hp->subbands * hp->istft_hop * 2 * (hp->pqmf_taps + 1);
  } else {
    dec += rate * 2 * C * 7;
  }
  frame += dec;
  return (double)B * ((double)Tx * (tok + tok_quad * Tx) + (double)Ty * (frame + frame_quad * Ty));
}

int vits_stage_text_encoder(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const int64_t* sid,
                            float* x, float* m_p, float* logs_p) {
  if (m && !m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  if (!m || !ids || !lengths || !x || !m_p || !logs_p || B <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  for (int b = 0; b < B; ++b) if (lengths[b] < 0 || lengths[b] > Tx) return fail(VITS_ERR_ARG, "length out of range");
  HostStage hs(m);
  TRY(begin_stage(hs, B, Tx, 1, PERSIST_ENC));
  vits_session* s = hs.s;
  const int H = m->hp.hidden_channels, I = m->hp.inter_channels;
  int64_t* d_ids = hs.to_dev(ids, (size_t)B * Tx);
  int64_t* d_len = hs.to_dev(lengths, B);
  int64_t* d_sid = hs.to_dev(sid, B);
  if (!d_ids || !d_len) return fail(VITS_ERR_NOMEM, "device alloc failed");
  set_lengths(s, d_len, s->len_x, B, Tx);
  run_cond(s, d_sid, B);
  run_text_encoder(s, d_ids, B, Tx);
  HIP_TRY(hipMemcpyAsync(x, s->x, sizeof(float) * (size_t)B * H * Tx, hipMemcpyDeviceToHost, s->stream));
  for (int b = 0; b < B; ++b) {
    HIP_TRY(hipMemcpyAsync(m_p + (size_t)b * I * Tx, s->stats + (size_t)b * 2 * I * Tx, sizeof(float) * (size_t)I * Tx, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(logs_p + (size_t)b * I * Tx, s->stats + ((size_t)b * 2 * I + I) * Tx, sizeof(float) * (size_t)I * Tx, hipMemcpyDeviceToHost, s->stream));
  }
  return check_err(s);
}

int vits_stage_duration(vits_model* m, const float* x, const int64_t* lengths, int32_t B, int32_t Tx, const int64_t* sid,
                        const float* noise, float noise_scale_w, float* logw) {
  if (m && !m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  if (!m || !x || !lengths || !noise || !logw || B <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  HostStage hs(m);
  TRY(begin_stage(hs, B, Tx, 1, PERSIST_SDP));
  vits_session* s = hs.s;
  const int H = m->hp.hidden_channels;
  float* d_x = hs.to_dev(x, (size_t)B * H * Tx);
  int64_t* d_len = hs.to_dev(lengths, B);
  int64_t* d_sid = hs.to_dev(sid, B);
  float* d_noise = hs.to_dev(noise, (size_t)B * 2 * Tx);
  if (!d_x || !d_len || !d_noise) return fail(VITS_ERR_NOMEM, "device alloc failed");
  set_lengths(s, d_len, s->len_x, B, Tx);
  run_cond(s, d_sid, B);
  run_duration(s, d_x, d_noise, noise_scale_w, 0, B, Tx);
  HIP_TRY(hipMemcpyAsync(logw, s->logw, sizeof(float) * (size_t)B * Tx, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

int vits_stage_regulate(vits_model* m, const float* logw, const int32_t* forced, const int64_t* lengths, int32_t B, int32_t Tx,
                        float length_scale, const float* m_p, const float* logs_p, const float* noise, float noise_scale,
                        int32_t Tcap, int32_t* durations, int64_t* y_lengths, float* z_p) {
  if (m && !m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  if (!m || !lengths || !durations || !y_lengths || (!logw && !forced) || B <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  if (z_p && (!m_p || !logs_p || Tcap <= 0)) return fail(VITS_ERR_ARG, "m_p/logs_p/T_cap required");
  HostStage hs(m);
  TRY(begin_stage(hs, B, Tx, Tcap > 0 ? Tcap : 1, 0));
  vits_session* s = hs.s;
  const int I = m->hp.inter_channels;
  int64_t* d_len = hs.to_dev(lengths, B);
  int* d_forced = hs.to_dev(forced, (size_t)B * Tx);
  if (logw) HIP_TRY(hipMemcpyAsync(s->logw, logw, sizeof(float) * (size_t)B * Tx, hipMemcpyHostToDevice, s->stream));
  set_lengths(s, d_len, s->len_x, B, Tx);
  run_durations(s, d_forced, length_scale, B, Tx, z_p ? Tcap : 0);
  if (z_p) {
    for (int b = 0; b < B; ++b) {
      HIP_TRY(hipMemcpyAsync(s->stats + (size_t)b * 2 * I * Tx, m_p + (size_t)b * I * Tx, sizeof(float) * (size_t)I * Tx, hipMemcpyHostToDevice, s->stream));
      HIP_TRY(hipMemcpyAsync(s->stats + ((size_t)b * 2 * I + I) * Tx, logs_p + (size_t)b * I * Tx, sizeof(float) * (size_t)I * Tx, hipMemcpyHostToDevice, s->stream));
    }
    float* d_noise = hs.to_dev(noise, (size_t)B * I * Tcap);
    float* d_zero = nullptr;
    if (!d_noise) {  // noise == NULL means eps = 0 here (stage API), not Philox
      d_zero = hs.dev_alloc<float>((size_t)B * I * Tcap);
      if (!d_zero) return fail(VITS_ERR_NOMEM, "device alloc failed");
      HIP_TRY(hipMemsetAsync(d_zero, 0, sizeof(float) * (size_t)B * I * Tcap, s->stream));
      d_noise = d_zero;
    }
    run_expand(s, d_noise, Tcap, noise_scale, 0, s->zA, B, Tx, Tcap);
    HIP_TRY(hipMemcpyAsync(z_p, s->zA, sizeof(float) * (size_t)B * I * Tcap, hipMemcpyDeviceToHost, s->stream));
  }
  HIP_TRY(hipMemcpyAsync(durations, s->dur, sizeof(int) * (size_t)B * Tx, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(y_lengths, s->ylen64, sizeof(int64_t) * B, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

int vits_stage_flow(vits_model* m, const float* z_p, const int64_t* y_lengths, int32_t B, int32_t Ty, const int64_t* sid, float* z) {
  if (m && !m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  if (!m || !z_p || !y_lengths || !z || B <= 0 || Ty <= 0) return fail(VITS_ERR_ARG, "bad argument");
  HostStage hs(m);
  TRY(begin_stage(hs, B, 1, Ty, PERSIST_FLOW));
  vits_session* s = hs.s;
  const int I = m->hp.inter_channels;
  int64_t* d_len = hs.to_dev(y_lengths, B);
  int64_t* d_sid = hs.to_dev(sid, B);
  HIP_TRY(hipMemcpyAsync(s->zA, z_p, sizeof(float) * (size_t)B * I * Ty, hipMemcpyHostToDevice, s->stream));
  set_lengths(s, d_len, s->len_y, B, Ty);
  run_cond(s, d_sid, B);
  float* r = run_flow(s, B, Ty);
  HIP_TRY(hipMemcpyAsync(z, r, sizeof(float) * (size_t)B * I * Ty, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

int vits_stage_decoder(vits_model* m, const float* z, int32_t B, int32_t Ty, const int64_t* sid, float* audio, float* audio_mb) {
  if (!m || !z || !audio || B <= 0 || Ty <= 0) return fail(VITS_ERR_ARG, "bad argument");
  HostStage hs(m);
  TRY(begin_stage(hs, B, 1, Ty, 0));
  vits_session* s = hs.s;
  const vits_hparams& hp = m->hp;
  const int I = hp.inter_channels;
  const long long S = (long long)Ty * hp.hop_length;
  HIP_TRY(hipMemcpyAsync(s->zA, z, sizeof(float) * (size_t)B * I * Ty, hipMemcpyHostToDevice, s->stream));
  if (m->cond_dec_off >= 0) { int64_t* d_sid = hs.to_dev(sid, B); run_cond(s, d_sid, B); }
  float* d_audio = hs.dev_alloc<float>((size_t)B * S);
  if (!d_audio) return fail(VITS_ERR_NOMEM, "device alloc failed");
  run_decoder(s, s->zA, false, B, Ty, d_audio, S, nullptr);
  HIP_TRY(hipMemcpyAsync(audio, d_audio, sizeof(float) * (size_t)B * S, hipMemcpyDeviceToHost, s->stream));
  if (audio_mb && hp.dec_type == 0)
    HIP_TRY(hipMemcpyAsync(audio_mb, s->dec_bufs[16], sizeof(float) * (size_t)B * S, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

// ---- the hot path, host buffers (SynthesizerTrn.infer, models.py:1679-1704)
// Everything up to and including the flow (models.py:1680-1701) for host inputs: leaves z [B,inter,T_y] in the
// session workspace (masked by the decoder's first staging), the per-item frame counts in ylen.
static int acoustic_host(HostStage& hs, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                         const int64_t* sid, const vits_synth_opts* opts, std::vector<int64_t>& ylen, int64_t& Ty_out, float*& z_out) {
  if (!hs.m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  vits_model* m = hs.m;
  const vits_hparams& hp = m->hp;
  const int I = hp.inter_channels;
  const float noise_scale = scales[0], length_scale = scales[1], noise_scale_w = scales[2];
  const uint64_t seed = opts ? opts->seed : 0;
  TRY(begin_stage(hs, B, Tx, 1, PERSIST_ENC | PERSIST_SDP));  // (text side first: no flow program for a one-frame layout)
  vits_session* s = hs.s;
  int64_t* d_ids = hs.to_dev(ids, (size_t)B * Tx);
  int64_t* d_len = hs.to_dev(lengths, B);
  int64_t* d_sid = hs.to_dev(sid, B);
  if (!d_ids || !d_len) return fail(VITS_ERR_NOMEM, "device alloc failed");
  s->ragged = B > 1;
  s->solo = opts && (opts->flags & VITS_FLAG_SOLO_BATCH);
  s->item_seeds = nullptr;
  if (s->solo && opts->item_seeds) {
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "seed width");
    s->item_seeds = hs.to_dev(reinterpret_cast<const unsigned long long*>(opts->item_seeds), (size_t)B);
  }
  s->tile_keys.clear();
  struct RaggedOff { vits_session* s; ~RaggedOff() { s->ragged = false; s->solo = false; } } ragged_off{s};
  float* d_bert = nullptr;
  if (hp.bert_dim > 0) {
    if (!opts || !opts->bert) return fail(VITS_ERR_ARG, "this voice is BERT-conditioned: the bert feed [B,%d,T_x] is required", hp.bert_dim);
    d_bert = hs.to_dev(opts->bert, (size_t)B * hp.bert_dim * Tx);
    if (!d_bert) return fail(VITS_ERR_NOMEM, "device alloc failed");
  } else if (opts && opts->bert) {
    return fail(VITS_ERR_ARG, "the bert feed was given but this voice has no BERT projection (hparams.bert_dim == 0)");
  }
  run_cond(s, d_sid, B, d_len, s->len_x, Tx);
  run_text_encoder(s, d_ids, B, Tx, d_bert);
  int* d_forced = nullptr;
  if (opts && opts->forced_durations) {
    d_forced = hs.to_dev(opts->forced_durations, (size_t)B * Tx);
  } else {
    float* d_ndp = (opts && opts->noise_dp) ? hs.to_dev(opts->noise_dp, (size_t)B * 2 * Tx) : nullptr;
    run_duration(s, s->x, d_ndp, noise_scale_w, seed, B, Tx, true);
  }
  run_durations(s, d_forced, length_scale, B, Tx, 0);
  // the one host round trip of the free-running path: T_y sizes everything downstream
  ylen.assign(B, 0);
  HIP_TRY(hipMemcpyAsync(ylen.data(), s->ylen64, sizeof(int64_t) * B, hipMemcpyDeviceToHost, s->stream));
  TRY(check_err(s));
  int64_t Ty = 1;
  for (int b = 0; b < B; ++b) if (ylen[b] > Ty) Ty = ylen[b];
  if (opts && opts->max_frames > 0 && Ty > opts->max_frames) return fail(VITS_ERR_ARG, "T_y %lld exceeds max_frames %d", (long long)Ty, opts->max_frames);
  if (Ty > (1 << 24)) return fail(VITS_ERR_ARG, "T_y unreasonably large");
  // grow the workspace for T_y: encoder outputs live in the arena, so keep them across the re-plan
  const int H = hp.hidden_channels;
  float* keep_stats = hs.dev_alloc<float>((size_t)B * 2 * I * Tx);
  int* keep_cum = hs.dev_alloc<int>((size_t)B * Tx);
  if (!keep_stats || !keep_cum) return fail(VITS_ERR_NOMEM, "device alloc failed");
  HIP_TRY(hipMemcpyAsync(keep_stats, s->stats, sizeof(float) * (size_t)B * 2 * I * Tx, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(keep_cum, s->cum, sizeof(int) * (size_t)B * Tx, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->ps_roles = PERSIST_FLOW;  // (frame side: the text-side programs are not built again)
  TRY(session_reserve(s, B, Tx, (int)Ty));
  s->tile_keys.clear();  // the tile tables live in the (re-planned) workspace
  HIP_TRY(hipMemcpyAsync(s->stats, keep_stats, sizeof(float) * (size_t)B * 2 * I * Tx, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(s->cum, keep_cum, sizeof(int) * (size_t)B * Tx, hipMemcpyDeviceToDevice, s->stream));
  std::vector<int> ylen32(B);
  for (int b = 0; b < B; ++b) ylen32[b] = (int)ylen[b];
  HIP_TRY(hipMemcpyAsync(s->len_y, ylen32.data(), sizeof(int) * B, hipMemcpyHostToDevice, s->stream));
  set_lengths(s, d_len, s->len_x, B, Tx);
  run_cond(s, d_sid, B);
  (void)H;
  float* d_npr = nullptr;
  long long nstride = Ty;
  if (opts && opts->noise_prior) {
    if (opts->noise_prior_stride < Ty) return fail(VITS_ERR_ARG, "noise_prior stride %lld < T_y %lld", (long long)opts->noise_prior_stride, (long long)Ty);
    nstride = opts->noise_prior_stride;
    d_npr = hs.to_dev(opts->noise_prior, (size_t)B * I * nstride);
    if (!d_npr) return fail(VITS_ERR_NOMEM, "device alloc failed");
  }
  run_expand(s, d_npr, nstride, noise_scale, seed, s->zA, B, Tx, (int)Ty);
  z_out = run_flow(s, B, (int)Ty);
  Ty_out = Ty;
  return VITS_OK;
}

#include "engine_fastpath.hip.h"
#include "engine_stream.hip.h"
// ---- monotonic alignment search (monotonic_align/core.pyx:7-42), host buffers
int vits_mas_maximum_path(int device, const float* values, const int32_t* t_ys, const int32_t* t_xs, int32_t B, int32_t Ty,
                          int32_t Tx, int32_t* paths) {
  if (!values || !t_ys || !t_xs || !paths || B <= 0 || Ty <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  for (int b = 0; b < B; ++b)
    if (t_ys[b] < 0 || t_ys[b] > Ty || t_xs[b] < 0 || t_xs[b] > Tx) return fail(VITS_ERR_ARG, "extent out of range");
  if ((size_t)2 * Tx * sizeof(float) > 160 * 1024) return fail(VITS_ERR_ARG, "T_x %d too large for the LDS row buffers", Tx);
  HIP_TRY(hipSetDevice(device));
  const size_t n = (size_t)B * Ty * Tx;
  float* d_v = nullptr; int *d_ty = nullptr, *d_tx = nullptr, *d_p = nullptr; unsigned char* d_d = nullptr;
  struct Free { std::vector<void*> p; ~Free() { for (void* q : p) hipFree(q); } } fr;
  auto alloc = [&](void** q, size_t bytes) { if (hipMalloc(q, bytes) != hipSuccess) return false; fr.p.push_back(*q); return true; };
  if (!alloc((void**)&d_v, n * 4) || !alloc((void**)&d_p, n * 4) || !alloc((void**)&d_d, n) || !alloc((void**)&d_ty, B * 4) ||
      !alloc((void**)&d_tx, B * 4))
    return fail(VITS_ERR_NOMEM, "device alloc failed");
  HIP_TRY(hipMemcpy(d_v, values, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_ty, t_ys, B * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_tx, t_xs, B * 4, hipMemcpyHostToDevice));
  const size_t lds = (size_t)2 * Tx * sizeof(float);
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)mas_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mas_kernel, dim3(B), dim3(256), lds, 0, d_v, d_ty, d_tx, Ty, Tx, d_d, d_p);
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(VITS_ERR_DEVICE, "launch failed: %s", hipGetErrorString(le));
  HIP_TRY(hipMemcpy(paths, d_p, n * 4, hipMemcpyDeviceToHost));
  return VITS_OK;
}

// ---- device-resident sessions (bench / serving loop)
int vits_session_create(vits_model* m, int32_t max_B, int32_t max_Tx, int32_t max_Ty, vits_session** out) {
  if (!m || !out || max_B <= 0 || max_Tx <= 0 || max_Ty <= 0) return fail(VITS_ERR_ARG, "bad argument");
  HIP_TRY(hipSetDevice(m->device));
  vits_session* s = nullptr;
  TRY(session_new(m, &s));
  int rc = session_reserve(s, max_B, max_Tx, max_Ty);
  if (rc != VITS_OK) { session_free(s); return rc; }
  // the asynchronous entry point cannot hand the token back per call (nobody waits for the kernels): the first device session of a
  // device keeps it until it is destroyed; others (and host calls in the meantime) run the launch path
  s->ps_owner = g_persist != 0 && persist_token_try(m->device);  // (the token, not the mask: persist_cfg() decides per call)
  *out = s;
  return VITS_OK;
}

void vits_session_destroy(vits_session* s) {
  if (s && s->ps_owner) { hipStreamSynchronize(s->stream); persist_token_release(s->m->device); }
  session_free(s);
}

int vits_session_synthesize_device(vits_session* s, const int64_t* d_ids, const int64_t* d_lengths, int32_t B, int32_t Tx,
                                   const float* scales, const int64_t* d_sid, const int32_t* d_forced, int32_t Ty, uint64_t seed,
                                   float* d_audio, int64_t cap, void* stream) {
  if (s && !s->m->acoustic) return fail(VITS_ERR_UNSUPPORTED, "vocoder-only model: only the decoder stage is available");
  if (!s || !d_ids || !d_lengths || !scales || !d_audio || B <= 0 || Tx <= 0 || Ty <= 0) return fail(VITS_ERR_ARG, "bad argument");
  vits_model* m = s->m;
  if (cap < (int64_t)Ty * m->hp.hop_length) return fail(VITS_ERR_ARG, "audio capacity %lld < T_y*hop", (long long)cap);
  HIP_TRY(hipSetDevice(m->device));
  (void)stream;  // sessions run on their own stream; the argument is reserved
  TRY(session_reserve(s, B, Tx, Ty));
  struct Mask { Mask(int v) { tl_persist = v; } ~Mask() { tl_persist = -1; } } mask(s->ps_owner ? persist_cfg() : 0);
  HIP_TRY(hipEventRecord(s->ev0, s->stream));
  if (s->use_graph && !s->profile) {
    vits_session::GKey key(d_ids, d_lengths, d_sid, d_forced, d_audio, B, Tx, Ty, seed, scales[0], scales[1], scales[2], persist_mask());
    auto it = s->graphs.find(key);
    if (it == s->graphs.end()) {
      if (s->graphs.size() >= 64) drop_graphs(s);  // bound the cache: a caller that varies shapes/pointers forever must not leak
      hipGraph_t g = nullptr;
      HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
      forward_device(s, d_ids, d_lengths, B, Tx, scales, d_sid, d_forced, Ty, seed, d_audio, cap);
      HIP_TRY(hipStreamEndCapture(s->stream, &g));
      { size_t nn = 0; if (hipGraphGetNodes(g, nullptr, &nn) == hipSuccess) s->graph_nodes = (int)nn; }
      hipGraphExec_t ge = nullptr;
      const hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      if (ie != hipSuccess) return fail(VITS_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
      it = s->graphs.emplace(key, ge).first;
    }
    HIP_TRY(hipGraphLaunch(it->second, s->stream));
  } else {
    forward_device(s, d_ids, d_lengths, B, Tx, scales, d_sid, d_forced, Ty, seed, d_audio, cap);
  }
  HIP_TRY(hipEventRecord(s->ev1, s->stream));
  s->timed = true;
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(VITS_ERR_DEVICE, "launch failed: %s", hipGetErrorString(le));
  return VITS_OK;
}

int vits_session_last_ms(vits_session* s, float* ms) {
  if (!s || !ms || !s->timed) return fail(VITS_ERR_ARG, "no timed call");
  HIP_TRY(hipEventSynchronize(s->ev1));
  HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
  return check_err(s);
}

void vits_debug_force_tile(int mode) { g_force_tile = mode; }
void vits_debug_attention_impl(int impl) { g_attn_impl = impl; }
void vits_debug_ks_waves(int nw) { g_ks_waves = nw; }
void vits_debug_tail_impl(int impl) { g_tail_impl = impl; }
void vits_debug_wn_fold(int on) { g_wn_fold = on; }
void vits_debug_ln_stats(int on) { g_ln_stats = on; }
void vits_debug_persist(int on) {  // (also arms the programs at once: a test that switches them on means now)
  g_persist = on;
  g_ps_off_until_ns.store(0); g_ps_rearm_ns.store(0); g_ps_rearmed_at_ns.store(0);
}
void vits_debug_persist_when(int max_others_in_flight) { g_persist_when = max_others_in_flight < -1 ? -1 : max_others_in_flight; }
void vits_debug_persist_rearm_ms(int ms) {
  g_ps_rearm_base_ns = (ms > 0 ? (long long)ms : (getenv("VITS_PERSIST_REARM_MS") ? atoll(getenv("VITS_PERSIST_REARM_MS")) : 1000)) * 1000000LL;
}
// State of the persistent programs of this process (a server logs it; tests assert the re-arm).
int vits_persist_state(vits_model* m, vits_persist_info* out) {
  if (!out) return fail(VITS_ERR_ARG, "null out");
  memset(out, 0, sizeof *out);
  out->configured_mask = g_persist;
  const long long until = g_ps_off_until_ns.load(), now = steady_ns();
  out->active_mask = (until && now < until) ? 0 : g_persist.load();
  out->off_for_ms = (until && now < until) ? (int32_t)((until - now + 999999) / 1000000) : 0;
  out->timeouts = g_ps_timeouts.load();
  out->rearms = g_ps_rearms.load();
  out->launches = -1;
  out->process_owns_device = -1;
  if (m) {
    out->launches = vits_debug_persist_runs(m);
    std::lock_guard<std::mutex> g(g_tok_mu);
    if (m->device >= 0 && m->device < 64) out->process_owns_device = g_proc_lock[m->device];
  }
  return VITS_OK;
}
// The limit lives in a device word the kernel reads at run time, so graphs captured before or after the call follow it alike.
void vits_debug_persist_spin(int limit) {
  const int lim = limit > 0 ? limit : 0;
  g_ps_spin_limit = lim;
  std::lock_guard<std::mutex> g(g_models_mu);
  for (vits_model* m : g_models) {
    hipSetDevice(m->device);
    hipDeviceSynchronize();
    hipMemcpy(m->ps_dbg, &lim, sizeof(int), hipMemcpyHostToDevice);
  }
}
// persistent launches of this model that ran to completion (no timeout) since it was created
int vits_debug_persist_runs(vits_model* m) {
  if (!m) return -1;
  int v[2] = {0, 0};
  hipSetDevice(m->device);
  hipDeviceSynchronize();
  if (hipMemcpy(v, m->ps_dbg, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return v[1];
}
void vits_debug_conv_wp(int mode) { g_wp_mode = mode; }
// Host arithmetic only (no device): the per-layer limits decoder_needs derives for a padded-batch continuation.  Layout: [0] frames of z
// read beyond an item's end (conv_pre's output limit + its 3 taps to the right), [1] pre_out, [2] post_out, [3] tail_cols, then per
// upsampling stage: ups_q, c1_out[0..n_resd), c2_out[0..n_resd).  Returns the number of values (written up to cap).
int vits_debug_decoder_needs(const vits_hparams* hp, int32_t* out, int32_t cap) {
  if (!hp || !out || hp->n_ups < 0 || hp->n_ups > VITS_MAX_UPS || hp->n_resd < 0 || hp->n_resd > VITS_MAX_RESD || hp->n_resk < 0 || hp->n_resk > VITS_MAX_RESK)
    return -fail(VITS_ERR_ARG, "decoder_needs: bad arguments");
  for (int i = 0; i < hp->n_ups; ++i) if (hp->up_rates[i] < 1) return -fail(VITS_ERR_ARG, "decoder_needs: bad up_rates");
  if (hp->dec_type == 0 && (hp->istft_hop < 1 || hp->subbands < 1)) return -fail(VITS_ERR_ARG, "decoder_needs: bad tail geometry");
  const DecNeeds N = decoder_needs(*hp, true);
  std::vector<int32_t> v = {N.pre_out + 3, N.pre_out, N.post_out, N.tail_cols};
  for (int i = 0; i < hp->n_ups; ++i) {
    v.push_back(N.ups_q[i]);
    for (int d = 0; d < hp->n_resd; ++d) v.push_back(N.c1_out[i][d]);
    for (int d = 0; d < hp->n_resd; ++d) v.push_back(N.c2_out[i][d]);
  }
  for (int i = 0; i < (int)v.size() && i < cap; ++i) out[i] = v[i];
  return (int)v.size();
}
void vits_debug_conv_sp(int mode) { g_sp_mode = mode; }
void vits_debug_conv_sk(int mode) { g_sk_mode = mode; }
int vits_debug_clock_probe(int device, int32_t duration_us, double* ghz, int32_t n) {
  if (!ghz || n < 1 || n > 1024 || duration_us < 1 || duration_us > 2000000 || device < 0 || device >= 64) return fail(VITS_ERR_ARG, "clock probe: bad arguments");
  HIP_TRY(hipSetDevice(device));
  // stream and result buffer are created ONCE per device and kept: hipMalloc / hipStreamCreate wait for the device's work in flight, and
  // the probe exists to run NEXT to that work (call it once before the burst it is to watch)
  static std::mutex mu;
  static hipStream_t st[64];
  static double* buf[64];
  std::lock_guard<std::mutex> g(mu);
  if (!st[device]) {
    HIP_TRY(hipStreamCreateWithFlags(&st[device], hipStreamNonBlocking));  // its own hardware queue
    if (hipMalloc((void**)&buf[device], sizeof(double) * 1024) != hipSuccess) { buf[device] = nullptr; return fail(VITS_ERR_NOMEM, "clock probe: device alloc failed"); }
  }
  if (!buf[device]) return fail(VITS_ERR_NOMEM, "clock probe: no buffer");
  hipLaunchKernelGGL(clock_probe_kernel, dim3(n), dim3(64), 0, st[device], buf[device], (long long)duration_us * 100);
  hipError_t e = hipStreamSynchronize(st[device]);
  if (e == hipSuccess) e = hipMemcpyAsync(ghz, buf[device], sizeof(double) * n, hipMemcpyDeviceToHost, st[device]);
  if (e == hipSuccess) e = hipStreamSynchronize(st[device]);
  if (e != hipSuccess) return fail(VITS_ERR_DEVICE, "clock probe failed: %s", hipGetErrorString(e));
  return n;
}
void vits_debug_no_bf16x3(int on) { g_no_bf3 = on; }
void vits_debug_poison_workspace(int on) { g_poison = on; }

int vits_session_sync(vits_session* s) {
  if (!s) return fail(VITS_ERR_ARG, "null session");
  return check_err(s);
}

int vits_session_set_options(vits_session* s, int use_graph, int profile) {
  if (!s) return fail(VITS_ERR_ARG, "null session");
  s->use_graph = use_graph != 0;
  s->profile = profile != 0;
  return VITS_OK;
}

int vits_session_graph_nodes(vits_session* s) { return s ? s->graph_nodes : 0; }

int vits_session_set_sdp_always(vits_session* s, int on) {
  if (!s) return fail(VITS_ERR_ARG, "null session");
  if (s->sdp_always != (on != 0)) drop_graphs(s);
  s->sdp_always = on != 0;
  return VITS_OK;
}

// Per-kernel-family device time from HIP events recorded around every launch of the last
// profiled (eager) forwards.  Writes lines "name launches total_ms flops" into buf.
int vits_session_profile_report(vits_session* s, char* buf, size_t cap) {
  if (!s || !buf || !cap) return fail(VITS_ERR_ARG, "bad argument");
  HIP_TRY(hipStreamSynchronize(s->stream));
  std::map<std::string, std::tuple<int, double, double>> agg;
  for (auto& r : s->prof) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    auto& a = agg[r.name + " " + r.kernel];
    std::get<0>(a) += 1; std::get<1>(a) += ms; std::get<2>(a) += r.flops;
    hipEventDestroy(r.e0); hipEventDestroy(r.e1);
  }
  s->prof.clear();
  size_t off = 0;
  buf[0] = 0;
  for (auto& kv : agg) {
    int n = snprintf(buf + off, cap - off, "%s %d %.6f %.0f\n", kv.first.c_str(), std::get<0>(kv.second), std::get<1>(kv.second), std::get<2>(kv.second));
    if (n < 0 || (size_t)n >= cap - off) break;
    off += n;
  }
  return VITS_OK;
}

// ---- single generic op (kernel-level parity): y = conv1d(lrelu(x)), 'same' padding
int vits_op_conv1d(int device, const float* x, const float* w, const float* bias, int32_t B, int32_t Cin, int32_t Cout, int32_t T,
                   int32_t K, int32_t dil, float slope, float* y) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || T <= 0 || K <= 0 || dil <= 0) return fail(VITS_ERR_ARG, "bad argument");
  if (Cin % CONV_CI_T) return fail(VITS_ERR_UNSUPPORTED, "C_in must be a multiple of %d", CONV_CI_T);
  if ((K - 1) * dil > CONV_MAX_HALO) return fail(VITS_ERR_UNSUPPORTED, "(K-1)*dil > %d", CONV_MAX_HALO);
  if ((long long)(Cin > Cout ? Cin : Cout) * T * 4 >= (1LL << 31)) return fail(VITS_ERR_ARG, "one item's tensor must stay below 2 GiB (32-bit offsets inside an item)");
  HIP_TRY(hipSetDevice(device));
  vits_model tmp;
  tmp.device = device;
  ConvW W = make_conv(&tmp, Cout, Cin, K, bias, [&](int r, int ci, int kk) { return w[((size_t)r * Cin + ci) * K + kk]; });
  int rc = VITS_OK;
  float *dx = nullptr, *dy = nullptr;
  vits_session s;
  s.m = &tmp;
  if (tmp.missing) rc = VITS_ERR_NOMEM;
  if (rc == VITS_OK && hipMalloc((void**)&dx, sizeof(float) * (size_t)B * Cin * T) != hipSuccess) rc = fail(VITS_ERR_NOMEM, "alloc");
  if (rc == VITS_OK && hipMalloc((void**)&dy, sizeof(float) * (size_t)B * Cout * T) != hipSuccess) rc = fail(VITS_ERR_NOMEM, "alloc");
  if (rc == VITS_OK) {
    hipMemcpy(dx, x, sizeof(float) * (size_t)B * Cin * T, hipMemcpyHostToDevice);
    ConvParams P = conv_params(W, dx, dy, B, T, dil, (K - 1) * dil / 2);
    P.in_slope = slope;
    const char* dbg_env = getenv("VITS_CONV_DBG");
    long long* d_dbg = nullptr;
    constexpr size_t dbg_n = 128 + 4 * 4000;  // phase stamps + block trace (timing build)
    if (dbg_env) { hipMalloc((void**)&d_dbg, dbg_n * sizeof(long long)); hipMemset(d_dbg, 0, dbg_n * sizeof(long long)); P.dbg = d_dbg; }
    const int reps = dbg_env ? atoi(dbg_env) : 1;
    hipEvent_t e0, e1, ea; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&ea);
    for (int r = 0; r < reps; ++r) {
      if (r == 1 || reps == 1) hipEventRecord(ea, 0);  // all launches after the first (steady state, operands cache-warm)
      if (r == reps - 1) hipEventRecord(e0, 0);
      launch_conv(&s, P, EPI_STORE, "op.conv1d");
    }
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) rc = fail(VITS_ERR_DEVICE, "conv kernel failed: %s", hipGetErrorString(e));
    else hipMemcpy(y, dy, sizeof(float) * (size_t)B * Cout * T, hipMemcpyDeviceToHost);
    if (dbg_env && rc == VITS_OK) {
      long long h[128]; float ms = 0, msa = 0;
      hipMemcpy(h, d_dbg, sizeof h, hipMemcpyDeviceToHost);
      hipEventElapsedTime(&ms, e0, e1);
      hipEventElapsedTime(&msa, ea, e1);
      fprintf(stderr, "[conv dbg] B=%d Cin=%d Cout=%d T=%d K=%d dil=%d: last launch %.2f us (event), %.2f us/launch over the last %d back-to-back launches = %.1f TFLOP/s; block 0 cycles since kernel start:\n",
              B, Cin, Cout, T, K, dil, ms * 1e3, msa * 1e3 / (reps > 1 ? reps - 1 : 1), reps > 1 ? reps - 1 : 1,
              2.0 * B * Cin * Cout * (double)T * K / (msa * 1e-3 / (reps > 1 ? reps - 1 : 1)) / 1e12);
      const long blocks64 = (long)cdiv(W.Mpad, 64) * cdiv(T, 64) * B;
      {
        int nb = -1, nb2 = -1, nb3 = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_mfma_kernel<2, 2, 2, 2, EPI_STORE>, 256, 2 * CONV_CI_T * (128 + 64) * 4);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, conv_mfma_kernel<2, 2, 1, 1, EPI_STORE>, 256, 2 * CONV_CI_T * (64 + 64) * 4);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb3, conv_mfma_ks_kernel<1, 1, EPI_STORE, 1, 4>, 256, 16384);
        hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)conv_mfma_kernel<2, 2, 2, 2, EPI_STORE>);
        fprintf(stderr, "   occupancy API (blocks/CU): T128 %d  T64 %d  ks %d ; T128 numRegs %d sharedStatic %zu localMem %zu maxDynShared %d\n", nb, nb2, nb3, fa.numRegs,
                fa.sharedSizeBytes, fa.localSizeBytes, fa.maxDynamicSharedSizeBytes);
      }
      if (blocks64 >= 512) {
        for (int w = 0; w < 4; ++w)
          fprintf(stderr, "   [big-tile] wave %d: prologue %lld  taps %lld  store+barrier %lld  mainloop_end %lld  end %lld  (MFMA floor %lld)\n", w, h[w * 8], h[w * 8 + 1],
                  h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], (long long)(Cin / 2) * K * 4 * 64);
      } else
      for (int w = 0; w < 16; ++w)  // stamps relative to wave 0's start; HW_ID: simd = bits 5:4, cu = bits 11:8, se = bits 15:13
        if (h[w * 8]) fprintf(stderr, "   wave %2d simd %lld cu %lld: start %+lld | +%lld  +%lld  +%lld  +%lld  +%lld  +%lld\n", w, (h[w * 8 + 7] >> 4) & 3, (h[w * 8 + 7] >> 8) & 15,
                h[w * 8] - h[0], h[w * 8 + 1] - h[w * 8], h[w * 8 + 2] - h[w * 8], h[w * 8 + 3] - h[w * 8], h[w * 8 + 4] - h[w * 8], h[w * 8 + 5] - h[w * 8], h[w * 8 + 6] - h[w * 8]);
    }
    if (d_dbg && rc == VITS_OK && getenv("VITS_CONV_BT")) {  // block trace of the LAST launch (timing build): "blk id start end hw xcc", 10 ns units
      std::vector<long long> t(4 * 4000);
      hipMemcpy(t.data(), d_dbg + 128, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
      long long t0 = 0;
      for (int i = 0; i < 4000; ++i) if (t[4 * i] && (!t0 || t[4 * i] < t0)) t0 = t[4 * i];
      for (int i = 0; i < 4000; ++i)
        if (t[4 * i]) fprintf(stderr, "blk %d %lld %lld %lld %lld\n", i, t[4 * i] - t0, t[4 * i + 1] ? t[4 * i + 1] - t0 : -1, t[4 * i + 2], t[4 * i + 3]);
    }
    if (d_dbg) hipFree(d_dbg);
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(ea);
  }
  if (dx) hipFree(dx);
  if (dy) hipFree(dy);
  for (void* a : tmp.allocs) hipFree(a);
  return rc;
}

}  // extern "C"

#include "stts.hip.h"
