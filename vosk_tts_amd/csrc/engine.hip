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
#include "conv_mfma.hip.h"
#include "conv_small.hip.h"
#include "conv_sp.hip.h"
#include "conv_w1.hip.h"
#include "conv_bf3.hip.h"
#include "kernels_misc.hip.h"
#include "persist.hip.h"

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

// ---- test hooks (process-wide) -------------------------------------------------------------------
// 0 = size heuristic, 1 = force the big-tile kernel, 2 = force the K-split kernel (tests only)
static int g_force_tile = 0;
// 0 = fp32-MFMA flash attention (default), 1 = the VALU kernel (kept as an independent cross-check in tests)
static int g_attn_impl = 0;
// tests: fill every freshly planned workspace with NaN so stale padding can never hide as zeros
static int g_poison = 0;
// 0 = fused exp/sin + iSTFT + PQMF kernel (default), 1 = the two separate kernels (independent cross-check in tests)
static int g_tail_impl = 0;
// 1 = folded WN tail (stacked gate outputs, one skip+post conv; default), 0 = per-layer res/skip epilogue + post
static int g_wn_fold = 1;
// 1 = LayerNorm statistics of the folded encoder LayerNorms from the producer conv's epilogue (default), 0 = redone by the consumer
static int g_ln_stats = 1;
static int g_ps_spin_limit = 0;  // test hook (vits_debug_persist_spin): poll rounds before a persistent worker gives up; 0 = PS_SPIN_LIMIT
// single-utterance duration predictor as one persistent kernel (persist.hip.h): 1 = when eligible (default), 0 = launch path
// A persistent kernel needs ALL its workgroups resident at once (they spin on each other's cells): two of them in flight on one
// device could each hold half of the CUs and wait forever (the bounded poll loops turn that into an error, not a hang -- but it must
// not happen in normal operation).  So at most ONE caller per device owns the persistent path at a time (a token); everybody else
// takes the launch path for that call.  persist_mask() is what the stage launchers test: the owner's mask, 0 for everybody else.
static std::mutex g_tok_mu;
static bool g_tok_busy[64];
static thread_local int tl_persist = -1;  // >= 0: this thread's mask for the call in progress
static int g_persist = getenv("VITS_NO_PERSIST") ? 0 : (getenv("VITS_PERSIST") ? atoi(getenv("VITS_PERSIST")) : 7);  // mask: 1 duration predictor, 2 text encoder, 4 flow (environment switches: A/B runs of bench.py and tools/)
// A poll timeout (persist_timed_out) switches the programs off for a BOUNDED interval, not for the life of the process: a server that
// once lost co-residency (another process on the device, a transient) gets them back.  The interval starts at VITS_PERSIST_REARM_MS
// (default 1000) and doubles with every timeout that follows a re-arm within 10 intervals (cap: 64 x), so a device that is shared for
// good costs one failed forward per minute, not one per second.  persist_cfg() is the mask in effect now; vits_persist_state reports.
static long long g_ps_rearm_base_ns = (getenv("VITS_PERSIST_REARM_MS") ? atoll(getenv("VITS_PERSIST_REARM_MS")) : 1000) * 1000000LL;
static std::atomic<long long> g_ps_off_until_ns{0};  // steady-clock ns; 0 = armed
static std::atomic<long long> g_ps_rearmed_at_ns{0};
static std::atomic<long long> g_ps_rearm_ns{0};      // current interval (0 = base)
static std::atomic<int> g_ps_timeouts{0}, g_ps_rearms{0};
static inline long long steady_ns() {
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static int persist_cfg() {
  if (!g_persist) return 0;
  long long until = g_ps_off_until_ns.load(std::memory_order_relaxed);
  if (!until) return g_persist;
  const long long now = steady_ns();
  if (now < until) return 0;
  if (g_ps_off_until_ns.compare_exchange_strong(until, 0)) { g_ps_rearms.fetch_add(1); g_ps_rearmed_at_ns.store(now); }
  return g_persist;
}

// ------------------------------------------------------------------------------------ weights
struct ConvW {
  float* w = nullptr;     // packed, MFMA fragment order
  float* w16 = nullptr;   // packed for the small-tile kernel (16x16x4 fragment order), null when not built
  float* wb = nullptr;    // split-bf16 (hi, lo) pieces in 32x32x16 fragment order (conv_bf3.hip.h), null when not built
  float* bias = nullptr;  // original row order
  int M = 0, Mpad = 0, Cin = 0, K = 0, n_sg = 0;
};
struct EncLayerW {
  ConvW qkv, o, f1, f2;
  float *ek = nullptr, *ev = nullptr, *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};
struct EncoderW {
  std::vector<EncLayerW> layers;
  int H = 0, F = 0, K = 0;
};
struct DDSW {
  std::vector<float*> sw, sb, g1, b1, g2, b2;
  std::vector<float*> swk[3];  // the depthwise taps as three per-channel vectors (persistent column steps: thread = channel)
  std::vector<float*> wt;  // 1x1 weights transposed [ci][co] for the fused layer kernel
  std::vector<ConvW> pw;
};
struct ConvFlowW {
  float *pre_w = nullptr, *pre_b = nullptr;
  DDSW dds;
  ConvW proj;
};
struct CouplingW {
  ConvW pre, post;
  EncoderW enc;
  std::vector<ConvW> in_layers, rs_layers;
  // Folded form of the WN tail (modules.py:168-176 + models.py:379-380): the skip halves of all res_skip layers and the
  // coupling's `post` conv are linear maps with nothing between them, so
  //   post((sum_i W_skip_i acts_i + b_skip_i) * mask) = [W_post W_skip_0 | ... | W_post W_skip_{L-1}] [acts_0; ...; acts_{L-1}] + b'
  // on every valid column: ONE [I/2 x L*H] 1x1 conv over the stacked gate outputs replaces L skip accumulations and `post`;
  // rsx[i] (i < L-1) keeps only the residual half of res_skip layer i.
  std::vector<ConvW> rsx;
  ConvW skip_post;
  int cond_off = 0;
};
struct ResBlockW {
  ConvW c1[VITS_MAX_RESD], c2[VITS_MAX_RESD];
  int K = 0;
  int dil[VITS_MAX_RESD] = {0};
};
struct UpW {
  ConvW w;  // polyphase-packed: row = phase*cout + co, taps = Ku/u
  int u = 0, Ku = 0, cout = 0, taps = 0, pad_l = 0, halo = 0;
  int shift[8] = {0};
};

struct vits_session;

struct vits_model {
  bool acoustic = true;  // false: vocoder-only blob (n_vocab == 0)
  vits_hparams hp;
  int device = 0;
  std::vector<void*> allocs;
  char* slab = nullptr;  // current weight slab (weight_alloc)
  size_t slab_bytes = 0, slab_used = 0;
  const unsigned char* blob = nullptr;  // only during create
  size_t blob_bytes = 0;
  uint32_t n_entries = 0;
  const vits_blob_entry* entries = nullptr;
  bool missing = false;

  float *emb = nullptr, *emb_g = nullptr;
  float *cond_W = nullptr, *cond_b = nullptr;  // all cond(g)/Linear(g) matrices row-concatenated
  int cond_rows = 0, cond_enc_off = -1, cond_dp_off = -1, cond_dec_off = -1;
  EncoderW enc_p;
  ConvW enc_proj;
  ConvW bert_proj;  // BERT-conditioned flavour (hparams.bert_dim > 0): 1x1 projection of the "bert" feed onto the embedding
  ConvW dp_pre, dp_proj;
  DDSW dp_dds;
  std::vector<ConvFlowW> cf;  // index k -> dp.flows.(2k+1), k = 1..n-1 (k = 0 unused)
  float *ea_m = nullptr, *ea_logs = nullptr;
  float ea_m_h[2] = {0, 0}, ea_logs_h[2] = {0, 0};
  std::vector<CouplingW> flow;
  ConvW conv_pre, conv_post;
  std::vector<UpW> ups;
  std::vector<ResBlockW> rb;
  float *istft_basis = nullptr, *pqmf = nullptr;
  bool use_g = false;
  float* zeros = nullptr;  // 4096 zeros: the "unused" parameter pointers of persistent-kernel steps (persist.hip.h)
  int* ps_dbg = nullptr;   // device words read / written by persist_kernel: [0] poll-round limit (0 = default), [1] completed persistent launches
  std::mutex pack_mu;      // packed per-thread parameter vectors of persistent steps, keyed by their sources (persist_plan.hip.h)
  std::map<std::vector<long long>, const float*> packs;
  int n_cu = 0;       // compute units of the device = workgroups of a persistent kernel (persist.hip.h)
  int rag_halo = 32;  // frames decoded beyond an item's end in ragged batches / streaming windows: >= the decoder's receptive field

  std::mutex pool_mu;
  std::vector<vits_session*> pool;
  // fast path: idle front sessions by (B, T_x bucket), least-recently-used eviction under a device-memory cap
  std::multimap<std::pair<int, int>, vits_session*> fronts;
  uint64_t use_clock = 0;
  size_t fronts_bytes = 0;
};

// live models (vits_debug_persist_spin writes the poll limit into each model's device word)
static std::mutex g_models_mu;
static std::vector<vits_model*> g_models;

static const float* tget(vits_model* m, int ndim, int d0, int d1, int d2, const char* fmt, ...) {
  char name[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(name, sizeof name, fmt, ap);
  va_end(ap);
  for (uint32_t i = 0; i < m->n_entries; ++i) {
    const vits_blob_entry* e = &m->entries[i];
    if (strncmp(e->name, name, sizeof e->name) == 0) {
      const int want[3] = {d0, d1, d2};
      if ((int)e->ndim != ndim) { m->missing = true; fail(VITS_ERR_BLOB, "tensor %s: ndim %u != %d", name, e->ndim, ndim); return nullptr; }
      for (int k = 0; k < ndim && k < 3; ++k)
        if (want[k] >= 0 && (int)e->dims[k] != want[k]) {
          m->missing = true;
          fail(VITS_ERR_BLOB, "tensor %s: dim %d is %u, expected %d", name, k, e->dims[k], want[k]);
          return nullptr;
        }
      return reinterpret_cast<const float*>(m->blob + e->offset);
    }
  }
  m->missing = true;
  fail(VITS_ERR_BLOB, "tensor %s missing from blob", name);
  return nullptr;
}

static bool thas(const vits_model* m, const char* name) {  // optional tensors
  for (uint32_t i = 0; i < m->n_entries; ++i)
    if (strncmp(m->entries[i].name, name, sizeof m->entries[i].name) == 0) return true;
  return false;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Weights live in a few large slabs (bump allocation, 256-byte aligned) instead of one hipMalloc per tensor: 2 MB-fragment
// mappings, a layer's tensors adjacent, ~470 fewer allocations per model.  (Measured: no effect on the forward's time, so
// address translation of the small per-tensor allocations was not what bounds the short-utterance kernels.)
static void* weight_alloc(vits_model* m, size_t bytes) {
  static const bool no_slab = getenv("VITS_NO_SLAB") != nullptr;  // A/B switch for tools/
  bytes = align_up(bytes ? bytes : 4, 256);
  // every allocation is followed by >= 64 KB of mapped memory: the weight streams are prefetched past their end by design (one or
  // two steps in the conv kernels, whose packings are padded for it); the slack makes an overrun of any of them a read of mapped
  // memory instead of a fault
  constexpr size_t guard = (size_t)64 << 10;
  if (no_slab) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes + guard) != hipSuccess) return nullptr;
    m->allocs.push_back(d);
    return d;
  }
  if (m->slab_used + bytes + guard > m->slab_bytes) {
    size_t want = m->blob_bytes + m->blob_bytes / 4 + ((size_t)8 << 20);  // first slab: the whole model with packing slack
    if (!m->allocs.empty()) want = (size_t)32 << 20;
    if (want < bytes + guard) want = bytes + guard;
    want = align_up(want, (size_t)2 << 20);
    void* d = nullptr;
    if (hipMalloc(&d, want) != hipSuccess) return nullptr;
    m->allocs.push_back(d);
    m->slab = static_cast<char*>(d);
    m->slab_bytes = want;
    m->slab_used = 0;
  }
  void* p = m->slab + m->slab_used;
  m->slab_used += bytes;
  return p;
}

static float* upload(vits_model* m, const float* host, size_t n) {
  if (!host) return nullptr;
  void* d = weight_alloc(m, n * sizeof(float));
  if (!d) { m->missing = true; fail(VITS_ERR_NOMEM, "hipMalloc of %zu floats failed", n); return nullptr; }
  if (hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { m->missing = true; fail(VITS_ERR_DEVICE, "hipMemcpy H2D failed"); return nullptr; }
  return static_cast<float*>(d);
}

// Generic packer: rows x Cin x K from a row-source functor (row may be remapped / zero padded).
// small16: also pack for conv16_kernel (encoder / duration predictor / flow convs; the decoder never runs in the few-column
// regime with a single group, so its convs skip the second copy).  src16: row source of that packing when its row order
// differs (WN gate: [8 tanh | 8 sigmoid] per 16-row block instead of [32 | 32]).
template <typename F, typename F16>
static ConvW make_conv2(vits_model* m, int M, int Cin, int K, const float* bias, F src, bool small16, F16 src16) {
  ConvW c;
  c.M = M; c.Mpad = cdiv(M, 32) * 32; c.Cin = Cin; c.K = K;
  if (Cin % CONV_CI_T != 0) { m->missing = true; fail(VITS_ERR_UNSUPPORTED, "conv C_in=%d is not a multiple of %d", Cin, CONV_CI_T); return c; }
  c.n_sg = Cin / CONV_CI_T * 2 * K;
  std::vector<float> packed((size_t)c.Mpad * Cin * K);
  pack_conv_weights(packed.data(), c.Mpad, Cin, K, [&](int row, int ci, int kk) -> float { return row < M ? src(row, ci, kk) : 0.f; });
  c.w = upload(m, packed.data(), packed.size());
  if (small16) {
    const int Mp16 = cdiv(M, 16) * 16;
    packed.assign((size_t)Mp16 * Cin * K, 0.f);
    pack_conv_weights16(packed.data(), Mp16, Cin, K, [&](int row, int ci, int kk) -> float { return row < M ? src16(row, ci, kk) : 0.f; });
    c.w16 = upload(m, packed.data(), packed.size());
  }
  c.bias = bias ? upload(m, bias, M) : nullptr;
  return c;
}
template <typename F>
static ConvW make_conv(vits_model* m, int M, int Cin, int K, const float* bias, F src, bool small16 = true) {
  return make_conv2(m, M, Cin, K, bias, src, small16, src);
}

// nn.Conv1d weight [Cout, Cin, K] (+ optional bias)
// third packing of a conv: bf16 (hi, lo) pieces for conv_bf3_kernel (hparams.conv_precision == 1, decoder ResBlock convs)
template <typename F>
static void add_bf3_packing(vits_model* m, ConvW& c, F src) {
  if (!c.w || c.Mpad % 32 || c.Cin % CONV_CI_T) return;
  std::vector<uint16_t> pk((size_t)c.Mpad * c.Cin * c.K * 2 + 2 * 128 * 8, 0);  // + two steps: the kernel prefetches up to two steps past the end
  const int M = c.M;
  pack_conv_weights_bf3(pk.data(), c.Mpad, c.Cin, c.K, [&](int row, int ci, int kk) -> float { return row < M ? src(row, ci, kk) : 0.f; });
  c.wb = upload(m, reinterpret_cast<const float*>(pk.data()), pk.size() / 2);
}
static ConvW conv_from(vits_model* m, const char* name, int Cout, int Cin, int K, bool has_bias, bool small16 = true, bool bf3 = false) {
  const float* w = tget(m, 3, Cout, Cin, K, "%s.weight", name);
  const float* b = has_bias ? tget(m, 1, Cout, -1, -1, "%s.bias", name) : nullptr;
  if (m->missing) return ConvW();
  auto src = [&](int r, int ci, int kk) { return w[((size_t)r * Cin + ci) * K + kk]; };
  ConvW c = make_conv(m, Cout, Cin, K, b, src, small16);
  if (bf3) add_bf3_packing(m, c, src);
  return c;
}

static void load_encoder(vits_model* m, EncoderW& E, const char* pfx, int n_layers, int H, int F, int K) {
  const vits_hparams& hp = m->hp;
  const int dk = H / hp.n_heads, NW = 2 * hp.window_size + 1;
  E.H = H; E.F = F; E.K = K;
  E.layers.resize(n_layers);
  char nm[200];
  for (int i = 0; i < n_layers && !m->missing; ++i) {
    EncLayerW& L = E.layers[i];
    // q,k,v 1x1 convs fused into one M = 3H GEMM (attentions.py:156-158)
    const float* wq = tget(m, 3, H, H, 1, "%s.attn_layers.%d.conv_q.weight", pfx, i);
    const float* wk = tget(m, 3, H, H, 1, "%s.attn_layers.%d.conv_k.weight", pfx, i);
    const float* wv = tget(m, 3, H, H, 1, "%s.attn_layers.%d.conv_v.weight", pfx, i);
    const float* bq = tget(m, 1, H, -1, -1, "%s.attn_layers.%d.conv_q.bias", pfx, i);
    const float* bk = tget(m, 1, H, -1, -1, "%s.attn_layers.%d.conv_k.bias", pfx, i);
    const float* bv = tget(m, 1, H, -1, -1, "%s.attn_layers.%d.conv_v.bias", pfx, i);
    if (m->missing) return;
    std::vector<float> b3(3 * H);
    memcpy(b3.data(), bq, sizeof(float) * H); memcpy(b3.data() + H, bk, sizeof(float) * H); memcpy(b3.data() + 2 * H, bv, sizeof(float) * H);
    auto qkv_src = [&](int r, int ci, int) {
      const float* w = r < H ? wq : (r < 2 * H ? wk : wv);
      return w[(size_t)(r % H) * H + ci];
    };
    L.qkv = make_conv(m, 3 * H, H, 1, b3.data(), qkv_src);
    const bool bf3 = hp.conv_precision == 1 && H % 64 == 0 && F % 64 == 0;  // batch-size STORE convs of the encoders as split-bf16 too
    if (bf3) add_bf3_packing(m, L.qkv, qkv_src);
    snprintf(nm, sizeof nm, "%s.attn_layers.%d.conv_o", pfx, i);
    L.o = conv_from(m, nm, H, H, 1, true, true, bf3);
    L.ek = upload(m, tget(m, 3, 1, NW, dk, "%s.attn_layers.%d.emb_rel_k", pfx, i), (size_t)NW * dk);
    L.ev = upload(m, tget(m, 3, 1, NW, dk, "%s.attn_layers.%d.emb_rel_v", pfx, i), (size_t)NW * dk);
    snprintf(nm, sizeof nm, "%s.ffn_layers.%d.conv_1", pfx, i);
    L.f1 = conv_from(m, nm, F, H, K, true, true, bf3);
    snprintf(nm, sizeof nm, "%s.ffn_layers.%d.conv_2", pfx, i);
    L.f2 = conv_from(m, nm, H, F, K, true, true, bf3);
    L.g1 = upload(m, tget(m, 1, H, -1, -1, "%s.norm_layers_1.%d.gamma", pfx, i), H);
    L.b1 = upload(m, tget(m, 1, H, -1, -1, "%s.norm_layers_1.%d.beta", pfx, i), H);
    L.g2 = upload(m, tget(m, 1, H, -1, -1, "%s.norm_layers_2.%d.gamma", pfx, i), H);
    L.b2 = upload(m, tget(m, 1, H, -1, -1, "%s.norm_layers_2.%d.beta", pfx, i), H);
  }
}

static void load_dds(vits_model* m, DDSW& D, const char* pfx, int C, int K, int n) {
  char nm[200];
  for (int i = 0; i < n && !m->missing; ++i) {
    D.sw.push_back(upload(m, tget(m, 3, C, 1, K, "%s.convs_sep.%d.weight", pfx, i), (size_t)C * K));
    if (K == 3 && !m->missing) {
      const float* w = tget(m, 3, C, 1, K, "%s.convs_sep.%d.weight", pfx, i);
      for (int k = 0; k < 3; ++k) {
        std::vector<float> t(C);
        for (int c = 0; c < C; ++c) t[c] = w[(size_t)c * 3 + k];
        D.swk[k].push_back(upload(m, t.data(), t.size()));
      }
    }
    D.sb.push_back(upload(m, tget(m, 1, C, -1, -1, "%s.convs_sep.%d.bias", pfx, i), C));
    snprintf(nm, sizeof nm, "%s.convs_1x1.%d", pfx, i);
    D.pw.push_back(conv_from(m, nm, C, C, 1, true));
    {
      const float* w = tget(m, 3, C, C, 1, "%s.weight", nm);
      std::vector<float> t((size_t)C * C);
      // wt4[ci/4][co][ci%4] (dds_layer_kernel: one dwordx4 per thread per 4 input channels)
      if (w && C % 4 == 0) for (int co = 0; co < C; ++co) for (int ci = 0; ci < C; ++ci) t[((size_t)(ci / 4) * C + co) * 4 + (ci & 3)] = w[(size_t)co * C + ci];
      D.wt.push_back(upload(m, t.data(), t.size()));
    }
    D.g1.push_back(upload(m, tget(m, 1, C, -1, -1, "%s.norms_1.%d.gamma", pfx, i), C));
    D.b1.push_back(upload(m, tget(m, 1, C, -1, -1, "%s.norms_1.%d.beta", pfx, i), C));
    D.g2.push_back(upload(m, tget(m, 1, C, -1, -1, "%s.norms_2.%d.gamma", pfx, i), C));
    D.b2.push_back(upload(m, tget(m, 1, C, -1, -1, "%s.norms_2.%d.beta", pfx, i), C));
  }
}

static double bessel_i0(double x) {
  double s = 1.0, term = 1.0, q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) { term *= q / ((double)k * k); s += term; if (term < 1e-18 * s) break; }
  return s;
}

// ---- decoder weights (Multiband_iSTFT_Generator models.py:975-1054 / Generator :845-898)
static int load_decoder(vits_model* m) {
  const vits_hparams& hp = m->hp;
  const int I = hp.inter_channels;
  char nm[200];
  int C = hp.dec_initial_channel;
  m->conv_pre = conv_from(m, "dec.conv_pre", C, I, 7, true, false, hp.conv_precision == 1 && C % 128 == 0);
  // Geometry checks before anything divides by a rate or sizes a buffer from hop_length: the decoder writes
  // T_y * prod(up_rates) [* istft_hop * subbands] samples per item while every output buffer is T_y * hop_length.
  {
    long long rate = 1;
    for (int i = 0; i < hp.n_ups; ++i) {
      const int u = hp.up_rates[i], Ku = hp.up_kernels[i];
      if (u <= 0 || Ku < u) return fail(VITS_ERR_BLOB, "decoder stage %d: upsample rate %d / kernel %d invalid", i, u, Ku);
      rate *= u;
    }
    if (hp.dec_type == 0) {
      if (hp.subbands <= 0 || hp.istft_hop <= 0 || hp.istft_n_fft <= 0 || hp.istft_n_fft % hp.istft_hop || hp.pqmf_taps <= 0)
        return fail(VITS_ERR_BLOB, "iSTFT / PQMF parameters invalid (subbands %d, n_fft %d, hop %d, taps %d)", hp.subbands, hp.istft_n_fft, hp.istft_hop, hp.pqmf_taps);
      rate *= (long long)hp.istft_hop * hp.subbands;
    }
    if (hp.hop_length <= 0 || rate != hp.hop_length)
      return fail(VITS_ERR_BLOB, "decoder produces %lld samples per frame but hop_length is %d", rate, hp.hop_length);
    for (int j = 0; j < hp.n_resk; ++j)
      if (hp.res_kernels[j] <= 0 || hp.res_kernels[j] % 2 == 0) return fail(VITS_ERR_BLOB, "resblock kernel %d invalid", hp.res_kernels[j]);
  }
  {
    // One-sided receptive field of the decoder in frames (24.9 for the default config): ragged batches and
    // streaming windows reproduce the dense result only if the halo they keep is at least this wide.
    double rf = 3.0, rate = 1.0;  // conv_pre k = 7
    for (int i = 0; i < hp.n_ups; ++i) {
      const int u = hp.up_rates[i], Ku = hp.up_kernels[i];
      rf += (double)((Ku + u - 1) / u / 2 + 1) / rate;
      rate *= u;
      double worst = 0;
      for (int j = 0; j < hp.n_resk; ++j) {
        double span = 0;
        for (int d = 0; d < hp.n_resd; ++d) span += (hp.res_kernels[j] - 1) * hp.res_dilations[j][d] / 2.0 + (hp.res_kernels[j] - 1) / 2.0;
        if (span > worst) worst = span;
      }
      rf += worst / rate;
    }
    rf += 4.0 / rate;  // conv_post k = 7 (+ reflection pad)
    if (hp.dec_type == 0) rf += ((double)hp.istft_n_fft / hp.istft_hop + (hp.pqmf_taps / 2.0) / hp.subbands / hp.istft_hop) / rate;
    m->rag_halo = (int)ceil(rf) + 2;
    if (m->rag_halo < 32) m->rag_halo = 32;
    if (m->rag_halo > 4096) return fail(VITS_ERR_UNSUPPORTED, "decoder receptive field of %d frames is not supported", m->rag_halo);
  }
  m->ups.resize(hp.n_ups);
  m->rb.resize((size_t)hp.n_ups * hp.n_resk);
  for (int i = 0; i < hp.n_ups && !m->missing; ++i) {
    UpW& U = m->ups[i];
    const int u = hp.up_rates[i], Ku = hp.up_kernels[i], Co = C / 2, p = (Ku - u) / 2;
    if (u > 8 || Ku % u || (Ku - u) % 2 || C % 64) return fail(VITS_ERR_UNSUPPORTED, "upsample rate/kernel unsupported");
    const float* w = tget(m, 3, C, Co, Ku, "dec.ups.%d.weight", i);  // [Cin, Cout, K]
    const float* b = tget(m, 1, Co, -1, -1, "dec.ups.%d.bias", i);
    if (m->missing) break;
    U.u = u; U.Ku = Ku; U.cout = Co; U.taps = Ku / u;
    // out[u*q + r] = sum_{delta} x[q + delta] * W[.., r + p - u*delta]; delta in [dmin(r), dmin(r)+taps-1]
    int dmin[8], dmin_all = 1 << 30, dmax_all = -(1 << 30);
    for (int r = 0; r < u; ++r) {
      const int dmax = (r + p) / u;  // floor, r+p >= 0
      dmin[r] = dmax - U.taps + 1;
      if (dmin[r] < dmin_all) dmin_all = dmin[r];
      if (dmax > dmax_all) dmax_all = dmax;
    }
    U.pad_l = -dmin_all;
    U.halo = dmax_all - dmin_all;
    for (int r = 0; r < u; ++r) U.shift[r] = dmin[r] + U.pad_l;
    const int Cin = C;
    auto ups_src = [&](int row, int ci, int j) {
      const int r = row / Co, co = row % Co;
      const int k = r + p - u * (dmin[r] + j);
      return (k >= 0 && k < Ku) ? w[((size_t)ci * Co + co) * Ku + k] : 0.f;
    };
    U.w = make_conv(m, u * Co, Cin, U.taps, nullptr, ups_src, false);
    if (hp.conv_precision == 1 && Co % 128 == 0) add_bf3_packing(m, U.w, ups_src);  // (used when the input is a single tensor)
    U.w.bias = upload(m, b, Co);
    C = Co;
    for (int j = 0; j < hp.n_resk && !m->missing; ++j) {
      ResBlockW& R = m->rb[(size_t)i * hp.n_resk + j];
      R.K = hp.res_kernels[j];
      for (int d = 0; d < hp.n_resd; ++d) {
        R.dil[d] = hp.res_dilations[j][d];
        if ((R.K - 1) * R.dil[d] > CONV_MAX_HALO) return fail(VITS_ERR_UNSUPPORTED, "resblock receptive field too wide");
        snprintf(nm, sizeof nm, "dec.resblocks.%d.convs1.%d", i * hp.n_resk + j, d);
        const bool bf3 = hp.conv_precision == 1 && C % 64 == 0;  // split-bf16 variant of the batch-size kernel (128- or 64-row tiles)
        R.c1[d] = conv_from(m, nm, C, C, R.K, true, false, bf3);
        snprintf(nm, sizeof nm, "dec.resblocks.%d.convs2.%d", i * hp.n_resk + j, d);
        R.c2[d] = conv_from(m, nm, C, C, R.K, true, false, bf3);
      }
    }
  }
  if (m->missing) return VITS_ERR_BLOB;
  if (hp.dec_type == 0) {
    const int S = hp.subbands, N = hp.istft_n_fft, hop = hp.istft_hop, cut = N / 2 + 1;
    m->conv_post = conv_from(m, "dec.subband_conv_post", S * (N + 2), C, 7, false, false);
    // OnnxSTFT inverse basis (stft.py:191-214): pinv(scale*[Re F;Im F]).T * hann == irfft synthesis rows / scale
    std::vector<float> basis((size_t)2 * cut * N);
    const double PI_D = 3.14159265358979323846, scale = (double)N / hop;
    for (int n = 0; n < N; ++n) {
      const double win = 0.5 - 0.5 * cos(2.0 * PI_D * n / N);
      for (int k = 0; k < cut; ++k) {
        const double wk = (k == 0 || k == N / 2) ? 1.0 : 2.0, th = 2.0 * PI_D * k * n / N;
        basis[(size_t)k * N + n] = (float)(wk * cos(th) / N / scale) * (float)win;
        basis[(size_t)(cut + k) * N + n] = (float)(-wk * sin(th) / N / scale) * (float)win;
      }
    }
    m->istft_basis = upload(m, basis.data(), basis.size());
    // PQMF synthesis filter (pqmf.py:15-43,64-75)
    const int taps = hp.pqmf_taps, Lf = taps + 1;
    std::vector<double> hpz(Lf);
    for (int n = 0; n < Lf; ++n) {
      const double xx = n - 0.5 * taps;
      const double hi = (n == taps / 2) ? (double)hp.pqmf_cutoff : sin(PI_D * hp.pqmf_cutoff * xx) / (PI_D * xx);
      const double r = (n - (Lf - 1) / 2.0) / ((Lf - 1) / 2.0), arg = 1.0 - r * r;
      hpz[n] = hi * bessel_i0(hp.pqmf_beta * sqrt(arg < 0 ? 0 : arg)) / bessel_i0(hp.pqmf_beta);
    }
    std::vector<float> filt((size_t)S * Lf);
    for (int k = 0; k < S; ++k)
      for (int n = 0; n < Lf; ++n)
        filt[(size_t)k * Lf + n] = (float)(2.0 * hpz[n] * cos((2 * k + 1) * (PI_D / (2.0 * S)) * (n - ((taps - 1) / 2.0)) - ((k % 2 == 0) ? 1.0 : -1.0) * PI_D / 4.0));
    m->pqmf = upload(m, filt.data(), filt.size());
  } else {
    // VITS' Generator has no conv_post bias (models.py:866); the HiFi-GAN bundled with StableTTS has one
    m->conv_post = conv_from(m, "dec.conv_post", 1, C, 7, thas(m, "dec.conv_post.bias"), false);
  }
  return m->missing ? VITS_ERR_BLOB : VITS_OK;
}

static int load_model(vits_model* m) {
  const vits_hparams& hp = m->hp;
  const int H = hp.hidden_channels, I = hp.inter_channels, F = hp.filter_channels, G = hp.gin_channels;
  const int D = hp.dp_filter_channels;
  // n_vocab == 0: a vocoder-only blob (e.g. the HiFi-GAN bundled with StableTTS, matcha/hifigan/models.py:148-199):
  // only the decoder tensors exist and only vits_stage_decoder / vits_stream-less decoding is available
  m->acoustic = hp.n_vocab > 0;
  if (!m->acoustic) {
    if (I % CONV_CI_T) return fail(VITS_ERR_UNSUPPORTED, "decoder input channels must be a multiple of %d", CONV_CI_T);
    if (hp.n_ups > VITS_MAX_UPS || hp.n_resk > 3 || hp.n_resd > VITS_MAX_RESD || hp.n_ups < 1) return fail(VITS_ERR_UNSUPPORTED, "hparams out of range");
    return load_decoder(m);
  }
  if (hp.n_heads <= 0 || H % hp.n_heads) return fail(VITS_ERR_UNSUPPORTED, "hidden %% n_heads != 0");
  const int dk = H / hp.n_heads;
  if (dk != 32 && dk != 64 && dk != 96) return fail(VITS_ERR_UNSUPPORTED, "head dim %d not in {32,64,96}", dk);
  if (hp.window_size > 4 || hp.window_size < 0) return fail(VITS_ERR_UNSUPPORTED, "window_size > 4");
  if (H % 32 || I % 32 || (I / 2) % 16 || D % 32) return fail(VITS_ERR_UNSUPPORTED, "channel counts must be multiples of 32");
  if (H > LN_MAXV * LN_CG || D > LN_MAXV * LN_CG) return fail(VITS_ERR_UNSUPPORTED, "LayerNorm width > %d", LN_MAXV * LN_CG);
  if (hp.dp_num_bins > 15 || hp.n_ups > VITS_MAX_UPS || hp.n_resk > 3 || hp.n_resd > VITS_MAX_RESD || hp.n_ups < 1)
    return fail(VITS_ERR_UNSUPPORTED, "hparams out of range");
  if (hp.flow_dilation_rate != 1) return fail(VITS_ERR_UNSUPPORTED, "flow dilation_rate != 1");
  m->use_g = G > 0 && hp.n_speakers > 1;

  m->emb = upload(m, tget(m, 2, hp.n_vocab, H, -1, "enc_p.emb.weight"), (size_t)hp.n_vocab * H);
  load_encoder(m, m->enc_p, "enc_p.encoder", hp.n_layers, H, F, hp.kernel_size);
  m->enc_proj = conv_from(m, "enc_p.proj", 2 * I, H, 1, true);
  if (hp.bert_dim < 0 || hp.bert_dim % CONV_CI_T) return fail(VITS_ERR_UNSUPPORTED, "bert_dim %d must be a multiple of %d", hp.bert_dim, CONV_CI_T);
  if (hp.conv_precision != 0 && hp.conv_precision != 1) return fail(VITS_ERR_UNSUPPORTED, "conv_precision %d (0 = fp32, 1 = split-bf16 decoder convs)", hp.conv_precision);
  if (hp.bert_dim > 0) m->bert_proj = conv_from(m, "enc_p.bert_proj", H, hp.bert_dim, 1, true);
  if (m->missing) return VITS_ERR_BLOB;

  // ---- all speaker-conditioning matrices in one GEMV table
  std::vector<float> cW, cB;
  auto add_cond = [&](const float* w, const float* b, int rows) {
    const int off = (int)cB.size();
    if (!w || !b) return off;
    cW.insert(cW.end(), w, w + (size_t)rows * G);
    cB.insert(cB.end(), b, b + rows);
    return off;
  };
  if (m->use_g) {
    m->emb_g = upload(m, tget(m, 2, hp.n_speakers, G, -1, "emb_g.weight"), (size_t)hp.n_speakers * G);
    if (hp.enc_cond_layer >= 0)
      m->cond_enc_off = add_cond(tget(m, 2, H, G, -1, "enc_p.encoder.spk_emb_linear.weight"),
                                 tget(m, 1, H, -1, -1, "enc_p.encoder.spk_emb_linear.bias"), H);
    m->cond_dp_off = add_cond(tget(m, 3, D, G, 1, "dp.cond.weight"), tget(m, 1, D, -1, -1, "dp.cond.bias"), D);
  }

  // ---- duration predictor (reverse path)
  m->dp_pre = conv_from(m, "dp.pre", D, H, 1, true);
  m->dp_proj = conv_from(m, "dp.proj", D, D, 1, true);
  load_dds(m, m->dp_dds, "dp.convs", D, hp.dp_kernel_size, hp.dp_dds_layers);
  m->cf.resize(hp.dp_n_flows);
  char nm[200];
  const int P = 3 * hp.dp_num_bins - 1;
  for (int k = 1; k < hp.dp_n_flows && !m->missing; ++k) {
    ConvFlowW& c = m->cf[k];
    c.pre_w = upload(m, tget(m, 3, D, 1, 1, "dp.flows.%d.pre.weight", 2 * k + 1), D);
    c.pre_b = upload(m, tget(m, 1, D, -1, -1, "dp.flows.%d.pre.bias", 2 * k + 1), D);
    snprintf(nm, sizeof nm, "dp.flows.%d.convs", 2 * k + 1);
    load_dds(m, c.dds, nm, D, hp.dp_kernel_size, hp.dp_dds_layers);
    snprintf(nm, sizeof nm, "dp.flows.%d.proj", 2 * k + 1);
    c.proj = conv_from(m, nm, P, D, 1, true);
  }
  {
    const float* em = tget(m, 2, 2, 1, -1, "dp.flows.0.m");
    const float* el = tget(m, 2, 2, 1, -1, "dp.flows.0.logs");
    if (m->missing) return VITS_ERR_BLOB;
    m->ea_m = upload(m, em, 2); m->ea_logs = upload(m, el, 2);
  }

  // ---- flow
  m->flow.resize(hp.flow_n_flows);
  const int K5 = hp.flow_kernel_size, L = hp.flow_wn_layers;
  for (int f = 0; f < hp.flow_n_flows && !m->missing; ++f) {
    CouplingW& c = m->flow[f];
    snprintf(nm, sizeof nm, "flow.flows.%d.pre", 2 * f);
    c.pre = conv_from(m, nm, H, I / 2, 1, true);
    snprintf(nm, sizeof nm, "flow.flows.%d.pre_transformer", 2 * f);
    load_encoder(m, c.enc, nm, 1, H, H, K5);
    for (int i = 0; i < L && !m->missing; ++i) {
      // in_layer rows permuted to [tanh 32 | sigmoid 32] per 32 channels for the fused gate epilogue
      const float* w = tget(m, 3, 2 * H, H, K5, "flow.flows.%d.enc.in_layers.%d.weight", 2 * f, i);
      const float* b = tget(m, 1, 2 * H, -1, -1, "flow.flows.%d.enc.in_layers.%d.bias", 2 * f, i);
      if (m->missing) break;
      auto gate_src = [&](int r, int ci, int kk) {
        const int j = r / 64, q = r % 64;
        const int orig = q < 32 ? j * 32 + q : H + j * 32 + (q - 32);
        return w[((size_t)orig * H + ci) * K5 + kk];
      };
      c.in_layers.push_back(make_conv2(m, 2 * H, H, K5, b, gate_src, true, [&](int r, int ci, int kk) {  // small-tile kernel: [8 tanh | 8 sigmoid] per 16 rows
        const int j = r / 16, q = r % 16;
        const int orig = q < 8 ? j * 8 + q : H + j * 8 + (q - 8);
        return w[((size_t)orig * H + ci) * K5 + kk];
      }));
      if (hp.conv_precision == 1 && (2 * H) % 128 == 0) add_bf3_packing(m, c.in_layers.back(), gate_src);
      snprintf(nm, sizeof nm, "flow.flows.%d.enc.res_skip_layers.%d", 2 * f, i);
      c.rs_layers.push_back(conv_from(m, nm, i < L - 1 ? 2 * H : H, H, 1, true));
    }
    if (m->use_g)
      c.cond_off = add_cond(tget(m, 3, 2 * H * L, G, 1, "flow.flows.%d.enc.cond_layer.weight", 2 * f),
                            tget(m, 1, 2 * H * L, -1, -1, "flow.flows.%d.enc.cond_layer.bias", 2 * f), 2 * H * L);
    snprintf(nm, sizeof nm, "flow.flows.%d.post", 2 * f);
    c.post = conv_from(m, nm, I / 2, H, 1, true);
    if (!m->missing) {
      const float* pw = tget(m, 3, I / 2, H, 1, "%s.weight", nm);
      const float* pb = tget(m, 1, I / 2, -1, -1, "%s.bias", nm);
      std::vector<const float*> rw(L), rb(L);
      for (int i = 0; i < L; ++i) {
        const int rows = i < L - 1 ? 2 * H : H;
        rw[i] = tget(m, 3, rows, H, 1, "flow.flows.%d.enc.res_skip_layers.%d.weight", 2 * f, i);
        rb[i] = tget(m, 1, rows, -1, -1, "flow.flows.%d.enc.res_skip_layers.%d.bias", 2 * f, i);
      }
      if (!m->missing) {
        for (int i = 0; i < L - 1; ++i) {  // residual half: rows [0, H)
          auto rs_src = [&](int r, int ci, int) { return rw[i][(size_t)r * H + ci]; };
          c.rsx.push_back(make_conv(m, H, H, 1, rb[i], rs_src));
          if (hp.conv_precision == 1 && H % 64 == 0) add_bf3_packing(m, c.rsx.back(), rs_src);
        }
        const int half = I / 2;
        std::vector<double> Wf((size_t)half * L * H, 0.0), bf(half, 0.0);
        for (int o = 0; o < half; ++o) {
          double bacc = pb[o];
          for (int i = 0; i < L; ++i) {
            const int off = i < L - 1 ? H : 0;  // skip rows of layer i (the last layer is all skip)
            for (int k = 0; k < H; ++k) {
              const double pwk = pw[(size_t)o * H + k];
              bacc += pwk * rb[i][off + k];
              const float* wr = rw[i] + (size_t)(off + k) * H;
              double* dst = &Wf[((size_t)o * L + i) * H];
              for (int ci = 0; ci < H; ++ci) dst[ci] += pwk * wr[ci];
            }
          }
          bf[o] = bacc;
        }
        std::vector<float> bff(half);
        for (int o = 0; o < half; ++o) bff[o] = (float)bf[o];
        c.skip_post = make_conv(m, half, L * H, 1, bff.data(), [&](int r, int ci, int) { return (float)Wf[(size_t)r * L * H + ci]; });
      }
    }
  }
  if (m->use_g && hp.dec_type == 1)  // Generator.cond (models.py:869-870, 873-875)
    m->cond_dec_off = add_cond(tget(m, 3, hp.dec_initial_channel, G, 1, "dec.cond.weight"),
                               tget(m, 1, hp.dec_initial_channel, -1, -1, "dec.cond.bias"), hp.dec_initial_channel);
  if (m->missing) return VITS_ERR_BLOB;
  m->cond_rows = (int)cB.size();
  if (m->cond_rows) { m->cond_W = upload(m, cW.data(), cW.size()); m->cond_b = upload(m, cB.data(), cB.size()); }

  return load_decoder(m);
}

// ------------------------------------------------------------------------------------ sessions
// A session owns one HIP stream and a bump-allocated activation workspace sized for
// (B, T_x, T_y).  vits_synthesize() borrows one from the model's pool, so concurrent calls from
// the gRPC server's worker threads (server/tts_server.py:39-40,57) never share buffers.
struct ProfRec { std::string name; std::string kernel; hipEvent_t e0, e1; double flops; };

struct vits_session {
  vits_model* m = nullptr;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipStream_t copy_stream = nullptr;  // D2H of streamed chunks next to the decode of the following window (created on first use)
  char* arena = nullptr;
  size_t arena_bytes = 0, arena_used = 0;
  int* d_err = nullptr;
  int* h_err = nullptr;  // back sessions of the fast path: pinned copy of d_err, written by the phase-2 graph
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  bool profile = false;
  std::vector<ProfRec> prof;
  // graph cache for the device entry point
  typedef std::tuple<const void*, const void*, const void*, const void*, void*, int, int, int, uint64_t, float, float, float, int> GKey;  // (last: persist mask)
  std::map<GKey, hipGraphExec_t> graphs;
  bool use_graph = true;
  const SynthDev* dv = nullptr;  // device parameter block of the graph-replayed fast path (null: scalars by value)
  const unsigned long long* item_seeds = nullptr;  // device [B]: per-item Philox seeds of a solo batch (null: seed + b)
  bool ragged = false;  // full-path calls with B > 1: skip padding tiles of masked stages (set per call)
  bool solo = false;    // VITS_FLAG_SOLO_BATCH: every item as if synthesized alone (noise streams, decoder halo 0)

  // named views (valid after plan())
  int B = 0, Tx = 0, Ty = 0;
  int *len_x = nullptr, *len_y = nullptr, *len_rag = nullptr, *len_tail = nullptr, *dur = nullptr, *cum = nullptr;
  // compact tile maps of the current forward (ragged batches): built on demand, reused by every launch with the
  // same (length array, scale, cap, tile width); reset at the start of each forward
  int* tile_tabs = nullptr;
  int n_tile_tabs = 0;
  std::vector<std::tuple<const int*, int, int, int, int>> tile_keys;
  int64_t* ylen64 = nullptr;
  float *x = nullptr, *qkv = nullptr, *att = nullptr, *y1 = nullptr, *ffh = nullptr, *stats = nullptr;
  float *xb = nullptr, *y1b = nullptr;  // second x / y pair of the LayerNorm-folded encoder schedule
  float *lnst = nullptr;                // per (item, 16-row block, column) LayerNorm partial statistics (conv16 PRO == 3)
  float *condv = nullptr;
  float *dh = nullptr, *dy = nullptr, *dy2 = nullptr, *dc = nullptr, *dz = nullptr, *dpr = nullptr, *logw = nullptr, *dfh = nullptr;
  float *dq1 = nullptr, *dq2 = nullptr;  // second x / y pair of the per-layer DDSConv launches (ping-pong with dy / dy2)
  float *zA = nullptr, *zB = nullptr, *fh = nullptr, *fx = nullptr, *facts = nullptr, *fskip = nullptr;
  std::vector<float*> dec_bufs;
  // persistent step programs of a single utterance (persist.hip.h / persist_plan.hip.h): text encoder and duration predictor
  // (laid out for T_x) and flow (T_y).  LL-cell exchange buffers live inside the arena and are zeroed at every re-plan; the
  // programs are rebuilt at every re-plan; the epoch / completion block survives re-plans (epochs only ever grow).
  struct PersistProg {
    PProgram h;            // host copy of the header
    PProgram* d = nullptr; // device copy
    std::vector<PRec> recs_h;  // per (step, worker) records: host copy (pageable source of the upload)
    PRec* recs_d = nullptr;
    size_t recs_bytes = 0;
    std::vector<int> kinds;    // kind of every step (tools)
    ll_t* ll = nullptr;    // exchange cells
    size_t cells = 0;
    bool ok = false;
    double flops = 0;
  };
  PersistProg ps_enc, ps_sdp, ps_flow;
  // programs of the graph-replayed paths (persist_plan.hip.h): front = text encoder [+ duration predictor] + durations, back = prior
  // sample + flow, full = both in ONE launch (device sessions: the caller brings the frame capacity); index = duration predictor included.
  // They work in the exchange regions of the three programs above plus ps_x (only its ll / cells are used)
  PersistProg ps_front[2], ps_back, ps_full[2], ps_x;
  ll_t *ps_x_stats = nullptr, *ps_x_logw = nullptr, *ps_x_cum = nullptr, *ps_x_leny = nullptr, *ps_x_zp = nullptr;
  int ps_planned_roles = -1;  // ps_roles of the current layout (a change of roles re-plans like a change of shape)
  int ps_roles = 7;        // PERSIST_* mask of the programs this session can ever launch: fronts of the fast path run the text encoder and the
                           // duration predictor, their backs the flow -- cells and records are only laid out / built for those
  bool ps_defer = false;   // the owner calls persist_plan itself after re-pointing shared tensors (backs): session_reserve skips it
  PersistCtl* ps_ctl = nullptr;
  std::vector<std::pair<std::vector<long long>, const float*>> ps_pending;  // parameter packs built by the plan in progress (persist_pack), published after its one stream sync
  const float* ps_bert = nullptr;  // BERT-conditioned voices: the fixed device buffer [bert_dim][Tx] the text-encoder program reads (front sessions: io_d + io_bert)
  bool ps_owner = false;   // device sessions (asynchronous entry point): this session holds the device's persistent-path token for its lifetime
  // staging area of the host-buffer entry points (inputs, noise, audio): a bump allocator that lives with the pooled
  // session, so a steady stream of vits_synthesize calls does no hipMalloc / hipFree (both synchronise the device)
  char* stage = nullptr;
  size_t stage_bytes = 0, stage_used = 0;

  // ---- graph-replayed fast path of vits_synthesize (see "fast path" below).  A FRONT session is laid out for
  // (B, T_x bucket) and owns phase 1 (text encoder .. durations); its BACK sessions, one per frame bucket, own phase 2
  // (prior .. decoder) and read the front's phase-1 results in place.
  int graph_nodes = 0;             // nodes (= launches) of the most recently captured forward graph
  bool ea_pending = false;         // run_duration left the final ElementwiseAffine to durations_kernel (row of z in ea_row)
  int ea_row = 0;
  bool rag_b1 = false;             // single utterance in a frame bucket: decoder sees zeros beyond the item's own end
  bool sdp_always = false;         // device-session option: run the duration predictor even when durations are forced
  char *io_h = nullptr, *io_d = nullptr;  // per-call inputs: pinned host mirror and device copy (SynthDev | lengths | sid | ids | forced)
  size_t io_bytes = 0, io_len = 0, io_sid = 0, io_ids = 0, io_forced = 0, io_seeds = 0, io_bert = 0;  // io_bert: float [B, bert_dim, TxB] (BERT-conditioned voices), 0 = none
  int64_t* h_ylen = nullptr;       // pinned [B] + one int error word behind it
  hipGraphExec_t g1[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [persist*4 + forced*2 + solo]
  std::map<int, vits_session*> backs;
  vits_session* front = nullptr;
  float* out_d = nullptr;          // back: fp32 audio [B, T_y bucket * hop] on the device
  int16_t* pcm_d = nullptr;        // back: int16 PCM, same shape
  char* out_h = nullptr;           // back: pinned host copy of whichever output the call asked for
  size_t out_elems = 0;
  hipGraphExec_t g2[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [persist*4 + solo*2 + pcm]
  uint64_t last_use = 0;
  size_t cache_bytes = 0;          // device bytes this session pins while cached (front: incl. its backs)
};


template <typename T>
static T* bump(vits_session* s, size_t n) {
  size_t off = align_up(s->arena_used, 256);
  s->arena_used = off + n * sizeof(T);
  return s->arena ? reinterpret_cast<T*>(s->arena + off) : nullptr;
}

#include "persist_plan.hip.h"

static inline int persist_mask() { return tl_persist >= 0 ? tl_persist : 0; }
// ... and across PROCESSES: two processes that run the persistent programs on one device at the same time starve each other into the
// poll timeout just the same (seen with two bench ranks on one device: "exchange timed out").  The token therefore includes an advisory
// lock -- flock on a file named after the device's PCI bus id -- taken for exactly as long as the token is held: the length of ONE host
// call (which launches and waits for its kernels), or the lifetime of an asynchronous device session.  Round 5: it used to be held for
// as long as the process had a model on the device, which pinned every other process on that GPU to the launch path even while the
// owner was idle; now an idle process holds nothing, and two busy processes share the programs call by call (a call that finds the
// lock taken runs on launches: slower, never wrong).  Processes that do not share the lock directory (containers with their own /tmp)
// are not covered; the bounded poll loops and the launch-path fallback still are.
// VITS_PERSIST_LOCK=0: no lock; VITS_PERSIST_LOCK_DIR: directory of the lock files (default /tmp).
static int g_proc_lock[64];     // last outcome per device: 0 = not asked yet, 1 = got it (or no lock is used), -1 = another process had it
static int g_proc_lock_fd[64];  // 0 = not opened yet (fd 0 is never ours), -1 = no lock in use, > 0 = the lock file
static bool persist_process_lock(int dev) {  // (g_tok_mu held)
  if (g_proc_lock_fd[dev] == 0) {
    g_proc_lock_fd[dev] = -1;
    if (!(getenv("VITS_PERSIST_LOCK") && atoi(getenv("VITS_PERSIST_LOCK")) == 0)) {
      char bus[64] = "dev";
      if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) snprintf(bus, sizeof bus, "dev%d", dev);
      for (char* c = bus; *c; ++c) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
      char path[512];
      snprintf(path, sizeof path, "%s/vits_mi355_persist_%s.lock", getenv("VITS_PERSIST_LOCK_DIR") ? getenv("VITS_PERSIST_LOCK_DIR") : "/tmp", bus);
      // read-only: another user's process can open it too (flock does not care); O_NOFOLLOW: a symlink planted under the predictable
      // name in a shared directory is refused, not followed (then: no lock, as without a lock directory)
      const int fd = open(path, O_CREAT | O_RDONLY | O_CLOEXEC | O_NOFOLLOW, 0644);
      if (fd > 0) g_proc_lock_fd[dev] = fd;
      else if (fd == 0) close(fd);
    }
  }
  if (g_proc_lock_fd[dev] < 0) { g_proc_lock[dev] = 1; return true; }  // no lock directory / switched off: as without other processes
  if (flock(g_proc_lock_fd[dev], LOCK_EX | LOCK_NB) != 0) {
    if (g_proc_lock[dev] >= 0 && !getenv("VITS_QUIET"))  // (once per change of fortune)
      fprintf(stderr, "[vits_mi355] device %d: another process is running the persistent programs: this call takes the launch path\n", dev);
    g_proc_lock[dev] = -1;
    return false;
  }
  g_proc_lock[dev] = 1;
  return true;
}
static void persist_process_unlock(int dev) {  // (g_tok_mu held)
  if (g_proc_lock_fd[dev] > 0) flock(g_proc_lock_fd[dev], LOCK_UN);
}
static void persist_process_release(int dev) {  // called when a model of this process on `dev` is gone
  std::lock_guard<std::mutex> g(g_tok_mu);
  if (dev < 0 || dev >= 64) return;
  {  // decided HERE, under the token mutex: a model created since the caller looked keeps the file it may be using
    std::lock_guard<std::mutex> gm(g_models_mu);
    for (vits_model* o : g_models) if (o->device == dev) return;
  }
  if (g_tok_busy[dev]) return;  // (a call in flight still holds the lock; it is unlocked with the token)
  if (g_proc_lock_fd[dev] > 0) close(g_proc_lock_fd[dev]);
  g_proc_lock_fd[dev] = 0; g_proc_lock[dev] = 0;
}
static bool persist_token_try(int dev) {
  std::lock_guard<std::mutex> g(g_tok_mu);
  if (dev < 0 || dev >= 64 || g_tok_busy[dev] || !persist_process_lock(dev)) return false;
  g_tok_busy[dev] = true;
  return true;
}
static void persist_token_release(int dev) {
  std::lock_guard<std::mutex> g(g_tok_mu);
  if (dev < 0 || dev >= 64) return;
  persist_process_unlock(dev);
  g_tok_busy[dev] = false;
}
// a host call that launches AND waits for its kernels: owns the token (when it is free) from here to its end
struct PersistScope {
  int dev; bool own;
  explicit PersistScope(int dev_) : dev(dev_) {
    const int cfg = persist_cfg();
    own = cfg != 0 && persist_token_try(dev_);
    tl_persist = own ? cfg : 0;
  }
  void release() { tl_persist = -1; if (own) persist_token_release(dev); own = false; }  // (the caller has waited for its kernels)
  ~PersistScope() { release(); }
};

// lays out every activation buffer for the given capacity; with arena == nullptr only measures
static void plan(vits_session* s, int B, int Tx, int Ty) {
  const vits_hparams& hp = s->m->hp;
  const size_t H = hp.hidden_channels, I = hp.inter_channels, F = hp.filter_channels, D = hp.dp_filter_channels;
  const size_t Fm = F > H ? F : H;
  const size_t Tm = (size_t)(Tx > Ty ? Tx : Ty);
  s->arena_used = 0;
  s->B = B; s->Tx = Tx; s->Ty = Ty;
  s->len_x = bump<int>(s, B); s->len_y = bump<int>(s, B); s->len_rag = bump<int>(s, B + 1); s->len_tail = bump<int>(s, B);
  s->ylen64 = bump<int64_t>(s, B);
  s->dur = bump<int>(s, (size_t)B * Tx); s->cum = bump<int>(s, (size_t)B * Tx);
  s->condv = bump<float>(s, (size_t)B * (s->m->cond_rows + 1));
  s->tile_tabs = bump<int>(s, (size_t)32 * (B + 1));
  // encoder-shaped scratch is shared by the text encoder (T_x) and the flow pre-transformers (T_y)
  s->x = bump<float>(s, B * H * Tm);
  s->qkv = bump<float>(s, B * 3 * H * Tm);
  s->att = bump<float>(s, B * H * Tm);
  s->y1 = bump<float>(s, B * H * Tm);
  s->xb = bump<float>(s, B * H * Tm);
  s->y1b = bump<float>(s, B * H * Tm);
  s->lnst = bump<float>(s, (size_t)B * 16 * Tm * 2);
  s->ffh = bump<float>(s, B * Fm * Tm);
  s->stats = bump<float>(s, B * 2 * I * Tx);
  s->dh = bump<float>(s, B * D * Tx); s->dy = bump<float>(s, B * D * Tx); s->dy2 = bump<float>(s, B * D * Tx);
  s->dc = bump<float>(s, B * D * Tx); s->dfh = bump<float>(s, B * D * Tx);
  s->dq1 = bump<float>(s, B * D * Tx); s->dq2 = bump<float>(s, B * D * Tx);
  s->dz = bump<float>(s, (size_t)B * 2 * Tx); s->dpr = bump<float>(s, (size_t)B * 32 * Tx); s->logw = bump<float>(s, (size_t)B * Tx);
  s->ps_enc.cells = (s->ps_roles & PERSIST_ENC) ? persist_enc_cells(s->m, B, Tx) : 0;
  s->ps_enc.ll = bump<ll_t>(s, s->ps_enc.cells);
  s->ps_sdp.cells = (s->ps_roles & PERSIST_SDP) ? persist_sdp_cells(s->m, B, Tx) : 0;
  s->ps_sdp.ll = bump<ll_t>(s, s->ps_sdp.cells);
  s->ps_flow.cells = (s->ps_roles & PERSIST_FLOW) ? persist_flow_cells(s->m, B, Ty) : 0;
  s->ps_flow.ll = bump<ll_t>(s, s->ps_flow.cells);
  {
    // cells of the multi-stage programs: stats [Tp_x][2I], logw [Tp_x], cum [Tp_x], frame count, z_p [Tp_y][I]
    const size_t Tpx = (size_t)cdiv(Tx, 16) * 16, Tpy = (size_t)cdiv(Ty, 16) * 16;
    s->ps_x.cells = (s->ps_enc.cells || s->ps_flow.cells) ? Tpx * 2 * I + Tpx + Tpx + 16 + Tpy * I : 0;
    s->ps_x.ll = bump<ll_t>(s, s->ps_x.cells);
    ll_t* p = s->ps_x.ll;
    s->ps_x_stats = p; p += Tpx * 2 * I;
    s->ps_x_logw = p; p += Tpx;
    s->ps_x_cum = p; p += Tpx;
    s->ps_x_leny = p; p += 16;
    s->ps_x_zp = p;
  }
  s->zA = bump<float>(s, B * I * Ty); s->zB = bump<float>(s, B * I * Ty);
  s->fh = bump<float>(s, B * H * Ty); s->fx = bump<float>(s, B * H * Ty);
  s->facts = bump<float>(s, B * H * Ty * (size_t)(hp.flow_wn_layers > 0 ? hp.flow_wn_layers : 1));  // gate outputs of all WN layers, stacked
  s->fskip = bump<float>(s, B * H * Ty);
  // decoder: conv_pre out, then per stage: ups out + 3 tmp + 3 res-chain (models.py:1026-1036)
  s->dec_bufs.clear();
  size_t C = hp.dec_initial_channel, T = Ty;
  s->dec_bufs.push_back(bump<float>(s, B * C * T));
  size_t stage_max = 0;
  {
    size_t c = C, t = T;
    for (int i = 0; i < hp.n_ups; ++i) { c /= 2; t *= hp.up_rates[i]; if (c * t > stage_max) stage_max = c * t; }
  }
  // two alternating sets of 7 stage buffers (stage i reads set (i-1)&1's res-chain, writes set i&1)
  for (int k = 0; k < 14; ++k) s->dec_bufs.push_back(bump<float>(s, B * stage_max));
  if (hp.dec_type == 0) {
    size_t P = (size_t)hp.subbands * (hp.istft_n_fft + 2);
    size_t t = T; for (int i = 0; i < hp.n_ups; ++i) t *= hp.up_rates[i];
    s->dec_bufs.push_back(bump<float>(s, B * P * (t + 1)));
    s->dec_bufs.push_back(bump<float>(s, B * hp.subbands * t * hp.istft_hop));
  } else {
    size_t t = T; for (int i = 0; i < hp.n_ups; ++i) t *= hp.up_rates[i];
    s->dec_bufs.push_back(bump<float>(s, B * t));
    s->dec_bufs.push_back(bump<float>(s, 64));
  }
}

static void drop_graphs(vits_session* s) {
  for (auto& kv : s->graphs) hipGraphExecDestroy(kv.second);
  s->graphs.clear();
}

// (re)lays the workspace out for exactly (B,Tx,Ty) so every [B,C,T] tensor is dense; grows the
// arena when needed.  Captured graphs hold raw workspace pointers, so a re-plan drops them.
static int session_reserve(vits_session* s, int B, int Tx, int Ty) {
  if (s->arena && B == s->B && Tx == s->Tx && Ty == s->Ty && s->ps_planned_roles == s->ps_roles) return VITS_OK;
  {
    // the batch-size conv kernels address one item's [C, T] tensor with 32-bit byte offsets (buffer loads, conv_mfma.hip.h bt_ld):
    // every per-item tensor must stay below 2 GiB.  The widest are the decoder stages, C_i x T_y x prod(rates[0..i]).
    const vits_hparams& hp = s->m->hp;
    int widest = hp.filter_channels;
    for (int c : {hp.dec_initial_channel, 2 * hp.hidden_channels, 2 * hp.inter_channels, hp.dp_filter_channels, hp.bert_dim})
      if (c > widest) widest = c;
    long long worst = (long long)widest * (Ty > Tx ? Ty : Tx);
    long long rate = 1;
    for (int i = 0; i < hp.n_ups && i < VITS_MAX_UPS; ++i) {
      rate *= hp.up_rates[i];
      const long long e = (long long)(hp.dec_initial_channel >> (i + 1)) * Ty * rate;
      if (e > worst) worst = e;
    }
    if (worst * 4 >= (1LL << 31)) return fail(VITS_ERR_ARG, "T_y = %d frames: a per-item decoder tensor would exceed 2 GiB", Ty);
  }
  drop_graphs(s);
  char* keep = s->arena;
  s->arena = nullptr;
  plan(s, B, Tx, Ty);  // measure
  const size_t need = s->arena_used + 4096;
  s->arena = keep;
  if (need > s->arena_bytes) {
    if (s->arena) { hipStreamSynchronize(s->stream); hipFree(s->arena); s->arena = nullptr; s->arena_bytes = 0; }
    const size_t want = need + need / 8;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) { s->B = s->Tx = s->Ty = 0; return fail(VITS_ERR_NOMEM, "workspace hipMalloc of %zu bytes failed", want); }
    s->arena = static_cast<char*>(p);
    s->arena_bytes = want;
  }
  plan(s, B, Tx, Ty);
  s->ps_planned_roles = s->ps_roles;
  if (g_poison) {  // 0xFFFFFFFF = NaN; synchronised: stts_synthesize runs the decoder of this session on ITS stream
    hipMemsetAsync(s->arena, 0xFF, s->arena_bytes, s->stream);
    hipStreamSynchronize(s->stream);
  }
  if (!s->ps_defer) persist_plan(s);  // never fails the reserve: a program that cannot be built leaves its stage on the launch path
  return VITS_OK;
}

static int session_new(vits_model* m, vits_session** out) {
  vits_session* s = new vits_session();
  s->m = m;
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc((void**)&s->d_err, sizeof(int)));
  // on the session's own stream: it is non-blocking, i.e. NOT ordered after null-stream work, and hipMemset on device
  // memory may return before it ran -- a plain hipMemset here raced with the first forward's error-word read
  HIP_TRY(hipMemsetAsync(s->d_err, 0, sizeof(int), s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));
  *out = s;
  return VITS_OK;
}

static void session_free(vits_session* s) {
  if (!s) return;
  hipSetDevice(s->m->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  drop_graphs(s);
  for (auto& kv : s->backs) session_free(kv.second);
  s->backs.clear();
  for (int i = 0; i < 8; ++i) { if (s->g1[i]) hipGraphExecDestroy(s->g1[i]); if (s->g2[i]) hipGraphExecDestroy(s->g2[i]); }
  if (s->io_h) hipHostFree(s->io_h);
  if (s->io_d) hipFree(s->io_d);
  if (s->h_ylen) hipHostFree(s->h_ylen);
  if (s->out_d) hipFree(s->out_d);
  if (s->pcm_d) hipFree(s->pcm_d);
  if (s->out_h) hipHostFree(s->out_h);
  for (auto& r : s->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
  if (s->arena) hipFree(s->arena);
  if (s->ps_ctl) hipFree(s->ps_ctl);
  for (vits_session::PersistProg* pp : {&s->ps_enc, &s->ps_sdp, &s->ps_flow, &s->ps_front[0], &s->ps_front[1], &s->ps_back, &s->ps_full[0], &s->ps_full[1]}) {
    if (pp->d) hipFree(pp->d);
    if (pp->recs_d) hipFree(pp->recs_d);
  }
  if (s->stage) hipFree(s->stage);
  if (s->d_err) hipFree(s->d_err);
  if (s->h_err) hipHostFree(s->h_err);
  if (s->ev0) hipEventDestroy(s->ev0);
  if (s->ev1) hipEventDestroy(s->ev1);
  if (s->copy_stream) hipStreamDestroy(s->copy_stream);
  if (s->stream && s->own_stream) hipStreamDestroy(s->stream);
  delete s;
}

static int pool_acquire(vits_model* m, vits_session** out) {
  {
    std::lock_guard<std::mutex> g(m->pool_mu);
    if (!m->pool.empty()) { *out = m->pool.back(); m->pool.pop_back(); return VITS_OK; }
  }
  return session_new(m, out);
}
static void pool_release(vits_model* m, vits_session* s) {
  std::lock_guard<std::mutex> g(m->pool_mu);
  m->pool.push_back(s);
}

// ------------------------------------------------------------------------------------ launch helpers
struct ProfScope {
  vits_session* s; bool on;
  ProfScope(vits_session* s_, const char* name, double flops, const char* kernel = "-") : s(s_), on(s_->profile) {
    if (!on) return;
    ProfRec r; r.name = name; r.kernel = kernel; r.flops = flops;
    hipEventCreate(&r.e0); hipEventCreate(&r.e1);
    hipEventRecord(r.e0, s->stream);
    s->prof.push_back(r);
  }
  void set_kernel(const char* k) { if (on) s->prof.back().kernel = k; }
  void add_template_arg(int v) {  // "name<a,b>" -> "name<a,b,v>"
    if (!on) return;
    std::string& k = s->prof.back().kernel;
    if (!k.empty() && k.back() == '>') { k.pop_back(); k += "," + std::to_string(v) + ">"; }
  }
  ~ProfScope() { if (on) hipEventRecord(s->prof.back().e1, s->stream); }
};


// ---- one persistent step program (persist.hip.h) as ONE launch of P = #CUs workgroups
static bool big_lds_needed(std::atomic<unsigned long long>& done);
static void persist_launch(vits_session* s, vits_session::PersistProg& pp, const char* name, const float* d_noise = nullptr, float nsw = 0.f,
                           uint64_t seed = 0, const int64_t* d_ids = nullptr, const int* d_forced = nullptr, float length_scale = 1.f,
                           float noise_scale = 0.f) {
  vits_model* m = s->m;
  ProfScope ps(s, name, pp.flops, "persist_kernel");
  PCall c;
  c.ctl = s->ps_ctl; c.ids = reinterpret_cast<const long long*>(d_ids); c.noise = d_noise; c.nsw = nsw; c.seed = seed;
  c.solo = s->solo ? 1 : 0; c.dv = s->dv; c.item_seeds = s->item_seeds; c.trace = nullptr;
  c.dbg = m->ps_dbg;
  c.forced = d_forced; c.length_scale = length_scale; c.noise_scale = noise_scale; c.noise_prior = nullptr; c.noise_stride = 0;
  static const int tune = getenv("VITS_PS_TUNE") ? atoi(getenv("VITS_PS_TUNE")) : PS_TUNE_DEFAULT;  // experiment switches (persist.hip.h)
  c.tune = tune;
  static const char* trace_path = getenv("VITS_PS_TRACE");  // tools/ps_trace.py: per-worker, per-step cycle stamps of an EAGER forward
  static const char* trace_name = getenv("VITS_PS_TRACE_PROG");  // which program ("dp.persist" by default)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool want_trace = trace_path && !strcmp(name, trace_name ? trace_name : "dp.persist");
  if (want_trace) hipStreamIsCapturing(s->stream, &cap);
  const size_t trace_n = (size_t)m->n_cu * PS_MAX_STEPS * 8;
  if (want_trace && cap == hipStreamCaptureStatusNone) {
    hipMalloc((void**)&c.trace, trace_n * sizeof(long long));
    hipMemsetAsync(c.trace, 0, trace_n * sizeof(long long), s->stream);
  }
  hipLaunchKernelGGL(persist_kernel, dim3(m->n_cu), dim3(PS_THREADS), 0, s->stream, pp.d, c);
  if (c.trace) {
    std::vector<long long> h(trace_n);
    hipMemcpyAsync(h.data(), c.trace, trace_n * sizeof(long long), hipMemcpyDeviceToHost, s->stream);
    hipStreamSynchronize(s->stream);
    hipFree(c.trace);
    if (FILE* f = fopen(trace_path, "wb")) {
      const int hdr[4] = {m->n_cu, PS_MAX_STEPS, pp.h.n_steps, pp.h.T};
      fwrite(hdr, sizeof hdr, 1, f);
      for (int i = 0; i < pp.h.n_steps; ++i) fwrite(&pp.kinds[i], sizeof(int), 1, f);
      fwrite(h.data(), sizeof(long long), trace_n, f);
      fclose(f);
    }
  }
}

// compact tile map for a ragged launch (see conv_decode_block); nullptr when no table slot is left
static const int* tile_table(vits_session* s, const int* len, int mul, int add, int cap, int tile, int has_cap = 0, int cap_add = 0) {
  auto key = std::make_tuple(len, mul, add + 100000 * cap_add, cap, tile);
  for (size_t i = 0; i < s->tile_keys.size(); ++i)
    if (s->tile_keys[i] == key) return s->tile_tabs + i * (s->B + 1);
  if (s->tile_keys.size() >= 32) return nullptr;
  int* tab = s->tile_tabs + s->tile_keys.size() * (s->B + 1);
  s->tile_keys.push_back(key);
  hipLaunchKernelGGL(ragged_tiles_kernel, dim3(1), dim3(64), 0, s->stream, len, s->B, mul, add, cap, tile, tab, has_cap, cap_add);
  return tab;
}

static void attach_tile_table(vits_session* s, ConvParams& P, int N_T) {
  static const bool pair = !(getenv("VITS_PAIR_MTILES") && atoi(getenv("VITS_PAIR_MTILES")) == 0);
  if (!pair && !P.xcd_mode) P.xcd_mode = 12;
  P.tile_start = nullptr;
  if (!s || !s->arena || s->B == 1) return;  // a single utterance in a padded bucket: the few dead tiles exit early instead
  if (P.rag) P.tile_start = tile_table(s, P.rag, P.rag_out_mul, P.rag_tab_add > P.rag_out_add ? P.rag_tab_add : P.rag_out_add, P.Tout, N_T, 1, P.rag_out_cap_add);  // (tiles the map lists beyond this launch's own limit exit at once)  // (the decoder's rag array carries its cap in rag[B])
  else if (P.skip_len) P.tile_start = tile_table(s, P.len, 1, 0, P.Tout, N_T);
}

template <int WM, int WN, int MI, int NI, int EPI>
static void launch_cfg(vits_session* s, ConvParams& P, int halo) {
  hipStream_t st = s->stream;
  constexpr int M_T = WM * MI * 32, N_T = WN * NI * 32;
  attach_tile_table(s, P, N_T);
  P.ntiles_m = cdiv(P.M, M_T);
  P.ntiles_n = cdiv(P.Tout, N_T);
  P.row_len = N_T + halo;
  const int nblk = P.ntiles_m * P.ntiles_n * P.B * P.n_groups;
  const size_t lds = (size_t)2 * CONV_CI_T * P.row_len * sizeof(float);
  hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, MI, NI, EPI>), dim3(nblk), dim3(WM * WN * 64), lds, st, P);
}

// waves per workgroup of the K-split kernel: 0 = heuristic (ks_pick_waves), else forced (tests / tools: VITS_KS_WAVES)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device: in-process multi-device replicas
// (MultiDeviceSynth) need it once per device, not once per process.  Returns true the first time per (flag word, device).
static bool big_lds_needed(std::atomic<unsigned long long>& done) {
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  return !(done.fetch_or(bit) & bit);
}
static int g_ks_waves = 0;
static int ks_pick_waves(const ConvParams& P, long nblk) {
  static const int env_nw = getenv("VITS_KS_WAVES") ? atoi(getenv("VITS_KS_WAVES")) : 0;
  const int force = g_ks_waves ? g_ks_waves : env_nw;
  if (force == 4 || force == 8 || force == 16) return force;
  int taps = 0;
  for (int g = 0; g < P.n_groups; ++g) { const int t = P.Cin / CONV_CI_T * P.g[g].K; if (t > taps) taps = t; }
  // few workgroups (less than one per CU): 16 waves each, i.e. 4 per SIMD, as long as every wave still gets >= 2 taps;
  // up to two workgroups per CU: 8 waves (the register file holds 2 x 8 waves of <= 128 registers)
  // measured on the c2 forward (profiles/r2_c2_nw*_bench.json.txt): 16 waves win wherever a wave still gets >= 2 taps, also
  // for the grouped decoder launches of ~450 workgroups; 8 waves only pay for launches of a few rounds of the chip
  if (nblk <= 1024 && taps >= 32) return 16;
  if (nblk <= 2048 && taps >= 16) return 8;
  return 4;
}

template <int MI, int NI, int EPI, int NIN, int NW>
static void launch_ks_inst(hipStream_t st, const ConvParams& P, dim3 grid) {
  constexpr size_t lds = (size_t)NW * MI * NI * 16 * 64 * sizeof(float);  // cross-wave reduction only
  auto kern = conv_mfma_ks_kernel<MI, NI, EPI, NIN, NW>;
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};  // once per (kernel instantiation, DEVICE): the attribute is per device
    if (big_lds_needed(done)) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, P);
}

template <int MI, int NI, int EPI>
static void launch_ks(vits_session* s, ConvParams& P, int halo, ProfScope* ps = nullptr) {
  hipStream_t st = s->stream;
  constexpr int M_T = MI * 32, N_T = NI * 32;
  attach_tile_table(s, P, N_T);
  (void)halo;  // no staging window: B fragments come straight from global memory
  P.ntiles_m = cdiv(P.M, M_T);
  P.ntiles_n = cdiv(P.Tout, N_T);
  P.row_len = 0;
  const int nblk = P.ntiles_m * P.ntiles_n * P.B * P.n_groups;
  const dim3 grid(nblk);
  const int nw = ks_pick_waves(P, nblk);
  if (ps) ps->add_template_arg(nw);
#define KS_GO(MI_, NI_, EPI_, NIN_)                                                            \
  do {                                                                                         \
    if (nw == 16) { launch_ks_inst<MI_, NI_, EPI_, NIN_, 16>(st, P, grid); break; }            \
    if (nw == 8) { launch_ks_inst<MI_, NI_, EPI_, NIN_, 8>(st, P, grid); break; }              \
    launch_ks_inst<MI_, NI_, EPI_, NIN_, 4>(st, P, grid);                                      \
  } while (0)
  if (EPI == EPI_STORE && MI * NI == 1 && P.x_split) KS_GO(1, 1, EPI_STORE, 2);
  else if (EPI == EPI_STORE && P.g[0].x2) KS_GO(MI, NI, EPI, (EPI == EPI_STORE ? 3 : 1));
  else KS_GO(MI, NI, EPI, 1);
#undef KS_GO
}

// ---- small-tile kernel (conv_small.hip.h): eligibility + launch
template <int EPI, int NW, int MAXU, int PRO = 0>
static void launch_c16_inst(hipStream_t st, const ConvParams& P, dim3 grid, size_t lds) {
  auto kern = conv16_kernel<EPI, NW, MAXU, PRO>;
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> done{0};  // once per (kernel instantiation, DEVICE)
    if (big_lds_needed(done)) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, P);
}
// returns 0 when the launch cannot take the small-tile kernel, else the wave count it would run with
static int c16_waves(const ConvParams& P, int epi) {
  const ConvGroup& G = P.g[0];
  if (P.n_groups != 1 || !G.w16 || P.ups_u || G.x3 || (G.x2 && !P.x_split) || P.reflect || P.rag || P.Cin % CONV_CI_T) return 0;
  if (epi == EPI_GATE && (P.H % 8)) return 0;
  const int halo = (G.K - 1) * G.dil;
  if (halo > 48) return 0;
  if (P.ln_g && P.Cin > 8 * C16_LN_MAXC) return 0;
  if (P.ln_g && (halo > 16 || P.in_slope != 1.f || P.in_scale != 1.f || P.x_split || P.x_ch_sign != 1 || P.x_ch_off || epi != EPI_STORE)) return 0;
  const size_t lds = ((size_t)P.Cin * c16_row_pitch(16 + halo) + 16 * 32) * sizeof(float);
  if (lds > 150 * 1024) return 0;
  const int units = P.Cin / CONV_CI_T * G.K;
  if (units <= 4 * C16_MAXU) return 4;
  if (units <= 8 * C16_MAXU) return 8;
  return 0;
}
static void launch_c16(vits_session* s, ConvParams& P, int epi, int nw) {
  const ConvGroup& G = P.g[0];
  P.tile_start = nullptr;
  P.ntiles_m = cdiv(epi == EPI_GATE ? 2 * P.H : P.Cout, 16);
  P.ntiles_n = cdiv(P.Tout, 16);
  P.row_len = c16_row_pitch(16 + (G.K - 1) * G.dil);
  size_t lds = ((size_t)P.Cin * P.row_len + ((P.ln_g && !P.ln_stat_in) ? (size_t)nw * 2 * 32 : 0)) * sizeof(float);
  const size_t red = ((size_t)nw * 4 * 64 + (P.ln_stat_out ? 64 : 0)) * sizeof(float);
  if (lds < red) lds = red;
  const dim3 grid(8 * cdiv(P.ntiles_m, 8) * P.ntiles_n * P.B);
  hipStream_t st = s->stream;
  const bool few = cdiv(P.Cin / CONV_CI_T * G.K, nw) <= 8;
#define C16_GO(EPI_)                                                                  \
  do {                                                                                \
    if (nw == 8) {                                                                    \
      if (few) launch_c16_inst<EPI_, 8, 8>(st, P, grid, lds);                         \
      else launch_c16_inst<EPI_, 8, C16_MAXU>(st, P, grid, lds);                      \
    } else {                                                                          \
      if (few) launch_c16_inst<EPI_, 4, 8>(st, P, grid, lds);                         \
      else launch_c16_inst<EPI_, 4, C16_MAXU>(st, P, grid, lds);                      \
    }                                                                                 \
  } while (0)
  if (P.ln_g && P.ln_stat_in) {  // LayerNorm-on-load from the producer's statistics (EPI_STORE only)
    if (nw == 8) {
      if (few) launch_c16_inst<EPI_STORE, 8, 8, 3>(st, P, grid, lds);
      else launch_c16_inst<EPI_STORE, 8, C16_MAXU, 3>(st, P, grid, lds);
    } else {
      if (few) launch_c16_inst<EPI_STORE, 4, 8, 3>(st, P, grid, lds);
      else launch_c16_inst<EPI_STORE, 4, C16_MAXU, 3>(st, P, grid, lds);
    }
  } else if (P.ln_g) {  // LayerNorm-on-load, statistics redone per workgroup (EPI_STORE only)
    if (nw == 8) {
      if (few) launch_c16_inst<EPI_STORE, 8, 8, 2>(st, P, grid, lds);
      else launch_c16_inst<EPI_STORE, 8, C16_MAXU, 2>(st, P, grid, lds);
    } else {
      if (few) launch_c16_inst<EPI_STORE, 4, 8, 2>(st, P, grid, lds);
      else launch_c16_inst<EPI_STORE, 4, C16_MAXU, 2>(st, P, grid, lds);
    }
  } else if (epi == EPI_GATE) C16_GO(EPI_GATE);
  else if (epi == EPI_RESSKIP) C16_GO(EPI_RESSKIP);
  else if (epi == EPI_COUPLE) C16_GO(EPI_COUPLE);
  else C16_GO(EPI_STORE);
#undef C16_GO
}

// 1x1 conv whose B operand is produced by the DDSConv prologue (conv_small.hip.h PRO == 1); P.dds_* set by the caller
static bool c16_dds_ok(const ConvParams& P, int dds_K) {
  return P.g[0].w16 && P.g[0].K == 1 && P.Cin % 32 == 0 && P.Cin <= 16 * DDS_MAXI && dds_K == 3 && P.Cin / CONV_CI_T <= 8 * 8 && P.len &&
         (!P.dds_sw || P.dds_dil <= 9);
}
static void launch_c16_dds(vits_session* s, ConvParams& P, const char* name, double flops) {
  ProfScope ps(s, name, flops, "conv16_kernel<STORE,dds>");
  P.tile_start = nullptr;
  P.ntiles_m = cdiv(P.Cout, 16);
  P.ntiles_n = cdiv(P.Tout, 16);
  P.row_len = 16;
  const size_t lds = ((size_t)P.Cin * (16 + DDS_XP + 8) + 16 * 32) * sizeof(float);  // B tile | x_in over the tap range | reductions | parameters
  const dim3 grid(8 * cdiv(P.ntiles_m, 8) * P.ntiles_n * P.B);
#ifdef CONV_TIMING
  // timing build: VITS_DBG_DDS=<i> prints the phase stamps (cycles since kernel start, block 0, wave 0) of the i-th DDS launch
  static long dds_counter = 0;
  static const long dds_want = getenv("VITS_DBG_DDS") ? atol(getenv("VITS_DBG_DDS")) : -1;
  static long long* dds_buf = nullptr;
  const bool dds_this = (dds_counter++ == dds_want);
  if (dds_this) {
    if (!dds_buf) hipMalloc((void**)&dds_buf, 128 * sizeof(long long));
    hipMemsetAsync(dds_buf, 0, 128 * sizeof(long long), s->stream);
    P.dbg = dds_buf;
  }
  struct DdsPrint {
    bool on; hipStream_t st; long long* buf; const char* name;
    ~DdsPrint() {
      if (!on) return;
      long long h[128];
      hipStreamSynchronize(st);
      hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost);
      fprintf(stderr, "[dds dbg] %s: wave 0 cycles since start: prefetch-issued %lld | phaseA-done %lld | dw+sum1 %lld | staged %lld | mfma-done %lld | end %lld\n", name,
              h[1] - h[0], h[6] - h[0], h[7] - h[0], h[2] - h[0], h[3] - h[0], h[5] - h[0]);
    }
  } dds_print{dds_this, s->stream, dds_buf, name};
#endif
  hipLaunchKernelGGL((conv16_kernel<EPI_STORE, 8, 8, 1>), grid, dim3(512), lds, s->stream, P);
}

// ---- wave-pipelined kernel for the single-utterance decoder's ResBlock convs (conv_small.hip.h conv_wp_kernel)
static int g_wp_mode = 0;  // 0 = heuristic, 1 = never, 2 = whenever eligible (tests)
// split-bf16 kernels: VITS_BF3_PC=1 runs the producer / consumer workgroups (6 waves, conv_bf3.hip.h) instead of the 4-wave form.
// MEASURED (profiles/r3_bf3_ab.txt): 25 % slower -- two 6-wave workgroups per CU leave two MFMA waves per SIMD instead of three, which
// costs more than taking the staging out of their instruction streams gains.  Kept for A/B runs, off by default.
static bool bf3_pc() {
  static const bool on = getenv("VITS_BF3_PC") && atoi(getenv("VITS_BF3_PC")) == 1;
  return on;
}
// weight-fragment slots of conv_bf3_kernel<2, STORE>: 2; VITS_BF3_SLOTS=3 runs the variant with two taps of prefetch lead and the
// activation loads one chunk ahead (MEASURED 8 % slower, profiles/r3_bf3_ab.txt; A/B knob)
static int bf3_slots() {
  static const int n = getenv("VITS_BF3_SLOTS") ? atoi(getenv("VITS_BF3_SLOTS")) : 2;
  return n == 3 ? 3 : 2;
}
static int g_no_bf3 = 0;   // test hook: 1 = a conv_precision == 1 model runs its fp32 kernels (A/B of the split-bf16 variant)
static bool conv_wp_ok(const ConvParams& P, int epi, int halo, bool small) {
  static const int env_mode = getenv("VITS_CONV_WP") ? atoi(getenv("VITS_CONV_WP")) : 0;
  const int mode = g_wp_mode ? g_wp_mode : env_mode;
  if (mode == 1 || epi != EPI_STORE) return false;
  if (P.x_ch_sign != 1 || P.x_ch_off || P.tile_start || P.ups_u || P.reflect || P.in_scale != 1.f || P.ln_g || P.dds_y2 || P.ln_stat_out) return false;
  if (P.Cin % CONV_CI_T || P.Tin < 4 || 32 + halo > WP_PITCH || P.in_slope < 0.f || P.in_slope > 1.f) return false;
  if (P.x_split && (P.n_groups != 1 || P.x_split % CONV_CI_T || !P.g[0].x2)) return false;
  for (int g = 0; g < P.n_groups; ++g)
    if (P.g[g].x3 || (P.g[g].x2 && !P.x_split)) return false;
  if (mode == 2) return true;
  // every wave gets at least one 16-channel chunk; enough columns that 32-column tiles pay (the few-column regime belongs to conv16)
  return small && P.Cin >= 8 * CONV_CI_T && (long)P.B * P.Tout >= 256;
}
static void launch_conv_wp(vits_session* s, ConvParams& P, ProfScope& ps) {
  constexpr int NW = 8;
  P.tile_start = nullptr;
  P.ntiles_m = cdiv(P.M, 32);
  P.ntiles_n = cdiv(P.Tout, 32);
  const size_t lds = (size_t)NW * CONV_CI_T * WP_PITCH * sizeof(float);
  int owned = 0;
  {
    // a grouped launch whose workgroups are all resident at once (two per CU): choose the CU mates (conv_decode_block, mode 11).
    // Measured on the C = 256 stage of c2 (profiles/r3_blocktrace_c2.txt): makespan 26.9 -> 23.0 us.  Launches of several rounds keep
    // the heaviest-first order (the same mapping made the 900-workgroup C = 128 launch 14 % slower).  VITS_WP_ORDER=0: off (A/B).
    // (Tried before that, measured in profiles/r3_xcd_map.txt, removed: giving every XCD one group's input and a range of its weight
    // rows or columns -- fabric traffic -35..44 %, launches 13-15 % slower.)
    static const int order = getenv("VITS_WP_ORDER") ? atoi(getenv("VITS_WP_ORDER")) : 1;
    const int per_xcd = cdiv(P.ntiles_m * P.ntiles_n, 8);
    if (order && P.B == 1 && P.n_groups == 3 && P.g[0].K >= P.g[1].K && P.g[1].K >= P.g[2].K && 3 * per_xcd <= 64 &&
        per_xcd <= 32) {
      P.xcd_mode = 11;
      owned = 8 * 3 * per_xcd;
    }
  }
  const dim3 grid(owned ? owned : P.ntiles_m * P.ntiles_n * P.B * P.n_groups);
  // A/B (round 5, VITS_WP_NW4=1): four waves per workgroup where a contraction has only 8 chunks (the C = 128 decoder stage: one chunk per
  // wave and an 8-way reduction with 8 waves).  Measured: see profiles/r5_wp_nw4.txt
  static const bool nw4 = getenv("VITS_WP_NW4") && atoi(getenv("VITS_WP_NW4")) != 0;
  if (nw4 && P.Cin / CONV_CI_T <= 8 && !owned) {
    ps.set_kernel("conv_wp_kernel<4>");
    hipLaunchKernelGGL(conv_wp_kernel<4>, grid, dim3(4 * 64), (size_t)4 * CONV_CI_T * WP_PITCH * sizeof(float), s->stream, P);
    return;
  }
  ps.set_kernel("conv_wp_kernel<8>");
  hipLaunchKernelGGL(conv_wp_kernel<NW>, grid, dim3(NW * 64), lds, s->stream, P);
}

// Column counts (B x T) up to which the 16-column-tile kernels run.  Round 4, measured on single utterances of 300 - 1000 tokens and on
// batches of 8 / 16 short requests (profiles/r4_c16_threshold.txt): beyond ~256 columns the K-split / wave-pipelined kernels win the
// plain convolutions (although the LayerNorm is then a launch of its own), the gate conv to ~512, the fused DDSConv layer to ~800.
// VITS_C16_COLS overrides both, VITS_C16_DDS_COLS the second.
static long c16_cols_conv(int epi) {  // (the WaveNet gate conv -- 5 taps, 2H rows, tanh * sigmoid epilogue -- crosses over at ~500 columns)
  static const long v = getenv("VITS_C16_COLS") ? atol(getenv("VITS_C16_COLS")) : 0;
  return v ? v : (epi == EPI_GATE ? 512 : 256);
}
static long c16_cols_dds() {
  static const long v = getenv("VITS_C16_DDS_COLS") ? atol(getenv("VITS_C16_DDS_COLS")) : (getenv("VITS_C16_COLS") ? atol(getenv("VITS_C16_COLS")) : 800);
  return v;
}
// would launch_conv route this launch to the small-tile kernel?  (callers that fold a LayerNorm into the consumer's staging
// must know before they drop the LayerNorm launch: only that kernel has the prologue)
static bool conv_takes_c16(const ConvParams& P, int epi) {
  const long c16_cols = c16_cols_conv(epi);
  if (!(g_force_tile == 3 || (g_force_tile == 0 && (long)P.B * P.Tout <= c16_cols))) return false;
  // (launch_conv hands 200..1000-column convs with C_in >= 256 to the wave-pipelined kernel first, unless they carry a prologue or
  // write LayerNorm statistics)
  if (g_force_tile == 0 && (long)P.B * P.Tout > 192 && P.Cin >= 256 && !P.ln_g && !P.dds_y2 && !P.ln_stat_out &&
      conv_wp_ok(P, epi, (P.g[0].K - 1) * P.g[0].dil, true))
    return false;
  return c16_waves(P, epi) != 0;
}

// ---- software-pipelined 64 x 64 kernel (conv_sp.hip.h): stands in for conv_mfma_kernel<2,2,1,1,*> on launches that leave a CU with
// few workgroups.  VITS_SP: 0 = never, 1 = by size (default), 2 = whenever eligible (A/B, tests); VITS_SP_MAXBLK: largest grid it takes.
static int g_sp_mode = -1;
static int sp_mode() {
  static const int env = getenv("VITS_SP") ? atoi(getenv("VITS_SP")) : 1;
  return g_sp_mode >= 0 ? g_sp_mode : env;
}
static bool conv_sp_ok(const ConvParams& P, int epi, int halo) {
  if (sp_mode() == 0 || epi == EPI_GATE) return false;
  if (P.Cin % SP_STAGE_CH || P.ups_u || P.reflect || P.x_split || P.ln_g || P.dds_y2 || P.ln_stat_out || 64 + halo > 128 || P.Tin < 2) return false;
  for (int g = 0; g < P.n_groups; ++g)
    if (P.g[g].x2 || P.g[g].x3) return false;
  return true;
}
template <int EPI>
static void launch_sp(vits_session* s, ConvParams& P, int halo) {
  attach_tile_table(s, P, 64);
  P.ntiles_m = cdiv(P.M, 64);
  P.ntiles_n = cdiv(P.Tout, 64);
  P.row_len = 64 + halo;
  const int nblk = P.ntiles_m * P.ntiles_n * P.B * P.n_groups;
  const size_t lds = (size_t)2 * 4 * P.row_len * SP_PITCH * sizeof(float);  // two stage buffers of four chunks [column][SP_PITCH]
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> done1{0}, done2{0};  // once per (instantiation, device)
    if (P.row_len <= 64) { if (big_lds_needed(done1)) hipFuncSetAttribute((const void*)conv_sp_kernel<EPI, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    else if (big_lds_needed(done2)) hipFuncSetAttribute((const void*)conv_sp_kernel<EPI, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  if (P.row_len <= 64) hipLaunchKernelGGL((conv_sp_kernel<EPI, 1>), dim3(nblk), dim3(256), lds, s->stream, P);
  else hipLaunchKernelGGL((conv_sp_kernel<EPI, 2>), dim3(nblk), dim3(256), lds, s->stream, P);
}
// the 64 x 64 tile of a launch that was routed to conv_mfma_kernel<2,2,1,1,EPI>: the pipelined kernel when the grid is small
static bool sp_takes(const ConvParams& P, int epi, int halo) {
  static const long max_blk = getenv("VITS_SP_MAXBLK") ? atol(getenv("VITS_SP_MAXBLK")) : 2048;
  const long nblk = (long)cdiv(P.M, 64) * cdiv(P.Tout, 64) * P.B * P.n_groups;
  return conv_sp_ok(P, epi, halo) && (sp_mode() == 2 || nblk <= max_blk);
}

// ---- independent-wave 64 x 64 tiles (conv_w1.hip.h): stands in for conv_mfma_kernel<2,2,2,2,STORE> (the ResBlock convs of a batch).
// MEASURED (profiles/r6_w1_ab.txt): parity green, 1.3 % SLOWER than the four-wave kernel on c3 / c4 (20.40 -> 20.66 ms, 127.3 -> 129.1) --
// the barrier was not what the 128 x 128 kernel loses: its launches run at a shader clock of 1.86 - 2.11 GHz instead of 2.4
// (profiles/r6_bt_clock.txt) with the matrix pipe ~90 % busy at THAT clock, and this form moves 2.6 x the activation bytes through L2.
// Kept as an A/B (like the producer / consumer split-bf16 kernel): VITS_W1=1 takes the launches the 128 x 128 kernel would; default off.
static int w1_mode() {
  static const int env = getenv("VITS_W1") ? atoi(getenv("VITS_W1")) : 0;
  return env;
}
static bool conv_w1_ok(const ConvParams& P, int epi, int halo) {
  if (w1_mode() == 0 || epi != EPI_STORE) return false;
  if (P.M % 64 || P.Cin % CONV_CI_T || P.ups_u || P.reflect || P.x_split || P.ln_g || P.dds_y2 || P.ln_stat_out || 64 + halo > W1_PITCH || P.Tin < 2) return false;
  for (int g = 0; g < P.n_groups; ++g)
    if (P.g[g].x2 || P.g[g].x3) return false;
  return true;
}
static void launch_w1(vits_session* s, ConvParams& P, int halo) {
  attach_tile_table(s, P, 64);
  P.ntiles_m = cdiv(P.M, 64);
  P.ntiles_n = cdiv(P.Tout, 64);
  P.row_len = 64 + halo;
  const int nblk = P.ntiles_m * P.ntiles_n * P.B * P.n_groups;
  const size_t lds = (size_t)CONV_CI_T * W1_PITCH * sizeof(float);
  if (P.row_len <= 64) hipLaunchKernelGGL((conv_w1_kernel<EPI_STORE, 1>), dim3(nblk), dim3(64), lds, s->stream, P);
  else hipLaunchKernelGGL((conv_w1_kernel<EPI_STORE, 2>), dim3(nblk), dim3(64), lds, s->stream, P);
}

// dispatch on epilogue + problem size.  halo = max over groups of (K-1)*dil (or the polyphase spread).
// Large problems (>= 2 workgroups per CU with 64x64 tiles) use the big-tile kernel (more operand
// reuse); everything smaller uses the K-split kernel so that one utterance still fills the chip.
static void launch_conv(vits_session* s, ConvParams& P, int epi, const char* name, int halo_override = -1) {
  int halo = 0;
  double macs = 0;
  for (int g = 0; g < P.n_groups; ++g) {
    const int hg = halo_override >= 0 ? halo_override : (P.g[g].K - 1) * P.g[g].dil;
    if (hg > halo) halo = hg;
    // rows the conv actually computes: the gate kernel stores H channels but contracts 2H rows (tanh | sigmoid)
    macs += (double)(epi == EPI_GATE ? 2 * P.H : P.Cout) * P.Cin * P.g[g].K;
  }
  ProfScope ps(s, name, 2.0 * macs * (double)P.Tout * P.B);
  if (ps.on) {  // tools/profile_ops.py with VITS_PROF_SHAPES=1: one report line per distinct launch shape
    static const bool shapes = getenv("VITS_PROF_SHAPES") != nullptr;
    if (shapes) {
      char sh[96];
      snprintf(sh, sizeof sh, "/M%d.K%dx%d.N%dx%d.g%d", P.M, P.Cin, P.g[0].K, P.B, P.Tout, P.n_groups);
      s->prof.back().name += sh;
    }
  }
#ifdef CONV_TIMING
  hipStream_t st = s->stream;
  // timing build only: VITS_DBG_LAUNCH=<i> attaches the phase-stamp buffer to the i-th conv launch of the process
  // and prints the stamps (cycles since kernel start, block 0) right after it
  static long dbg_counter = 0;
  static const long dbg_want = getenv("VITS_DBG_LAUNCH") ? atol(getenv("VITS_DBG_LAUNCH")) : -1;
  static long long* dbg_buf = nullptr;
  // VITS_DBG_GROUPED=<n>: the n-th three-group launch of the process instead (the single-utterance decoder's ResBlock launches)
  static long grouped_counter = 0;
  static const long grouped_want = getenv("VITS_DBG_GROUPED") ? atol(getenv("VITS_DBG_GROUPED")) : -1;
  const bool dbg_this = (dbg_counter++ == dbg_want) || (P.n_groups == 3 && grouped_counter++ == grouped_want);
  if (dbg_this) {
    if (!dbg_buf) hipMalloc((void**)&dbg_buf, (128 + 4 * 4000) * sizeof(long long));
    hipMemsetAsync(dbg_buf, 0, (128 + 4 * 4000) * sizeof(long long), st);
    P.dbg = dbg_buf;
  }
  struct DbgPrint {
    bool on; hipStream_t st; long long* buf; const char* name; int M, Cin, K, T, B;
    ~DbgPrint() {
      if (!on) return;
      long long h[128];
      hipStreamSynchronize(st);
      hipMemcpy(h, buf, sizeof h, hipMemcpyDeviceToHost);
      fprintf(stderr, "[in-forward conv dbg] %s M=%d Cin=%d K=%d T=%d B=%d\n", name, M, Cin, K, T, B);
      for (int w = 0; w < 4; ++w)
        fprintf(stderr, "   wave %d: +%lld first-loads-issued  +%lld loop_done  +%lld barrier  +%lld reduced  +%lld end\n", w, h[w * 8 + 1] - h[w * 8],
                h[w * 8 + 2] - h[w * 8], h[w * 8 + 3] - h[w * 8], h[w * 8 + 4] - h[w * 8], h[w * 8 + 5] - h[w * 8]);
      // block trace: "blk <id> <start> <end> <hw_id> <xcc_id>" (wall clock, 10 ns units, relative to the earliest start)
      std::vector<long long> t(4 * 4000);
      hipMemcpy(t.data(), buf + 128, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
      long long t0 = 0;
      for (int i = 0; i < 4000; ++i) if (t[4 * i] && (!t0 || t[4 * i] < t0)) t0 = t[4 * i];
      for (int i = 0; i < 4000; ++i)
        if (t[4 * i]) fprintf(stderr, "blk %d %lld %lld %lld %lld\n", i, t[4 * i] - t0, t[4 * i + 1] ? t[4 * i + 1] - t0 : -1, t[4 * i + 2], t[4 * i + 3]);
    }
  } dbg_print{dbg_this, st, dbg_buf, name, P.Cout, P.Cin, P.g[0].K, P.Tout, P.B};
#endif
  // heaviest group first (longest-processing-time order; see the big-tile kernel's block decode)
  for (int a = 0; a < P.n_groups; ++a)
    for (int c = a + 1; c < P.n_groups; ++c)
      if (P.g[c].K > P.g[a].K) { ConvGroup t = P.g[a]; P.g[a] = P.g[c]; P.g[c] = t; }
  const long blocks64 = (long)cdiv(P.M, 64) * cdiv(P.Tout, 64) * P.B * P.n_groups;
  static const long ks_threshold = getenv("VITS_KS_THRESHOLD") ? atol(getenv("VITS_KS_THRESHOLD")) : 512;
  bool small = g_force_tile == 2 || (g_force_tile == 0 && blocks64 < ks_threshold);
  // (the polyphase upsamplers and the 32-row conv_post leave the K-split kernel earlier: 300-token utterance ups 0.20 -> 0.135 ms,
  //  conv_post 0.092 -> 0.046 ms -- profiles/r4_c16_threshold.txt)
  if (small && g_force_tile == 0 && epi == EPI_STORE && (P.ups_u || P.M % 64 == 32) && blocks64 >= 256) small = false;
  if (!P.g[0].x2 && P.in_scale != 1.0f) small = false;  // the K-split kernel folds in_scale into the multi-input sum only
  // few-column regime (a single utterance's encoder / duration predictor / flow): many small workgroups, LDS-staged B
  const long c16_cols = c16_cols_conv(epi);
  // between ~200 and ~1000 columns the 16-column tiles re-read every weight once per column tile (19 times at 304 columns: the
  // StableTTS estimator, 20 us per conv): the wave-pipelined 32x32 kernel takes those when it can
  const bool wp_first = g_force_tile == 0 && (long)P.B * P.Tout > 192 && P.Cin >= 256 && !P.ln_g && !P.dds_y2 && conv_wp_ok(P, epi, halo, small);
  if (!wp_first && (g_force_tile == 3 || (g_force_tile == 0 && (long)P.B * P.Tout <= c16_cols))) {
    const int nw16 = c16_waves(P, epi);
    if (nw16) {
      static const char* names[4] = {"conv16_kernel<STORE>", "conv16_kernel<GATE>", "conv16_kernel<RESSKIP>", "conv16_kernel<COUPLE>"};
      ps.set_kernel(P.ln_g ? "conv16_kernel<STORE,ln>" : names[epi]);
      ps.add_template_arg(nw16);
      launch_c16(s, P, epi, nw16);
      return;
    }
  }
  if (sp_mode() == 2 && g_force_tile == 0 && conv_sp_ok(P, epi, halo)) {  // A/B: the pipelined kernel wherever it is eligible
    static const char* names[4] = {"conv_sp_kernel<STORE>", "-", "conv_sp_kernel<RESSKIP>", "conv_sp_kernel<COUPLE>"};
    ps.set_kernel(names[epi]);
    if (epi == EPI_STORE) launch_sp<EPI_STORE>(s, P, halo);
    else if (epi == EPI_RESSKIP) launch_sp<EPI_RESSKIP>(s, P, halo);
    else launch_sp<EPI_COUPLE>(s, P, halo);
    return;
  }
  if (epi == EPI_GATE) {
    if (small) { ps.set_kernel("conv_mfma_ks_kernel<2,1,GATE,1>"); launch_ks<2, 1, EPI_GATE>(s, P, halo, &ps); }
    else if (!g_no_bf3 && P.g[0].wb && P.n_groups == 1 && P.M % 128 == 0 && P.x_ch_sign == 1 && !P.x_ch_off && !P.g[0].x2 && !P.ln_g &&
             P.in_scale >= 0.f && P.in_slope >= 0.f && P.in_slope <= 1.f &&
             (long)cdiv(P.M, 128) * cdiv(P.Tout, 128) * P.B >= 256) {  // split-bf16 WaveNet gate conv (conv_precision == 1)
      ps.set_kernel("conv_bf3_kernel<2,GATE>");
      attach_tile_table(s, P, 128);
      P.ntiles_m = cdiv(P.M, 128);
      P.ntiles_n = cdiv(P.Tout, 128);
      P.row_len = 128 + halo;
      const size_t lds = (size_t)2 * 2 * P.row_len * (BF3_PITCH * 2);
      if (bf3_pc()) hipLaunchKernelGGL((conv_bf3pc_kernel<2, EPI_GATE>), dim3(P.ntiles_m * P.ntiles_n * P.B), dim3(384), lds, s->stream, P);
      else hipLaunchKernelGGL((conv_bf3_kernel<2, EPI_GATE>), dim3(P.ntiles_m * P.ntiles_n * P.B), dim3(256), lds, s->stream, P);
    } else {  // (128 x 128 tiles for the gate conv: 2.30 against 1.77 ms per c3 forward, round 4, profiles/r4_c3_tile_ab.txt)
      // Round 6: a grid of 1 - 3 four-wave workgroups per CU (all resident at once) lasts as long as the CU with the most of them; the
      // same wave tiles in TWO-wave workgroups of 128 x 32 halve the quantum (c3: 580 tiles -> 1160).  VITS_GATE2W: 0 = never,
      // 1 = by grid size (default), 2 = whenever the window fits (A/B)
      static const int g2w = getenv("VITS_GATE2W") ? atoi(getenv("VITS_GATE2W")) : 1;
      const long nblk64 = (long)cdiv(P.M, 128) * cdiv(P.Tout, 64) * P.B;
      if (g_force_tile == 0 && g2w && 32 + halo <= 64 && (g2w == 2 || nblk64 <= 1536)) {
        ps.set_kernel("conv_mfma_kernel<2,1,2,1,GATE>"); launch_cfg<2, 1, 2, 1, EPI_GATE>(s, P, halo);
      } else {
        ps.set_kernel("conv_mfma_kernel<2,2,2,1,GATE>"); launch_cfg<2, 2, 2, 1, EPI_GATE>(s, P, halo);
      }
    }
    return;
  }
  if (epi == EPI_RESSKIP) {
    if (small) { ps.set_kernel("conv_mfma_ks_kernel<1,1,RESSKIP,1>"); launch_ks<1, 1, EPI_RESSKIP>(s, P, halo, &ps); }
    else if (g_force_tile == 0 && sp_takes(P, epi, halo)) { ps.set_kernel("conv_sp_kernel<RESSKIP>"); launch_sp<EPI_RESSKIP>(s, P, halo); }
    else { ps.set_kernel("conv_mfma_kernel<2,2,1,1,RESSKIP>"); launch_cfg<2, 2, 1, 1, EPI_RESSKIP>(s, P, halo); }
    return;
  }
  if (epi == EPI_COUPLE) {
    if (small) { ps.set_kernel("conv_mfma_ks_kernel<1,1,COUPLE,1>"); launch_ks<1, 1, EPI_COUPLE>(s, P, halo, &ps); }
    else if (g_force_tile == 0 && sp_takes(P, epi, halo)) { ps.set_kernel("conv_sp_kernel<COUPLE>"); launch_sp<EPI_COUPLE>(s, P, halo); }
    else { ps.set_kernel("conv_mfma_kernel<2,2,1,1,COUPLE>"); launch_cfg<2, 2, 1, 1, EPI_COUPLE>(s, P, halo); }
    return;
  }
  if (conv_wp_ok(P, epi, halo, small)) { launch_conv_wp(s, P, ps); return; }
  if (small) {
    const long blocks32 = (long)cdiv(P.M, 32) * cdiv(P.Tout, 32) * P.B * P.n_groups;
    const bool multi = P.g[0].x2 != nullptr;
    static const int ks_shape = getenv("VITS_KS_SHAPE") ? atoi(getenv("VITS_KS_SHAPE")) : 0;  // tools/ks_shapes.py: 11 or 12 forces the tile
    if (P.x_split) { ps.set_kernel("conv_mfma_ks_kernel<1,1,STORE,2>"); launch_ks<1, 1, EPI_STORE>(s, P, halo, &ps); return; }
    if (ks_shape == 12 || (ks_shape == 0 && blocks32 > 2048)) { ps.set_kernel(multi ? "conv_mfma_ks_kernel<1,2,STORE,3>" : "conv_mfma_ks_kernel<1,2,STORE,1>"); launch_ks<1, 2, EPI_STORE>(s, P, halo, &ps); }
    else { ps.set_kernel(multi ? "conv_mfma_ks_kernel<1,1,STORE,3>" : "conv_mfma_ks_kernel<1,1,STORE,1>"); launch_ks<1, 1, EPI_STORE>(s, P, halo, &ps); }
    return;
  }
  // 32-row outputs (polyphase upsamplers with C_out % 64 != 0, the 32-channel last stage of HiFi-GAN V1): a 64-row tile
  // would spend half its MFMAs on padding rows -> 32 x 128 tiles
  if ((P.ups_u && (P.ups_cout % 64)) || (!P.ups_u && P.M % 64 == 32)) {
    ps.set_kernel("conv_mfma_kernel<1,4,1,1,STORE>"); launch_cfg<1, 4, 1, 1, EPI_STORE>(s, P, halo); return;
  }
  auto bf3_ok = [&]() {
    bool ok = !g_no_bf3 && !P.reflect && !P.x_split && P.x_ch_sign == 1 && !P.x_ch_off && !P.ln_g;
    ok = ok && P.in_scale >= 0.f && P.in_slope >= 0.f && P.in_slope <= 1.f;  // the staging pass evaluates the leaky ReLU as a max
    if (P.ups_u && (P.ups_cout % 128 || P.n_groups != 1)) ok = false;  // a 128-row tile must lie inside one polyphase phase
    for (int g = 0; g < P.n_groups; ++g) ok = ok && P.g[g].wb && !P.g[g].x2 && !P.g[g].x3;
    return ok;
  };
  auto bf3_go = [&](int mi) {  // split-bf16 variant (hparams.conv_precision == 1): same staging pattern, 3 bf16 MFMAs per 16 channels x tap
    ps.set_kernel(mi == 2 ? "conv_bf3_kernel<2>" : "conv_bf3_kernel<1>");
    attach_tile_table(s, P, 128);
    P.ntiles_m = cdiv(P.M, 64 * mi);
    P.ntiles_n = cdiv(P.Tout, 128);
    P.row_len = 128 + halo;
    const size_t lds = (size_t)2 * 2 * P.row_len * (BF3_PITCH * 2);
    const dim3 grid(P.ntiles_m * P.ntiles_n * P.B * P.n_groups);
    if (bf3_pc()) {
      if (mi == 2) hipLaunchKernelGGL((conv_bf3pc_kernel<2, EPI_STORE>), grid, dim3(384), lds, s->stream, P);
      else hipLaunchKernelGGL((conv_bf3pc_kernel<1, EPI_STORE>), grid, dim3(384), lds, s->stream, P);
    } else if (mi == 2 && bf3_slots() == 3) hipLaunchKernelGGL((conv_bf3_kernel<2, EPI_STORE, 3>), grid, dim3(256), lds, s->stream, P);
    else if (mi == 2) hipLaunchKernelGGL((conv_bf3_kernel<2, EPI_STORE>), grid, dim3(256), lds, s->stream, P);
    else hipLaunchKernelGGL((conv_bf3_kernel<1, EPI_STORE>), grid, dim3(256), lds, s->stream, P);
  };
  // 64-row outputs at batch size: 64 x 128 tiles (twice the columns per weight fragment of the 64 x 64 tile)
  if (!P.ups_u && P.M == 64 && (long)cdiv(P.Tout, 128) * P.B * P.n_groups >= 512) {
    if (bf3_ok()) { bf3_go(1); return; }
    ps.set_kernel("conv_mfma_kernel<2,2,1,2,STORE>"); launch_cfg<2, 2, 1, 2, EPI_STORE>(s, P, halo); return;
  }
  const long big_blocks = (long)cdiv(P.M, 128) * cdiv(P.Tout, 128) * P.B * P.n_groups;
  const bool m_fits = (P.M % 128 == 0) && (!P.ups_u || P.ups_cout % 128 == 0);
  static const long big_min = getenv("VITS_BIG_BLOCKS") ? atol(getenv("VITS_BIG_BLOCKS")) : 512;
  if (m_fits && big_blocks >= big_min) {
    if (bf3_ok()) { bf3_go(2); return; }
    if (g_force_tile == 0 && conv_w1_ok(P, epi, halo)) { ps.set_kernel("conv_w1_kernel<STORE>"); launch_w1(s, P, halo); return; }
    ps.set_kernel("conv_mfma_kernel<2,2,2,2,STORE>"); launch_cfg<2, 2, 2, 2, EPI_STORE>(s, P, halo); return;
  }
  // 64-row multiples at batch size (encoder / flow STORE convs: 192, 576, 768 rows) of a conv_precision == 1 model
  if (P.M % 64 == 0 && (long)cdiv(P.M, 64) * cdiv(P.Tout, 128) * P.B * P.n_groups >= 256 && bf3_ok()) { bf3_go(1); return; }
  // (64 x 128 fp32 tiles for these convs were measured on the c3 batch in round 4: 2.21 - 2.42 ms against 2.17 ms per forward for the
  //  64 x 64 tiles -- profiles/r4_c3_tile_ab.txt; not a tile-shape problem)
  if (g_force_tile == 0 && sp_takes(P, epi, halo)) { ps.set_kernel("conv_sp_kernel<STORE>"); launch_sp<EPI_STORE>(s, P, halo); return; }
  ps.set_kernel("conv_mfma_kernel<2,2,1,1,STORE>");
  launch_cfg<2, 2, 1, 1, EPI_STORE>(s, P, halo);
}

// common-case parameter block: one group, same-length 'same'-padded Conv1d over [B,C,T]
static ConvParams conv_params(const ConvW& W, const float* x, float* y, int B, int T, int dil, int pad_l) {
  ConvParams P;
  memset(&P, 0, sizeof P);
  P.n_groups = 1;
  P.g[0].x = x; P.g[0].w = W.w; P.g[0].w16 = W.w16; P.g[0].wb = W.wb; P.g[0].bias = W.bias; P.g[0].y = y;
  P.g[0].K = W.K; P.g[0].dil = dil; P.g[0].pad_l = pad_l; P.g[0].n_sg = W.n_sg;
  P.B = B; P.Cin = W.Cin; P.x_ch_off = 0; P.x_ch_sign = 1;
  P.x_bstride = (long long)W.Cin * T; P.Tin = T; P.Tin_stride = T;
  P.M = W.Mpad; P.Cout = W.M; P.Tout = T; P.Tout_stride = T; P.y_bstride = (long long)W.M * T;
  P.in_slope = 1.f; P.in_scale = 1.f;
  return P;
}

// masked-stage conv of a ragged batch: tiles beyond len[b] are skipped (conv_mfma.hip.h, skip_len)
static void mark_masked(vits_session* s, ConvParams& P, const int* len) {
  if (s->ragged) { P.skip_len = 1; P.len = len; }
}

static void launch_ln(vits_session* s, const float* a, const float* b, const float* base, float* y, const float* gamma,
                      const float* beta, const int* len, int B, int C, int T, int gelu, int mask) {
  ProfScope ps(s, "layernorm", 0, "layernorm_c_kernel");
  LNParams P{a, b, base, y, gamma, beta, len, C, T, gelu, mask, (s->ragged && len) ? 1 : 0, 0, 1e-5f, nullptr, nullptr};
  launch_layernorm(s->stream, P, B);
}

// ek / ev: relative-position tables [2W+1][dk] or null (plain scaled-dot-product attention: StableTTS DiT blocks, BERT)
static void launch_attention_raw(vits_session* s, const float* qkv, const float* ek, const float* ev, const int* len, float* out, int B,
                                 int H, int T, int nh, int W) {
  const int dk = H / nh;
  struct { const float* ek; const float* ev; } L{ek, ev};
  // 16-query tiles (more, smaller workgroups) while the 32-query MFMA kernel's grid would not fill the chip: measured round 4
  // (profiles/r4_c16_threshold.txt) single utterances of 200 - 600 tokens (T_y 600 - 1800) -15..-35 % attention time against the old rule
  // (T <= 512), the 32-item batch c3 -6 % (its 200-token text side now runs the MFMA kernel).  VITS_ATT16_MAXT=<T> restores a pure T rule.
  static const int t16_max = getenv("VITS_ATT16_MAXT") ? atoi(getenv("VITS_ATT16_MAXT")) : 0;
  const bool small_grid = (long)cdiv(T, 32) * nh * B < 256;
  const bool use16 = g_attn_impl == 3 || (g_attn_impl == 0 && (t16_max ? T <= t16_max : (T <= 64 || (small_grid && T <= 4096))));
  ProfScope ps(s, "attention", 4.0 * (double)B * H * T * T,
               use16 ? "relpos_attention16_kernel" : (g_attn_impl == 1 ? "relpos_attention_kernel" : "relpos_attention_mfma_kernel"));
  if (use16) {  // short sequences: 16-query tiles, more and smaller workgroups
    dim3 grid(cdiv(T, 16), nh, B);
    const bool w8 = T > 64;
    const int nwv = w8 ? 8 : 4;
    const int wreg = 16 * (dk + 4) + 12 * 16 + 12 * 16, nv = (dk / 16) * 4 + 2;
    const size_t lds = (size_t)nwv * (wreg > nv * 64 ? wreg : nv * 64) * sizeof(float);
#define ATT16_GO(DK_)                                                                                                                  \
  do {                                                                                                                                 \
    if (w8) hipLaunchKernelGGL((relpos_attention16_kernel<DK_, 8>), grid, dim3(512), lds, s->stream, qkv, L.ek, L.ev, len, out, H, T, W); \
    else hipLaunchKernelGGL((relpos_attention16_kernel<DK_, 4>), grid, dim3(256), lds, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);   \
  } while (0)
    if (dk == 96) ATT16_GO(96);
    else if (dk == 64) ATT16_GO(64);
    else ATT16_GO(32);
#undef ATT16_GO
    return;
  }
  if (g_attn_impl != 1) {  // fp32-MFMA flash kernel (32-query tiles)
    dim3 grid(cdiv(T, 32), nh, B);
    const int wreg = dk * 33 + 10 * 32 + 9 * 32;
    const size_t lds = (size_t)4 * wreg * sizeof(float);
    if (dk == 96) hipLaunchKernelGGL((relpos_attention_mfma_kernel<96>), grid, dim3(256), lds, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);
    else if (dk == 64) hipLaunchKernelGGL((relpos_attention_mfma_kernel<64>), grid, dim3(256), lds, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);
    else hipLaunchKernelGGL((relpos_attention_mfma_kernel<32>), grid, dim3(256), lds, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);
    return;
  }
  dim3 grid(cdiv(T, ATT_TQ), nh, B);
  if (dk == 96) hipLaunchKernelGGL((relpos_attention_kernel<96>), grid, dim3(256), 0, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);
  else if (dk == 64) hipLaunchKernelGGL((relpos_attention_kernel<64>), grid, dim3(256), 0, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);
  else hipLaunchKernelGGL((relpos_attention_kernel<32>), grid, dim3(256), 0, s->stream, qkv, L.ek, L.ev, len, out, H, T, W);
}

static void launch_attention(vits_session* s, const float* qkv, const EncLayerW& L, const int* len, float* out, int B, int H, int T) {
  launch_attention_raw(s, qkv, L.ek, L.ev, len, out, B, H, T, s->m->hp.n_heads, s->m->hp.window_size);
}

// attentions.Encoder.forward (attentions.py:48-65).  x in place [B,H,T]; final_base (optional):
// out = final_base + encoder(x) (the VITS2 residual at models.py:377), written to final_out.
//
// Few-column regime (every conv of the layer runs on the small-tile kernel): no LayerNorm launches.  norm_layers_1 is folded
// into the staging of conv_1 of the FFN, norm_layers_2 (and the speaker-embedding add before layer `cond_layer`,
// attentions.py:52-56) into the staging of the next layer's fused q/k/v conv; each writes the normalised tensor once (the
// residual path needs it).  The last norm_layers_2 is handed to the caller's consumer through `pend` when it has one
// (TextEncoder.proj), otherwise it runs as the LayerNorm kernel (flow: + final_base, masked).
struct PendingLN { const float* raw = nullptr; const float* g = nullptr; const float* b = nullptr; const float* stat = nullptr; int nmb = 0; };

static bool enc_fold_ok(vits_session* s, const EncoderW& E, int B, int T) {
  static const bool no_fold = getenv("VITS_NO_LN_FOLD") != nullptr;  // A/B switch for tools/ and tests
  if (no_fold || E.layers.empty() || !s->xb || !s->y1b) return false;
  const EncLayerW& L = E.layers[0];
  static const float dummy = 0.f;
  ConvParams P = conv_params(L.qkv, s->x, s->qkv, B, T, 1, 0);
  P.ln_g = &dummy;
  if (!conv_takes_c16(P, EPI_STORE)) return false;
  P = conv_params(L.f1, s->x, s->ffh, B, T, 1, (E.K - 1) / 2);
  P.ln_g = &dummy;
  if (!conv_takes_c16(P, EPI_STORE)) return false;
  return true;
}

static void run_encoder(vits_session* s, const EncoderW& E, float* x, const int* len, int B, int T, int cond_layer,
                        int cond_off, const float* final_base, float* final_out, PendingLN* pend = nullptr) {
  vits_model* m = s->m;
  const int H = E.H, F = E.F, K = E.K;
  const int n = (int)E.layers.size();
  const bool fold = enc_fold_ok(s, E, B, T);
  // statistics of the folded LayerNorms come from the producing conv's epilogue (conv16 PRO == 3) instead of being redone by every
  // workgroup of the consumer: test hook vits_debug_ln_stats
  bool pstat = fold && g_ln_stats && s->lnst && H % 16 == 0 && H / 16 <= 16;
  if (pstat) {  // the producers (conv_o, FFN conv_2) must run on the small-tile kernel too: only its epilogue writes the statistics
    float dummy_stat = 0.f;
    const EncLayerW& L0 = E.layers[0];
    ConvParams Pp = conv_params(L0.o, s->att, s->y1, B, T, 1, 0);
    Pp.ln_stat_out = &dummy_stat;
    pstat = conv_takes_c16(Pp, EPI_STORE);
    Pp = conv_params(L0.f2, s->ffh, s->y1, B, T, 1, (K - 1) / 2);
    Pp.ln_stat_out = &dummy_stat; Pp.in_mask = 1; Pp.out_mask = 1; Pp.len = len;
    pstat = pstat && conv_takes_c16(Pp, EPI_STORE);
  }
  PendingLN prev;  // norm_layers_2 of the previous layer, not yet applied (fold only)
  for (int i = 0; i < n; ++i) {
    const EncLayerW& L = E.layers[i];
    const bool cond_here = i == cond_layer && cond_off >= 0;
    if (cond_here && !prev.raw)
      hipLaunchKernelGGL(add_vec_mask_kernel, dim3(cdiv(T, 64), H, B), dim3(64), 0, s->stream, x, s->condv, m->cond_rows,
                         cond_off, len, H, T);
    ConvParams P = conv_params(L.qkv, prev.raw ? prev.raw : x, s->qkv, B, T, 1, 0);
    if (prev.raw) {
      P.ln_g = prev.g; P.ln_b = prev.b; P.ln_out = x; P.len = len;
      P.ln_stat_in = prev.stat; P.ln_nmb = prev.nmb;
      if (cond_here) { P.ln_vec = s->condv; P.ln_vec_stride = m->cond_rows; P.ln_vec_off = cond_off; }
    }
    mark_masked(s, P, len);
    launch_conv(s, P, EPI_STORE, "enc.qkv");
    launch_attention(s, s->qkv, L, len, s->att, B, H, T);
    P = conv_params(L.o, s->att, s->y1, B, T, 1, 0);  // y1 = x + conv_o(att)
    P.g[0].res = x;
    if (pstat) { P.ln_stat_out = s->lnst; P.ln_nmb = H / 16; }
    mark_masked(s, P, len);
    launch_conv(s, P, EPI_STORE, "enc.o");
    const float* xf = x;  // input of the FFN (after norm_layers_1)
    if (!fold) launch_ln(s, s->y1, nullptr, nullptr, x, L.g1, L.b1, len, B, H, T, 0, 0);
    // FFN (attentions.py:308-317): conv_1(pad(x*mask)) -> relu -> *mask -> conv_2(pad(.)) -> *mask
    P = conv_params(L.f1, fold ? s->y1 : x, s->ffh, B, T, 1, (K - 1) / 2);
    P.in_mask = 1; P.len = len; P.relu = 1; P.out_mask = 1;
    if (fold) { P.ln_g = L.g1; P.ln_b = L.b1; P.ln_out = s->xb; xf = s->xb; }
    if (pstat) { P.ln_stat_in = s->lnst; P.ln_nmb = H / 16; }
    mark_masked(s, P, len);
    launch_conv(s, P, EPI_STORE, "enc.ffn1");
    float* y2 = fold ? s->y1b : s->y1;
    P = conv_params(L.f2, s->ffh, y2, B, T, 1, (K - 1) / 2);
    P.in_mask = 1; P.len = len; P.out_mask = 1; P.g[0].res = xf;  // y = x + ffn(x)
    const bool lastl = i == n - 1;
    const bool to_consumer = fold && (!lastl || pend);
    if (pstat && to_consumer) { P.ln_stat_out = s->lnst; P.ln_nmb = H / 16; }
    mark_masked(s, P, len);
    launch_conv(s, P, EPI_STORE, "enc.ffn2");
    if (fold && !lastl) { prev.raw = y2; prev.g = L.g2; prev.b = L.b2; prev.stat = pstat ? s->lnst : nullptr; prev.nmb = H / 16; continue; }
    if (fold && lastl && pend) { pend->raw = y2; pend->g = L.g2; pend->b = L.b2; pend->stat = pstat ? s->lnst : nullptr; pend->nmb = H / 16; return; }
    launch_ln(s, y2, nullptr, lastl ? final_base : nullptr, (lastl && final_out) ? final_out : x, L.g2, L.b2, len, B, H,
              T, 0, lastl ? 1 : 0);
  }
  (void)F;
}

// A bounded poll of a persistent program ran out (its workgroups were not all co-resident: another process on the device, a
// partitioned GPU): the launch path stays available.  Persistent programs are switched off for the process and the host entry
// points run the call again on launches (synth_dispatch / vits_stream_open look at tl_ps_timed_out) -- the caller sees a slower
// call, not an error; asynchronous device sessions report VITS_ERR_DEVICE once.
static thread_local bool tl_ps_timed_out = false;
static int persist_timed_out() {
  const long long now = steady_ns();
  long long iv = g_ps_rearm_ns.load();
  const long long at = g_ps_rearmed_at_ns.load();
  if (!iv || !at || now - at > 10 * iv) iv = g_ps_rearm_base_ns;          // first timeout, or the last re-arm held: start over
  else if (iv < 64 * g_ps_rearm_base_ns) iv *= 2;                        // timed out again soon after a re-arm: back off
  if (iv < 1) iv = 1;
  g_ps_rearm_ns.store(iv);
  g_ps_off_until_ns.store(now + iv);
  const int n = g_ps_timeouts.fetch_add(1) + 1;
  tl_ps_timed_out = true;
  if (!getenv("VITS_QUIET"))
    fprintf(stderr, "[vits_mi355] persistent program: exchange timed out (workgroups not co-resident?) -- launch path for %.1f s, then re-armed (timeout #%d)\n",
            iv * 1e-9, n);
  return fail(VITS_ERR_DEVICE, "persistent kernel: exchange timed out (workgroups not co-resident?); off for %.1f s", iv * 1e-9);
}

static int check_err(vits_session* s) {
  int e = 0;
  HIP_TRY(hipMemcpyAsync(&e, s->d_err, sizeof(int), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  hipError_t le = hipGetLastError();
  if (le != hipSuccess) return fail(VITS_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(le));
  if (e) {
    hipMemsetAsync(s->d_err, 0, sizeof(int), s->stream);
    // an aborted persistent program leaves stale logw / x behind: every other bit may be a consequence of it (a garbage duration
    // sum raises bit 4), so the timeout is reported first -- the retry on launches surfaces the real argument errors
    if (e & PS_ERR_TIMEOUT) return persist_timed_out();
    if (e & 1) return fail(VITS_ERR_ARG, "token id out of range");
    if (e & 2) return fail(VITS_ERR_ARG, "speaker id out of range");
    if (e & 4) return fail(VITS_ERR_ARG, "T_y exceeds frame capacity");
  }
  return VITS_OK;
}

static void set_lengths(vits_session* s, const int64_t* d_len64, int* d_len32, int B, int clamp);
// ---- speaker conditioning vectors for the whole forward (one GEMV launch)
// d_len64 (optional): also converts the feed's int64 lengths to the clamped int32 array the kernels read (set_lengths folded in)
static void run_cond(vits_session* s, const int64_t* d_sid, int B, const int64_t* d_len64 = nullptr, int* d_len32 = nullptr, int clamp = 0) {
  vits_model* m = s->m;
  if (!m->use_g || !m->cond_rows) {
    if (d_len64) set_lengths(s, d_len64, d_len32, B, clamp);
    return;
  }
  hipLaunchKernelGGL(cond_gemv_kernel, dim3(cdiv(m->cond_rows, 4), B), dim3(256), 0, s->stream, m->cond_W, m->cond_b, m->emb_g,
                     d_sid, s->condv, m->cond_rows, m->hp.gin_channels, m->hp.n_speakers, s->d_err, d_len64, d_len32, clamp);
}

// ---- a2: TextEncoder.forward (models.py:317-326) -> s->x [B,H,Tx], s->stats [B,2I,Tx]
static void run_text_encoder(vits_session* s, const int64_t* d_ids, int B, int Tx, const float* d_bert = nullptr) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  const int H = hp.hidden_channels;
  if ((persist_mask() & PERSIST_ENC) && s->ps_enc.ok && B == 1 && Tx == s->Tx && (!d_bert || d_bert == s->ps_bert)) {  // one persistent kernel instead of ~35 launches
    persist_launch(s, s->ps_enc, "enc.persist", nullptr, 0.f, 0, d_ids);
    return;
  }
  hipLaunchKernelGGL(embed_kernel, dim3(cdiv(Tx, 64), 8, B), dim3(64), 0, s->stream, d_ids, s->len_x, m->emb, s->x, H, Tx,
                     hp.n_vocab, sqrtf((float)H), s->d_err);
  if (d_bert && m->bert_proj.w) {  // x = (emb * sqrt(H) + bert_proj(bert)) * mask   (BERT-conditioned flavour, synth.py:88-99)
    ConvParams Pb = conv_params(m->bert_proj, d_bert, s->x, B, Tx, 1, 0);
    Pb.g[0].res = s->x;  // every output element is read (residual) and written by the same thread: in place is safe
    Pb.out_mask = 1; Pb.len = s->len_x;
    // out_mask zeroes the projection beyond len; the embedding there is already 0
    mark_masked(s, Pb, s->len_x);
    launch_conv(s, Pb, EPI_STORE, "enc.bert_proj");
  }
  // the encoder's last LayerNorm is folded into proj's staging when both run on the small-tile kernel
  PendingLN pend;
  {
    static const float dummy = 0.f;
    ConvParams Pt = conv_params(m->enc_proj, s->x, s->stats, B, Tx, 1, 0);
    Pt.ln_g = &dummy; Pt.in_mask = 1; Pt.out_mask = 1; Pt.len = s->len_x;
    const bool can = conv_takes_c16(Pt, EPI_STORE);
    run_encoder(s, m->enc_p, s->x, s->len_x, B, Tx, m->use_g ? hp.enc_cond_layer : -1, m->cond_enc_off, nullptr, nullptr, can ? &pend : nullptr);
  }
  ConvParams P = conv_params(m->enc_proj, pend.raw ? pend.raw : s->x, s->stats, B, Tx, 1, 0);
  P.out_mask = 1; P.len = s->len_x;
  if (pend.raw) { P.ln_g = pend.g; P.ln_b = pend.b; P.ln_out = s->x; P.in_mask = 1; P.ln_stat_in = pend.stat; P.ln_nmb = pend.nmb; }  // x = encoder(...) * x_mask, also left in s->x
  mark_masked(s, P, s->len_x);
  launch_conv(s, P, EPI_STORE, "enc.proj");
}

// DDSConv.forward (modules.py:96-108) on h [B,D,T] (h already includes +g); returns the buffer holding the result
// (h itself, or s->dy after an odd number of fused layers)
static float* run_dds(vits_session* s, const DDSW& W, float* h, int B, int T) {
  const vits_hparams& hp = s->m->hp;
  const int D = hp.dp_filter_channels, K = hp.dp_kernel_size;
  int dil = 1;
  static const bool no_fuse = getenv("VITS_NO_DDS_FUSION") != nullptr;  // A/B switch for tools/ and tests
  // Single utterances / small batches only: there the three launches per layer are pure latency.  Every workgroup
  // of the fused kernel streams the whole D x D matrix and does its mat-vec on the VALU, so beyond ~one workgroup
  // per CU (B*T/8 > 256) the MFMA conv path below wins.
  if (D <= 256 && D % 64 == 0 && (long)B * T <= 2048 && !no_fuse) {  // dds_layer_kernel, ping-pong between h and s->dy
    float* src = h; float* dst = s->dy;
    for (size_t i = 0; i < W.pw.size(); ++i) {
      ProfScope ps(s, "dp.dds_layer", 2.0 * B * T * ((double)D * D + (double)D * K), "dds_layer_kernel");
      DdsParams dp{src, dst, W.sw[i], W.sb[i], W.g1[i], W.b1[i], W.wt[i], W.pw[i].bias, W.g2[i], W.b2[i], s->len_x, D, T, K, dil,
                   s->ragged ? 1 : 0};
      hipLaunchKernelGGL(dds_layer_kernel, dim3(cdiv(T, DDS_TL), B), dim3(256), 0, s->stream, dp);
      float* t = src; src = dst; dst = t;
      dil *= K;
    }
    return src;
  }
  for (size_t i = 0; i < W.pw.size(); ++i) {
    DwLnParams dp{h, s->dy, W.sw[i], W.sb[i], W.g1[i], W.b1[i], s->len_x, D, T, K, dil, s->ragged ? 1 : 0};
    hipLaunchKernelGGL(dwconv_ln_gelu_kernel, dim3(cdiv(T, LN_TL), B), dim3(256), 0, s->stream, dp);
    ConvParams P = conv_params(W.pw[i], s->dy, s->dy2, B, T, 1, 0);
    mark_masked(s, P, s->len_x);
    launch_conv(s, P, EPI_STORE, "dp.1x1");
    // x = x + gelu(LN2(y)) ; masked every layer (equivalent at valid positions)
    launch_ln(s, s->dy2, nullptr, h, h, W.g2[i], W.b2[i], s->len_x, B, D, T, 1, 1);
    dil *= K;
  }
  return h;
}

// DDSConv.forward (modules.py:96-108) on h [B,D,T] (h already includes +g) followed by the 1x1 `proj` conv that consumes it
// (dp.proj, models.py:59-63; ConvFlow.proj, modules.py:367-368): out = proj(DDSConv(h)) * mask.
// Few-column regime: every layer is ONE launch of the small-tile conv kernel whose prologue builds the layer's 1x1 input from
// the previous layer's raw tensors (finish LN2 + GELU + residual, depthwise conv, LN1, GELU: conv_small.hip.h), and `proj`
// finishes the last layer the same way -- n_layers + 1 launches of ~16 x T/16 small workgroups.  Larger problems keep one
// workgroup-per-8-columns fused layer kernel or the three-launch form, then the plain proj conv.
// pre (optional, ConvFlow): the layer input is pre->pw[c] * z[x0 row] + pre->pb[c] + cond -- folded into the first layer's
// prologue on the small-tile path, the convflow_pre_kernel launch into `h` otherwise
struct DdsPre { const float* z; int row; const float* pw; const float* pb; const float* cond; };
static void run_dds_proj(vits_session* s, const DDSW& W, float* h, const ConvW& proj, float* out, const char* proj_name, int B, int T,
                         const DdsPre* pre = nullptr) {
  const vits_hparams& hp = s->m->hp;
  const int D = hp.dp_filter_channels, K = hp.dp_kernel_size;
  static const bool no_c16 = getenv("VITS_NO_DDS_C16") != nullptr;  // A/B switch for tools/ and tests
  const long c16_cols = c16_cols_dds();
  {
    ConvParams P = conv_params(proj, h, out, B, T, 1, 0);
    P.len = s->len_x;
    int max_dil = 1;
    for (size_t i = 1; i < W.pw.size(); ++i) max_dil *= K;
    if (!no_c16 && (g_force_tile == 0 || g_force_tile == 3) && (long)B * T <= c16_cols && c16_dds_ok(P, K) && max_dil <= 9 && W.pw.size() >= 1 &&
        W.pw[0].w16) {
      const int n = (int)W.pw.size();
      float* X[2] = {s->dy, s->dq1};
      float* Y[2] = {s->dy2, s->dq2};
      const float* xin = pre ? pre->cond : h;
      int dil = 1;
      for (int i = 0; i <= n; ++i) {
        const bool fin = i == n;
        P = conv_params(fin ? proj : W.pw[i], xin, fin ? out : Y[i & 1], B, T, 1, 0);
        P.len = s->len_x;
        if (fin) P.out_mask = 1;
        mark_masked(s, P, s->len_x);
        if (i == 0 && pre) {
          P.dds_z = pre->z + (long long)pre->row * T; P.dds_z_bstride = 2LL * T; P.dds_pw = pre->pw; P.dds_pb = pre->pb;
          P.dds_xout = X[1];  // layer 1 reads the materialised layer input (x_in of layer 0) as its residual stream
        }
        if (i > 0) {
          P.dds_y2 = Y[(i - 1) & 1]; P.dds_g2 = W.g2[i - 1]; P.dds_b2 = W.b2[i - 1];
          if (!fin) P.dds_xout = X[(i - 1) & 1];
        }
        if (!fin) { P.dds_sw = W.sw[i]; P.dds_sb = W.sb[i]; P.dds_g1 = W.g1[i]; P.dds_b1 = W.b1[i]; P.dds_dil = dil; }
        launch_c16_dds(s, P, fin ? proj_name : "dp.dds_layer", 2.0 * B * T * ((double)P.Cout * D + (fin ? 0.0 : (double)D * K)));
        if (i == 0 && pre) xin = X[1];
        if (i > 0 && !fin) xin = X[(i - 1) & 1];
        dil *= K;
      }
      return;
    }
  }
  if (pre) {
    const int D2 = hp.dp_filter_channels;
    hipLaunchKernelGGL(convflow_pre_kernel, dim3(cdiv(T, 64), D2, B), dim3(64), 0, s->stream, pre->z, pre->row, pre->pw, pre->pb, pre->cond, h, D2, T);
  }
  const float* hd = run_dds(s, W, h, B, T);
  ConvParams P = conv_params(proj, hd, out, B, T, 1, 0);
  P.out_mask = 1; P.len = s->len_x;
  mark_masked(s, P, s->len_x);
  launch_conv(s, P, EPI_STORE, proj_name);
}

// ---- a6: StochasticDurationPredictor.forward(reverse=True) (models.py:56-63,93-101) -> s->logw
// defer_ea: the caller runs run_durations next on this session; the final ElementwiseAffine (logw from z) is then folded into
// durations_kernel instead of being its own launch (stage-level callers need logw itself and keep the launch)
static void run_duration(vits_session* s, const float* x, const float* d_noise, float nsw, uint64_t seed, int B, int Tx, bool defer_ea = false) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  const int D = hp.dp_filter_channels;
  if ((persist_mask() & PERSIST_SDP) && s->ps_sdp.ok && B == 1 && Tx == s->Tx) {  // one persistent kernel instead of ~21 launches (persist.hip.h)
    // the program reads the text-encoder output from the session's own buffer (stage-level callers bring theirs)
    if (x != s->x) hipMemcpyAsync(s->x, x, sizeof(float) * (size_t)hp.hidden_channels * Tx, hipMemcpyDeviceToDevice, s->stream);
    persist_launch(s, s->ps_sdp, "dp.persist", d_noise, nsw, seed);
    s->ea_pending = false;
    return;
  }
  ConvParams P = conv_params(m->dp_pre, x, s->dh, B, Tx, 1, 0);
  if (m->use_g) { P.bias_b = s->condv; P.bias_b_stride = m->cond_rows; P.bias_b_off = m->cond_dp_off; }
  mark_masked(s, P, s->len_x);
  launch_conv(s, P, EPI_STORE, "dp.pre");
  run_dds_proj(s, m->dp_dds, s->dh, m->dp_proj, s->dc, "dp.proj", B, Tx);
  hipLaunchKernelGGL(dp_init_z_kernel, dim3(cdiv(Tx, 64), 2, B), dim3(64), 0, s->stream, s->dz, d_noise, nsw, seed, Tx, s->solo ? 1 : 0, s->dv, s->item_seeds);
  int swap = 0;
  const float cst = (float)log(exp(1.0 - 1e-3) - 1.0);
  (void)cst;
  for (int k = hp.dp_n_flows - 1; k >= 1; --k) {
    swap ^= 1;  // Flip (modules.py:270-277) is a row relabel on the 2-channel z
    const ConvFlowW& c = m->cf[k];
    const DdsPre pre{s->dz, swap, c.pre_w, c.pre_b, s->dc};
    run_dds_proj(s, c.dds, s->dfh, c.proj, s->dpr, "dp.cfproj", B, Tx, &pre);
    hipLaunchKernelGGL(spline_inverse_kernel, dim3(cdiv(Tx, 64), B), dim3(64), 0, s->stream, s->dz, swap, s->dpr, c.proj.M,
                       s->len_x, Tx, hp.dp_num_bins, hp.dp_tail_bound, 1.0f / sqrtf((float)D));
  }
  swap ^= 1;
  if (defer_ea) { s->ea_pending = true; s->ea_row = swap; return; }
  hipLaunchKernelGGL(ea_logw_kernel, dim3(cdiv(Tx, 64), B), dim3(64), 0, s->stream, s->dz, swap, m->ea_m, m->ea_logs, s->len_x,
                     s->logw, Tx);
}

// ---- a10: durations / cumsum / y_lengths
static void run_durations(vits_session* s, const int* d_forced, float length_scale, int B, int Tx, int Tcap) {
  const bool ea = s->ea_pending && !d_forced;
  s->ea_pending = false;
  hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(256), 0, s->stream, s->logw, d_forced, s->len_x, length_scale, Tx, s->dur,
                     s->cum, s->len_y, s->ylen64, Tcap, s->d_err, s->dv, ea ? s->dz : (const float*)nullptr, s->ea_row,
                     (const float*)s->m->ea_m, (const float*)s->m->ea_logs);
}

// ---- a10/a11: expand prior + sample -> z_p [B,I,Ty]
static void run_expand(vits_session* s, const float* d_noise, long long noise_stride, float noise_scale, uint64_t seed,
                       float* z_p, int B, int Tx, int Ty) {
  const int I = s->m->hp.inter_channels;
  hipLaunchKernelGGL(expand_prior_kernel, dim3(cdiv(Ty, 64), cdiv(I, EXPAND_CPB), B), dim3(256), 0, s->stream, s->stats, s->cum, s->len_y,
                     d_noise, noise_stride, noise_scale, seed, z_p, I, Tx, Ty, s->solo ? 1 : 0, s->dv, s->item_seeds);
}

// ---- a12-a14: ResidualCouplingTransformersBlock.forward(reverse=True) (models.py:750-757).
// z in s->zA; result pointer returned (zA or zB).  Each Flip is folded into the next layer's
// channel-reversed read (pre conv) and the EPI_COUPLE write.
static float* run_flow(vits_session* s, int B, int Ty) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  const int H = hp.hidden_channels, I = hp.inter_channels, half = I / 2, L = hp.flow_wn_layers, K5 = hp.flow_kernel_size;
  if ((persist_mask() & PERSIST_FLOW) && s->ps_flow.ok && B == 1 && Ty == s->Ty) {  // one persistent kernel instead of ~75 launches
    persist_launch(s, s->ps_flow, "flow.persist");
    return s->zB;
  }
  float* u = s->zA;
  float* v = s->zB;
  for (int f = hp.flow_n_flows - 1; f >= 0; --f) {
    const CouplingW& C = m->flow[f];
    // h = pre(x0) * mask, x0[c] = u[I-1-c]  (models.py:375-376 after Flip)
    ConvParams P = conv_params(C.pre, u, s->fh, B, Ty, 1, 0);
    P.x_ch_off = I - 1; P.x_ch_sign = -1; P.x_bstride = (long long)I * Ty;
    P.out_mask = 1; P.len = s->len_y;
    mark_masked(s, P, s->len_y);
    P.g[0].y2 = s->x;  // second copy: the pre-transformer updates its input in place, fh stays the residual base
    launch_conv(s, P, EPI_STORE, "flow.pre");
    // h = h + pre_transformer(h * mask)  (models.py:377)
    run_encoder(s, C.enc, s->x, s->len_y, B, Ty, -1, -1, s->fh, s->fx);
    // WN (modules.py:148-176): fx is the running x.  Folded form (default): the gate outputs of all layers are kept, stacked
    // [L*H, T]; res_skip layer i < L-1 only updates x (its residual half); one [I/2 x L*H] conv = post o (sum of skip halves)
    // feeds the coupling tail.  Unfolded form (vits_debug_wn_fold(0)): res/skip epilogue per layer + post, as the reference runs it.
    const bool fold = g_wn_fold && !C.rsx.empty() && C.skip_post.w;
    const long long acts_b = (long long)(fold ? L : 1) * H * Ty;
    for (int i = 0; i < L; ++i) {
      float* acts = s->facts + (fold ? (size_t)i * H * Ty : 0);
      P = conv_params(C.in_layers[i], s->fx, acts, B, Ty, 1, (K5 - 1) / 2);
      P.Cout = H; P.H = H; P.y_bstride = acts_b;
      if (m->use_g) { P.bias_b = s->condv; P.bias_b_stride = m->cond_rows; P.bias_b_off = C.cond_off + i * 2 * H; }
      // x is masked in the reference (modules.py:171); reading it through the mask makes the K=5 window
      // independent of whatever a skipped padding tile left behind
      P.in_mask = 1; P.len = s->len_y;
      mark_masked(s, P, s->len_y);
      launch_conv(s, P, EPI_GATE, "flow.wn_in");
      if (fold) {
        if (i == L - 1) break;
        P = conv_params(C.rsx[i], acts, s->fx, B, Ty, 1, 0);  // x = (x + res_acts) * mask, in place (x is masked on entry)
        P.x_bstride = acts_b;
        P.g[0].res = s->fx; P.out_mask = 1; P.len = s->len_y;
        mark_masked(s, P, s->len_y);
        launch_conv(s, P, EPI_STORE, "flow.wn_rs");
        continue;
      }
      P = conv_params(C.rs_layers[i], s->facts, nullptr, B, Ty, 1, 0);
      P.io = s->fx; P.skip = s->fskip; P.H = H; P.first = i == 0; P.last = i == L - 1; P.len = s->len_y;
      P.y_bstride = (long long)H * Ty;
      mark_masked(s, P, s->len_y);
      launch_conv(s, P, EPI_RESSKIP, "flow.wn_rs");
    }
    // m = post(h) * mask ; x1 = (x1 - m) * mask ; cat (models.py:379-392)
    P = fold ? conv_params(C.skip_post, s->facts, nullptr, B, Ty, 1, 0) : conv_params(C.post, s->fskip, nullptr, B, Ty, 1, 0);
    P.u = u; P.io = v; P.H = half; P.len = s->len_y; P.y_bstride = (long long)I * Ty;
    mark_masked(s, P, s->len_y);
    launch_conv(s, P, EPI_COUPLE, "flow.post");
    float* t = u; u = v; v = t;
  }
  return u;
}

// ---- a15-a20: decoder (models.py:1016-1054 / 872-891).  z [B,I,Ty] (masked at staging with len_y
// when mask_in), audio -> d_audio [B, audio_bstride]
// Frames of halo kept beyond each item's length in a ragged batch.  The decoder's receptive field is < 25
// frames, so with 32 every sample below len*hop is bit-identical to the dense padded run.
#define VITS_RAGGED_HALO 32
static void set_rag(ConvParams& P, const int* rag, int in_mul, int in_add, int out_mul, int out_add) {
  P.rag = rag; P.rag_in_mul = in_mul; P.rag_in_add = in_add; P.rag_out_mul = out_mul; P.rag_out_add = out_add; P.rag_out_cap_add = 0; P.rag_tab_add = 0;
}
// What each decoder layer still has to produce BEYOND an item's end in a ragged batch, in columns of its own output (round 5).  The
// decoder has no masks: in the reference's padded batch an item's activations continue into the padding, and a valid sample depends on
// that continuation over the receptive field that is left between a layer and the waveform -- 25 frames at conv_pre, 5 columns after
// the last ResBlock.  Rounds 1-4 computed len + 32 frames at EVERY layer (10 % of the decoder's work at 330-frame items); now every
// launch carries its own limit: out = what the layers behind it need, in = what its producer made.  Walked backwards from the tail.
struct DecNeeds {
  int pre_out = 0, post_out = 0, tail_cols = 0;
  int ups_q[8] = {0};                       // polyphase launch: input positions q beyond len * rate_in
  int c1_out[8][VITS_MAX_RESD] = {{0}}, c2_out[8][VITS_MAX_RESD] = {{0}};
};
static DecNeeds decoder_needs(const vits_hparams& hp, bool continuation) {
  DecNeeds N;
  if (!continuation) return N;  // halo 0: every item is decoded as if alone (zeros beyond its own end at every stage)
  int need;  // columns the NEXT consumer wants beyond len * rate, at the current rate
  if (hp.dec_type == 0) {
    // iSTFT frame f feeds sub-band samples [f hop, f hop + n_fft); PQMF synthesis reaches (taps / 2) / subbands sub-band samples ahead
    need = (hp.istft_n_fft + hp.istft_hop - 1) / hp.istft_hop + ((hp.pqmf_taps / 2 + hp.subbands - 1) / hp.subbands + hp.istft_hop - 1) / hp.istft_hop + 2;
  } else {
    need = 0;
  }
  N.tail_cols = need;              // the tail reads conv_post columns 0 .. len * rate + need INCLUSIVE ...
  N.post_out = need + 1;           // ... so conv_post makes need + 1 of them beyond len * rate (it has T + 1 columns: the reflection pad)
  need += 4;                       // conv_post, 7 taps (pad 4 with the reflection column, 3 without)
  for (int i = hp.n_ups - 1; i >= 0; --i) {
    for (int d = hp.n_resd - 1; d >= 0; --d) {
      int h2 = 0, h1 = 0;
      for (int j = 0; j < hp.n_resk; ++j) {
        const int k = hp.res_kernels[j];
        h2 = std::max(h2, (k - 1) / 2);
        h1 = std::max(h1, (k - 1) * hp.res_dilations[j][d] / 2);
      }
      N.c2_out[i][d] = need; need += h2;
      N.c1_out[i][d] = need; need += h1;
    }
    const int u = hp.up_rates[i], taps = (hp.up_kernels[i] + u - 1) / u;
    N.ups_q[i] = (need + u - 1) / u + 1;  // output column c = u q + phase
    need = N.ups_q[i] + taps / 2 + 2;      // input positions a polyphase output reads: q -+ taps / 2 (+ slack for the phase shifts)
  }
  N.pre_out = need;
  return N;
}
// rag_halo > 0 (with ragged): the reference's padded-batch continuation -- every valid sample equals the dense padded run (the per-layer
// limits above; the value only has to be >= the receptive field and is otherwise unused); 0 decodes every item as if it were alone
// (zeros beyond its own end at every stage), which is what a batch of independent utterances wants (solo batches, the StableTTS path).
// VITS_RAG_UNIFORM=1: the round-4 form (len + rag_halo frames at every layer), the A/B reference.
static void run_decoder(vits_session* s, const float* z, bool mask_in, int B, int Ty, float* d_audio, long long audio_bstride,
                        float* d_mb, bool ragged = false, int rag_halo = -1) {
  vits_model* m = s->m;
  if (rag_halo < 0) rag_halo = m->rag_halo;  // default: the reference's padded-batch continuation over the receptive field
  const vits_hparams& hp = m->hp;
  int C = hp.dec_initial_channel, T = Ty;
  const int* rag = nullptr;
  const int* rag_tail = nullptr;
  int rate = 1;  // columns per frame at the current stage
  static const bool no_ragged_env = getenv("VITS_NO_RAGGED") != nullptr;
  static const bool uniform = getenv("VITS_RAG_UNIFORM") && atoi(getenv("VITS_RAG_UNIFORM")) != 0;
  int final_rate = 1;
  for (int i = 0; i < hp.n_ups; ++i) final_rate *= hp.up_rates[i];
  const bool layered = rag_halo > 0 && !uniform;
  const DecNeeds ND = decoder_needs(hp, layered);
  if (ragged && (B > 1 || s->rag_b1) && !no_ragged_env) {
    // uniform form: rag = len + halo, the tail may read (len + halo) * rate columns; layered form: rag = len, every launch adds its own need
    hipLaunchKernelGGL(ragged_len_kernel, dim3(cdiv(B + 1, 64)), dim3(64), 0, s->stream, s->len_y, s->len_rag, s->len_tail, B, Ty,
                       layered ? 0 : rag_halo, final_rate, layered ? ND.tail_cols : rag_halo * final_rate);
    rag = s->len_rag;
    rag_tail = s->len_tail;
  }
  float* cur = s->dec_bufs[0];
  ConvParams P = conv_params(m->conv_pre, z, cur, B, Ty, 1, 3);
  if (mask_in) { P.in_mask = 1; P.len = s->len_y; }  // (z * y_mask) models.py:1703
  if (m->cond_dec_off >= 0) { P.bias_b = s->condv; P.bias_b_stride = m->cond_rows; P.bias_b_off = m->cond_dec_off; }  // + cond(g)
  set_rag(P, rag, 1, 0, 1, ND.pre_out);
  launch_conv(s, P, EPI_STORE, "dec.conv_pre");
  int prod_add = ND.pre_out;  // columns (at the current rate) the tensor about to be consumed has beyond len * rate
  const float* in1 = cur; const float* in2 = nullptr; const float* in3 = nullptr;
  float in_scale = 1.f;
  for (int i = 0; i < hp.n_ups; ++i) {
    const UpW& U = m->ups[i];
    float** set = &s->dec_bufs[1 + 7 * (i & 1)];
    float* y = set[0];
    const int Co = U.cout, To = T * U.u;
    // x = leaky_relu(x, 0.1); x = ups[i](x)  (models.py:1027-1028), polyphase
    memset(&P, 0, sizeof P);
    P.n_groups = 1;
    P.g[0].x = in1; P.g[0].x2 = in2; P.g[0].x3 = in3; P.g[0].w = U.w.w; P.g[0].wb = U.w.wb; P.g[0].bias = U.w.bias; P.g[0].y = y;
    P.g[0].K = U.taps; P.g[0].dil = 1; P.g[0].pad_l = U.pad_l; P.g[0].n_sg = U.w.n_sg;
    P.B = B; P.Cin = C; P.x_ch_sign = 1; P.x_bstride = (long long)C * T; P.Tin = T; P.Tin_stride = T;
    P.M = U.w.Mpad; P.Cout = U.w.M; P.Tout = T; P.Tout_stride = To; P.y_bstride = (long long)Co * To;
    P.in_slope = 0.1f; P.in_scale = in_scale;
    P.ups_u = U.u; P.ups_cout = Co;
    for (int r = 0; r < U.u; ++r) P.ups_shift[r] = U.shift[r];
    set_rag(P, rag, rate, prod_add, rate, ND.ups_q[i]);  // polyphase: output "columns" are input positions q
    launch_conv(s, P, EPI_STORE, "dec.ups", U.halo);
    C = Co; T = To; rate *= U.u;
    prod_add = ND.ups_q[i] * U.u;
    // MRF: 3 ResBlock1 chains in grouped launches (modules.py:210-223)
    const int nk = hp.n_resk;
    // (Round 4 experiment, removed: the three chains as three branches of the captured graph -- one stream each, forked and joined
    //  with events -- so that a chain's per-launch fixed cost runs under the other chains' matrix work: c2 0.856 -> 0.921 ms, 19 -> 43
    //  graph nodes; the cross-queue dependencies cost more than the overlap returns.  profiles/r4_decoder_split.txt)
    for (int d = 0; d < hp.n_resd; ++d) {
      memset(&P, 0, sizeof P);
      P.n_groups = nk;
      for (int j = 0; j < nk; ++j) {  // xt = c1(leaky_relu(x))
        const ResBlockW& R = m->rb[(size_t)i * nk + j];
        P.g[j].x = d == 0 ? y : set[4 + j];
        P.g[j].w = R.c1[d].w; P.g[j].wb = R.c1[d].wb; P.g[j].bias = R.c1[d].bias; P.g[j].y = set[1 + j];
        P.g[j].K = R.K; P.g[j].dil = R.dil[d]; P.g[j].pad_l = (R.K - 1) * R.dil[d] / 2; P.g[j].n_sg = R.c1[d].n_sg;
      }
      P.B = B; P.Cin = C; P.x_ch_sign = 1; P.x_bstride = (long long)C * T; P.Tin = T; P.Tin_stride = T;
      P.M = m->rb[(size_t)i * nk].c1[d].Mpad; P.Cout = C; P.Tout = T; P.Tout_stride = T; P.y_bstride = (long long)C * T;
      P.in_slope = 0.1f; P.in_scale = 1.f;
      set_rag(P, rag, rate, prod_add, rate, ND.c1_out[i][d]);
      P.rag_tab_add = ND.c1_out[i][0];  // one compact tile map for the six launches of the stage (its widest limit)
      launch_conv(s, P, EPI_STORE, "dec.res_c1");
      for (int j = 0; j < nk; ++j) {  // x = c2(leaky_relu(xt)) + x
        const ResBlockW& R = m->rb[(size_t)i * nk + j];
        P.g[j].x = set[1 + j];
        P.g[j].w = R.c2[d].w; P.g[j].wb = R.c2[d].wb; P.g[j].bias = R.c2[d].bias; P.g[j].y = set[4 + j];
        P.g[j].res = d == 0 ? y : set[4 + j];
        P.g[j].K = R.K; P.g[j].dil = 1; P.g[j].pad_l = (R.K - 1) / 2; P.g[j].n_sg = R.c2[d].n_sg;
      }
      set_rag(P, rag, rate, ND.c1_out[i][d], rate, ND.c2_out[i][d]);
      P.rag_tab_add = ND.c1_out[i][0];
      launch_conv(s, P, EPI_STORE, "dec.res_c2");
      prod_add = ND.c2_out[i][d];
    }
    in1 = set[4]; in2 = nk > 1 ? set[5] : nullptr; in3 = nk > 2 ? set[6] : nullptr;
    in_scale = 1.0f / (float)nk;  // x = xs / num_kernels (models.py:1036), folded into the next staging
  }
  float* post = s->dec_bufs[15];
  float* mb = d_mb ? d_mb : s->dec_bufs[16];
  if (hp.dec_type == 0) {
    // leaky_relu(0.01) -> ReflectionPad1d((1,0)) -> subband_conv_post (models.py:1038-1040)
    const int Tp = T + 1, Pc = m->conv_post.M;
    memset(&P, 0, sizeof P);
    P.n_groups = 1;
    P.g[0].x = in1; P.g[0].x2 = in2; P.g[0].x3 = in3; P.g[0].w = m->conv_post.w; P.g[0].y = post;
    P.g[0].K = 7; P.g[0].dil = 1; P.g[0].pad_l = 4; P.g[0].n_sg = m->conv_post.n_sg;
    P.B = B; P.Cin = C; P.x_ch_sign = 1; P.x_bstride = (long long)C * T; P.Tin = T; P.Tin_stride = T;
    P.M = m->conv_post.Mpad; P.Cout = Pc; P.Tout = Tp; P.Tout_stride = Tp; P.y_bstride = (long long)Pc * Tp;
    P.in_slope = 0.01f; P.in_scale = in_scale; P.reflect = 1;
    set_rag(P, rag, rate, prod_add, rate, layered ? ND.post_out : 1);
    P.rag_out_cap_add = 1;  // T + 1 output columns
    launch_conv(s, P, EPI_STORE, "dec.conv_post");
    const int S = hp.subbands, N = hp.istft_n_fft, hop = hp.istft_hop, Tm = T * hop;
    if (g_tail_impl == 0) {  // one launch: exp/sin, iSTFT and PQMF through LDS
      ProfScope ps(s, "istft_pqmf", 0, "istft_pqmf_kernel");
      TailParams tp{post, m->istft_basis, m->pqmf, mb, d_audio, S, N, hop, Tp, Tm, hp.pqmf_taps, audio_bstride, rag_tail, hop};  // (rag_tail: conv_post columns that exist)
      const int HM = (hp.pqmf_taps / 2 + S - 1) / S + 1, nsub = TAIL_MB + 2 * HM, FR = (nsub + N) / hop + 2;
      const size_t lds = ((size_t)2 * S * (N / 2 + 1) * FR + (size_t)S * nsub + (size_t)(N + 2) * N + (size_t)S * (hp.pqmf_taps + 1)) * sizeof(float);
      hipLaunchKernelGGL(istft_pqmf_kernel, dim3(cdiv(Tm, TAIL_MB), B), dim3(256), lds, s->stream, tp);
    } else {
      {
        ProfScope ps(s, "istft", 0, "istft_kernel");
        hipLaunchKernelGGL(istft_kernel, dim3(cdiv(Tm, 256), S, B), dim3(256), 0, s->stream, post, m->istft_basis, mb, S, N, hop, Tp, Tm,
                           rag_tail, hop);
      }
      {
        ProfScope ps(s, "pqmf", 0, "pqmf_synthesis_kernel");
        hipLaunchKernelGGL(pqmf_synthesis_kernel, dim3(cdiv(Tm * S, 256), B), dim3(256), 0, s->stream, mb, m->pqmf, d_audio, S,
                           hp.pqmf_taps, Tm, audio_bstride, rag_tail, hop * S);
      }
    }
  } else {
    memset(&P, 0, sizeof P);
    P.n_groups = 1;
    P.g[0].x = in1; P.g[0].x2 = in2; P.g[0].x3 = in3; P.g[0].w = m->conv_post.w; P.g[0].bias = m->conv_post.bias; P.g[0].y = post;
    P.g[0].K = 7; P.g[0].dil = 1; P.g[0].pad_l = 3; P.g[0].n_sg = m->conv_post.n_sg;
    P.B = B; P.Cin = C; P.x_ch_sign = 1; P.x_bstride = (long long)C * T; P.Tin = T; P.Tin_stride = T;
    P.M = m->conv_post.Mpad; P.Cout = 1; P.Tout = T; P.Tout_stride = T; P.y_bstride = T;
    P.in_slope = 0.01f; P.in_scale = in_scale;
    set_rag(P, rag, rate, prod_add, rate, layered ? ND.post_out : 0);
    launch_conv(s, P, EPI_STORE, "dec.conv_post");
    hipLaunchKernelGGL(tanh_copy_kernel, dim3(cdiv(T, 256), B), dim3(256), 0, s->stream, post, d_audio, T, (long long)T, audio_bstride, rag_tail, 1);
  }
}

// ---- helpers for the host-buffer stage entry points
struct HostStage {
  vits_model* m; vits_session* s = nullptr; std::vector<void*> tmp;
  PersistScope pscope;  // (declared last-constructed / first-destroyed relative to the stream sync in ~HostStage: see below)
  explicit HostStage(vits_model* m_) : m(m_), pscope(m_ ? m_->device : -1) {}
  ~HostStage() {
    if (s) {
      hipStreamSynchronize(s->stream);
      if (!tmp.empty()) {  // the staging area overflowed during this call: grow it once, for the next one
        size_t want = s->stage_used + (s->stage_used >> 2) + (1 << 20);
        if (s->stage) hipFree(s->stage);
        s->stage = nullptr; s->stage_bytes = 0;
        void* p = nullptr;
        if (hipMalloc(&p, want) == hipSuccess) { s->stage = static_cast<char*>(p); s->stage_bytes = want; }
      }
      s->stage_used = 0;
      pool_release(m, s);
    }
    for (void* p : tmp) hipFree(p);
  }
  // bump allocation from the session's staging area; falls back to hipMalloc (freed at the end of the call) when full
  void* raw_alloc(size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, 256);
    const size_t off = s->stage_used;
    s->stage_used += bytes;  // also counts overflow, so the destructor knows how much this call needed
    if (off + bytes <= s->stage_bytes) return s->stage + off;
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    tmp.push_back(d);
    return d;
  }
  template <typename T> T* to_dev(const T* h, size_t n) {
    if (!h) return nullptr;
    void* d = raw_alloc(n * sizeof(T));
    if (!d) return nullptr;
    hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s->stream);
    return static_cast<T*>(d);
  }
  template <typename T> T* dev_alloc(size_t n) { return static_cast<T*>(raw_alloc(n * sizeof(T))); }
};

// roles: the persistent programs the caller will launch on this layout (pooled sessions are shared by callers with different needs:
// the mask is part of the layout key, session_reserve)
static int begin_stage(HostStage& hs, int B, int Tx, int Ty, int roles = 7) {
  hipError_t e = hipSetDevice(hs.m->device);
  if (e != hipSuccess) return fail(VITS_ERR_DEVICE, "hipSetDevice failed: %s", hipGetErrorString(e));
  TRY(pool_acquire(hs.m, &hs.s));
  hs.s->ps_roles = roles;
  TRY(session_reserve(hs.s, B, Tx, Ty));
  return VITS_OK;
}

static void set_lengths(vits_session* s, const int64_t* d_len64, int* d_len32, int B, int clamp) {
  hipLaunchKernelGGL(lengths_to_i32_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s->stream, d_len64, d_len32, B, clamp);
}


static void forward_device(vits_session* s, const int64_t* d_ids, const int64_t* d_len, int B, int Tx, const float* scales,
                           const int64_t* d_sid, const int32_t* d_forced, int Ty, uint64_t seed, float* d_audio, int64_t cap) {
  static const bool no_ragged = getenv("VITS_NO_RAGGED") != nullptr;  // A/B switch for tools/
  s->ragged = B > 1 && !no_ragged;
  s->tile_keys.clear();
  run_cond(s, d_sid, B, d_len, s->len_x, Tx);
  const bool with_sdp = !d_forced || s->sdp_always;  // (logw unused when durations are pinned)
  float* z;
  if (persist_mask() == (PERSIST_SDP | PERSIST_ENC | PERSIST_FLOW) && B == 1 && Tx == s->Tx && Ty == s->Ty && s->ps_full[with_sdp].ok) {
    // text encoder .. flow of a single utterance as ONE persistent launch (the frame capacity T_y is the caller's)
    persist_launch(s, s->ps_full[with_sdp], "acoustic.persist", nullptr, scales[2], seed, d_ids, d_forced, scales[1], scales[0]);
    s->ea_pending = false;
    z = s->zB;
  } else {
    run_text_encoder(s, d_ids, B, Tx);
    if (with_sdp) run_duration(s, s->x, nullptr, scales[2], seed, B, Tx, true);
    run_durations(s, d_forced, scales[1], B, Tx, Ty);
    run_expand(s, nullptr, Ty, scales[0], seed, s->zA, B, Tx, Ty);
    z = run_flow(s, B, Ty);
  }
  run_decoder(s, z, true, B, Ty, d_audio, cap, nullptr, true);
  s->ragged = false;
}


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
    else if (g_ps_spin_limit > 0) hipMemcpy(m->ps_dbg, &g_ps_spin_limit, sizeof(int), hipMemcpyHostToDevice);
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
static int g_fast_path = 1;
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
  PersistScope pscope(B == 1 ? m->device : -1);  // a single utterance takes the persistent stages when no other call on this device holds them
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
void vits_debug_persist_rearm_ms(int ms) {
  g_ps_rearm_base_ns = (ms > 0 ? (long long)ms : (getenv("VITS_PERSIST_REARM_MS") ? atoll(getenv("VITS_PERSIST_REARM_MS")) : 1000)) * 1000000LL;
}
// State of the persistent programs of this process (a server logs it; tests assert the re-arm).
int vits_persist_state(vits_model* m, vits_persist_info* out) {
  if (!out) return fail(VITS_ERR_ARG, "null out");
  memset(out, 0, sizeof *out);
  out->configured_mask = g_persist;
  const long long until = g_ps_off_until_ns.load(), now = steady_ns();
  out->active_mask = (until && now < until) ? 0 : g_persist;
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
  g_ps_spin_limit = limit > 0 ? limit : 0;
  std::lock_guard<std::mutex> g(g_models_mu);
  for (vits_model* m : g_models) {
    hipSetDevice(m->device);
    hipDeviceSynchronize();
    hipMemcpy(m->ps_dbg, &g_ps_spin_limit, sizeof(int), hipMemcpyHostToDevice);
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
