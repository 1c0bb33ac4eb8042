// engine_model.hip.h -- weights: the conv / encoder / flow / decoder weight records, vits_model and the VITSW001 loader (repacking into MFMA fragment order).
// Part of the ONE translation unit engine.hip (included there, in order; not a standalone header): split out in round 6 so that the
// planner / launch selection / stages / host paths can be read on their own.
#pragma once
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

