// stts.hip.h — StableTTS / Matcha ("multistream") inference on the MI355X, C ABI of include/stts_mi355.h.
// Included at the end of engine.hip: the contractions (1x1 / k=3 convs, fused qkv, cond_proj, long-skip convs) run on
// the shared fp32-MFMA conv kernels, attention on relpos_attention_mfma_kernel without relative tables (RoPE is
// applied to q/k beforehand), the adaLN / FiLM / Euler glue on the small kernels at the end of kernels_misc.hip.h.
// Reference citations are relative to /root/reference/training/stabletts/matcha/.
#include "../../include/stts_mi355.h"

struct DitW {  // DiTConVBlock (models/components/diffusion_transformer.py:82-118)
  ConvW qkv, o, c1, c2;
  float *a0w = nullptr, *a0b = nullptr, *a2w = nullptr, *a2b = nullptr;  // adaLN_modulation Linear(G,H), Linear(H,6H)
};

struct SttsFront;
static void stts_front_free(SttsFront* f);
struct stts_model {
  vits_model base;  // blob table, device allocations and the session pool (tget / upload / make_conv / pool_acquire)
  stts_hparams hp;
  vits_model* vocoder = nullptr;
  float *emb = nullptr, *pemb = nullptr, *spk_emb = nullptr, *dur_spk_emb = nullptr, *fake_speaker = nullptr, *fake_content = nullptr;
  float *zero_vec = nullptr;  // zeros: speaker vector when n_spks <= 1, relative-position tables of the attention kernel
  ConvW bert_proj, enc_proj, in_proj, final_proj, cp0, cp2, cp4;
  std::vector<DitW> enc, dec;
  std::vector<ConvW> lsc;
  float *t0w = nullptr, *t0b = nullptr, *t2w = nullptr, *t2b = nullptr;
  std::vector<float*> film_w, film_b;
  std::map<int, SttsFront*> fronts;  // graph-replayed fast path of stts_synthesize: cached contexts by T_x bucket (under base.pool_mu)
  uint64_t use_clock = 0;
};

static DitW load_dit(vits_model* b, const char* p, int H, int F, int K, int G) {
  DitW d;
  char nm[200];
  const float* wq = tget(b, 3, H, H, 1, "%s.attn.conv_q.weight", p);
  const float* wk = tget(b, 3, H, H, 1, "%s.attn.conv_k.weight", p);
  const float* wv = tget(b, 3, H, H, 1, "%s.attn.conv_v.weight", p);
  const float* bq = tget(b, 1, H, -1, -1, "%s.attn.conv_q.bias", p);
  const float* bk = tget(b, 1, H, -1, -1, "%s.attn.conv_k.bias", p);
  const float* bv = tget(b, 1, H, -1, -1, "%s.attn.conv_v.bias", p);
  if (b->missing) return d;
  std::vector<float> bias((size_t)3 * H);
  memcpy(bias.data(), bq, sizeof(float) * H); memcpy(bias.data() + H, bk, sizeof(float) * H); memcpy(bias.data() + 2 * H, bv, sizeof(float) * H);
  d.qkv = make_conv(b, 3 * H, H, 1, bias.data(), [&](int r, int ci, int) { return (r < H ? wq : (r < 2 * H ? wk : wv))[(size_t)(r % H) * H + ci]; });
  snprintf(nm, sizeof nm, "%s.attn.conv_o", p); d.o = conv_from(b, nm, H, H, 1, true);
  snprintf(nm, sizeof nm, "%s.mlp.conv_1", p); d.c1 = conv_from(b, nm, F, H, K, true);
  snprintf(nm, sizeof nm, "%s.mlp.conv_2", p); d.c2 = conv_from(b, nm, H, F, K, true);
  d.a0w = upload(b, tget(b, 2, H, G, -1, "%s.adaLN_modulation.0.weight", p), (size_t)H * G);
  d.a0b = upload(b, tget(b, 1, H, -1, -1, "%s.adaLN_modulation.0.bias", p), H);
  d.a2w = upload(b, tget(b, 2, 6 * H, H, -1, "%s.adaLN_modulation.2.weight", p), (size_t)6 * H * H);
  d.a2b = upload(b, tget(b, 1, 6 * H, -1, -1, "%s.adaLN_modulation.2.bias", p), (size_t)6 * H);
  return d;
}

static int stts_load(stts_model* m) {
  vits_model* b = &m->base;
  const stts_hparams& hp = m->hp;
  const int H = hp.enc_hidden, Hd = hp.dec_hidden, Fd = hp.dec_filter, G = hp.spk_emb_dim, NF = hp.n_feats;
  if (hp.emb_dim + 4 * hp.punc_dim + hp.bert_proj_dim != H) return fail(VITS_ERR_UNSUPPORTED, "stream widths do not add up to enc_hidden");
  if (H % 32 || Hd % 32 || NF % CONV_CI_T || hp.bert_dim % CONV_CI_T || hp.bert_proj_dim % 32)
    return fail(VITS_ERR_UNSUPPORTED, "channel counts must be multiples of 32 (mel/bert: 16)");
  if (H > LN_MAXV * LN_CG || Hd > LN_MAXV * LN_CG) return fail(VITS_ERR_UNSUPPORTED, "LayerNorm width > %d", LN_MAXV * LN_CG);
  const int dke = H / hp.enc_heads, dkd = Hd / hp.dec_heads;
  if ((dke != 32 && dke != 64 && dke != 96) || (dkd != 32 && dkd != 64 && dkd != 96)) return fail(VITS_ERR_UNSUPPORTED, "head dim not in {32,64,96}");
  if (hp.dec_layers % 2) return fail(VITS_ERR_UNSUPPORTED, "long skip connections need an even number of decoder layers");
  char nm[200];
  m->emb = upload(b, tget(b, 2, hp.n_vocab, hp.emb_dim, -1, "encoder.emb.weight"), (size_t)hp.n_vocab * hp.emb_dim);
  m->pemb = upload(b, tget(b, 2, hp.n_vocab, hp.punc_dim, -1, "encoder.punc_emb.weight"), (size_t)hp.n_vocab * hp.punc_dim);
  {  // bert_proj: Dropout + Linear(768, 32) applied per column == a 1x1 conv (text_encoder.py:108,129)
    const float* w = tget(b, 2, hp.bert_proj_dim, hp.bert_dim, -1, "encoder.bert_proj.1.weight");
    const float* bb = tget(b, 1, hp.bert_proj_dim, -1, -1, "encoder.bert_proj.1.bias");
    if (!b->missing) m->bert_proj = make_conv(b, hp.bert_proj_dim, hp.bert_dim, 1, bb, [&](int r, int ci, int) { return w[(size_t)r * hp.bert_dim + ci]; });
  }
  if (hp.n_spks > 1) {
    m->spk_emb = upload(b, tget(b, 2, hp.n_spks, G, -1, "spk_emb.weight"), (size_t)hp.n_spks * G);
    m->dur_spk_emb = upload(b, tget(b, 2, hp.n_spks, G, -1, "dur_spk_emb.weight"), (size_t)hp.n_spks * G);
  }
  m->fake_speaker = upload(b, tget(b, 2, 1, G, -1, "fake_speaker"), G);
  m->fake_content = upload(b, tget(b, 3, 1, H, 1, "fake_content"), H);
  {
    std::vector<float> z((size_t)(G > 9 * 96 ? G : 9 * 96), 0.f);
    m->zero_vec = upload(b, z.data(), z.size());
  }
  for (int i = 0; i < hp.enc_layers && !b->missing; ++i) {
    snprintf(nm, sizeof nm, "encoder.dp_encoder.encoder.%d", i);
    m->enc.push_back(load_dit(b, nm, H, hp.enc_filter, hp.enc_kernel, G));
  }
  m->enc_proj = conv_from(b, "encoder.dp_encoder.proj", hp.dp_out, H, 1, true);
  const char* e = "decoder.estimator";
  m->t0w = upload(b, tget(b, 2, Fd, Hd, -1, "%s.time_mlp.layer.0.weight", e), (size_t)Fd * Hd);
  m->t0b = upload(b, tget(b, 1, Fd, -1, -1, "%s.time_mlp.layer.0.bias", e), Fd);
  m->t2w = upload(b, tget(b, 2, Hd, Fd, -1, "%s.time_mlp.layer.2.weight", e), (size_t)Hd * Fd);
  m->t2b = upload(b, tget(b, 1, Hd, -1, -1, "%s.time_mlp.layer.2.bias", e), Hd);
  snprintf(nm, sizeof nm, "%s.in_proj", e); m->in_proj = conv_from(b, nm, Hd, NF + Hd, 1, true);
  snprintf(nm, sizeof nm, "%s.final_proj", e); m->final_proj = conv_from(b, nm, NF, Hd, 1, true);
  snprintf(nm, sizeof nm, "%s.cond_proj.0", e); m->cp0 = conv_from(b, nm, Fd, H, hp.dec_kernel, true);
  snprintf(nm, sizeof nm, "%s.cond_proj.2", e); m->cp2 = conv_from(b, nm, Fd, Fd, hp.dec_kernel, true);
  snprintf(nm, sizeof nm, "%s.cond_proj.4", e); m->cp4 = conv_from(b, nm, Hd, Fd, hp.dec_kernel, true);
  for (int i = 0; i < hp.dec_layers && !b->missing; ++i) {
    m->film_w.push_back(upload(b, tget(b, 3, 2 * Hd, Hd, 1, "%s.blocks.%d.time_fusion.film.weight", e, i), (size_t)2 * Hd * Hd));
    m->film_b.push_back(upload(b, tget(b, 1, 2 * Hd, -1, -1, "%s.blocks.%d.time_fusion.film.bias", e, i), (size_t)2 * Hd));
    snprintf(nm, sizeof nm, "%s.blocks.%d.block", e, i);
    m->dec.push_back(load_dit(b, nm, Hd, Fd, hp.dec_kernel, G));
  }
  for (int j = 0; j < hp.dec_layers / 2 && !b->missing; ++j) {
    snprintf(nm, sizeof nm, "%s.lsc_layers.%d", e, j);
    m->lsc.push_back(conv_from(b, nm, Hd, 2 * Hd, hp.dec_kernel, true));
  }
  return b->missing ? VITS_ERR_BLOB : VITS_OK;
}

// ---- workspace: a bump arena on a pooled session of &m->base (stream + error word + events)
static int stts_arena(vits_session* s, size_t bytes) {
  s->arena_used = 0;
  if (g_poison && s->arena) hipMemsetAsync(s->arena, 0xFF, s->arena_bytes, s->stream);  // vits_debug_poison_workspace: NaN
  if (bytes <= s->arena_bytes) return VITS_OK;
  if (s->arena) { hipStreamSynchronize(s->stream); hipFree(s->arena); s->arena = nullptr; s->arena_bytes = 0; }
  void* p = nullptr;
  const size_t want = bytes + bytes / 4 + 4096;
  if (hipMalloc(&p, want) != hipSuccess) return fail(VITS_ERR_NOMEM, "workspace hipMalloc of %zu bytes failed", want);
  s->arena = static_cast<char*>(p);
  s->arena_bytes = want;
  if (g_poison) hipMemsetAsync(s->arena, 0xFF, s->arena_bytes, s->stream);
  return VITS_OK;
}

static void stts_attention(vits_session* s, const stts_model* m, const float* qkv, const int* len, float* out, int B, int H, int T, int nh) {
  const int dk = H / nh;
  hipLaunchKernelGGL(rope_kernel, dim3(cdiv(T, 64), nh * (dk / 4), B * 2), dim3(64), 0, s->stream, const_cast<float*>(qkv), H, T, nh, dk);
  // null relative tables: the kernels skip the banded relative-position terms (F.scaled_dot_product_attention has none);
  // short sequences take the 16-query kernel like the VITS encoders (engine.hip launch_attention_raw)
  launch_attention_raw(s, qkv, nullptr, nullptr, len, out, B, H, T, nh, 4);
}

struct DitScratch { float *hn, *qkv, *att, *ffh; };

// Batches of ragged items (s->ragged): column tiles that start at or beyond len[b] are not dispatched.  Safe for every
// conv of this path: a tile boundary is a multiple of 32, so it is never below the item's length rounded up to 4 (the
// extent a single-utterance run has), and everything beyond is read through masks or zero-padding selects only.
static void stts_skip(vits_session* s, ConvParams& P, const int* len) {
  if (s->ragged && len) { P.skip_len = 1; if (!P.len) P.len = len; }
}

static void stts_ln_mod(vits_session* s, const float* x, float* y, const float* shift, const float* scale, int mod_stride, int B, int H, int T,
                        const float* film = nullptr, float* film_out = nullptr, const int* len = nullptr) {
  ProfScope ps(s, "layernorm", 0, "layernorm_c_kernel");
  LNParams P{x, nullptr, nullptr, y, scale, shift, len, H, T, 0, 0, (s->ragged && len) ? 1 : 0, mod_stride, 1e-5f, film, film_out};
  launch_layernorm(s->stream, P, B);
}

// DiTConVBlock.forward (diffusion_transformer.py:99-118) on h [B,H,T] in place; h must already be masked.
// mod [B][6H]: shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp | gate_mlp
// film != nullptr: the block input is FiLM(h_in) * mask (DitWrapper.forward), produced inside the first LayerNorm launch
// and written to h (h_in is left untouched)
static void stts_dit_block(vits_session* s, const stts_model* m, const DitW& W, float* h, const float* mod, const int* len, int B,
                           int H, int F, int nh, int K, int T, const DitScratch& sc, const char* tag, const float* film = nullptr,
                           const float* h_in = nullptr) {
  char nm[64];
  if (film) stts_ln_mod(s, h_in, sc.hn, mod, mod + H, 6 * H, B, H, T, film, h, len);
  else stts_ln_mod(s, h, sc.hn, mod, mod + H, 6 * H, B, H, T, nullptr, nullptr, len);
  ConvParams P = conv_params(W.qkv, sc.hn, sc.qkv, B, T, 1, 0);
  stts_skip(s, P, len);
  snprintf(nm, sizeof nm, "%s.qkv", tag); launch_conv(s, P, EPI_STORE, nm);
  stts_attention(s, m, sc.qkv, len, sc.att, B, H, T, nh);
  P = conv_params(W.o, sc.att, h, B, T, 1, 0);  // x = x + gate_msa * attn(...) * x_mask
  P.out_mask = 1; P.len = len; P.scale_b = mod; P.scale_b_stride = 6 * H; P.scale_b_off = 2 * H; P.g[0].res = h;
  stts_skip(s, P, len);
  snprintf(nm, sizeof nm, "%s.o", tag); launch_conv(s, P, EPI_STORE, nm);
  stts_ln_mod(s, h, sc.hn, mod + 3 * H, mod + 4 * H, 6 * H, B, H, T, nullptr, nullptr, len);
  P = conv_params(W.c1, sc.hn, sc.ffh, B, T, 1, K / 2);  // FFN: conv_1(x * mask) -> SiLU (diffusion_transformer.py:25-27)
  P.in_mask = 1; P.len = len; P.relu = 2;
  stts_skip(s, P, len);
  snprintf(nm, sizeof nm, "%s.ffn1", tag); launch_conv(s, P, EPI_STORE, nm);
  P = conv_params(W.c2, sc.ffh, h, B, T, 1, K / 2);  // conv_2(. * mask) * mask ; x = x + gate_mlp * mlp
  P.in_mask = 1; P.out_mask = 1; P.len = len; P.scale_b = mod; P.scale_b_stride = 6 * H; P.scale_b_off = 5 * H; P.g[0].res = h;
  stts_skip(s, P, len);
  snprintf(nm, sizeof nm, "%s.ffn2", tag); launch_conv(s, P, EPI_STORE, nm);
}

static void stts_gemv(vits_session* s, const float* W, const float* bias, const float* x, int x_stride, float* y, int y_stride, int rows,
                      int cols, int n, int act) {
  hipLaunchKernelGGL(gemv_rows_kernel, dim3(cdiv(rows, 4), n), dim3(256), 0, s->stream, W, bias, x, x_stride, y, y_stride, rows, cols, act);
}

// adaLN_modulation(c) for a stack of blocks: mods[i][n][6H], c [n][G] (diffusion_transformer.py:93-97,110)
static void stts_modulations(vits_session* s, const std::vector<DitW>& blocks, const float* c, int n, int H, int G, float* tmp, float* mods) {
  for (size_t i = 0; i < blocks.size(); ++i) {
    stts_gemv(s, blocks[i].a0w, blocks[i].a0b, c, G, tmp, H, H, G, n, 1);
    stts_gemv(s, blocks[i].a2w, blocks[i].a2b, tmp, H, mods + i * (size_t)n * 6 * H, 6 * H, 6 * H, H, n, 0);
  }
}

// ---- TextEncoder.forward (text_encoder.py:111-139) on device: d_x [B,H,T] (unmasked concat), d_mu [B,dp_out,T]
struct SttsEncBufs { float *h, *cvec, *tmp, *mods; DitScratch sc; int* len; };
static size_t stts_enc_bytes(const stts_hparams& hp, int B, int T) {
  const size_t H = hp.enc_hidden, F = hp.enc_filter;
  return ((size_t)B * T * (H * 2 + 3 * H + H + F) + (size_t)B * (hp.spk_emb_dim + H + hp.enc_layers * 6 * H)) * sizeof(float) + 64 * 1024;
}
// speaker ids: sid_host (eager calls: one D2D copy per item) or d_sid (graph capture: gathered on the device)
static void stts_run_encoder(vits_session* s, const stts_model* m, const int64_t* d_ids, const int* d_len, const int64_t* sid_host, int B,
                             int T, const float* d_bert, float* d_x, float* d_mu, const int64_t* d_sid = nullptr) {
  const stts_hparams& hp = m->hp;
  const int H = hp.enc_hidden, F = hp.enc_filter, G = hp.spk_emb_dim, E = hp.emb_dim, Pd = hp.punc_dim, NE = E + 4 * Pd;
  SttsEncBufs w;
  w.h = bump<float>(s, (size_t)B * H * T); w.sc.hn = bump<float>(s, (size_t)B * H * T); w.sc.qkv = bump<float>(s, (size_t)B * 3 * H * T);
  w.sc.att = bump<float>(s, (size_t)B * H * T); w.sc.ffh = bump<float>(s, (size_t)B * F * T);
  w.cvec = bump<float>(s, (size_t)B * G); w.tmp = bump<float>(s, (size_t)B * H); w.mods = bump<float>(s, (size_t)hp.enc_layers * B * 6 * H);
  hipLaunchKernelGGL(stts_embed_kernel, dim3(cdiv(T, 64), NE, B), dim3(64), 0, s->stream, d_ids, m->emb, m->pemb, d_x, H, E, Pd, T, hp.n_vocab,
                     sqrtf((float)E), sqrtf((float)Pd), s->d_err);
  ConvParams P = conv_params(m->bert_proj, d_bert, d_x + (size_t)NE * T, B, T, 1, 0);
  P.y_bstride = (long long)H * T;  // rows [NE, H) of the concatenated x
  launch_conv(s, P, EPI_STORE, "enc.bert_proj");
  hipMemcpyAsync(w.h, d_x, sizeof(float) * (size_t)B * H * T, hipMemcpyDeviceToDevice, s->stream);
  hipLaunchKernelGGL(mask_rows_kernel, dim3(cdiv(T, 64), H, B), dim3(64), 0, s->stream, w.h, d_len, H, T);
  if (d_sid && hp.n_spks > 1)
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(G, 64), B), dim3(64), 0, s->stream, w.cvec, m->dur_spk_emb, d_sid, G, hp.n_spks, s->d_err);
  else
  for (int b = 0; b < B; ++b)  // dur_spks = dur_spk_emb(sid) (matcha_tts.py:139)
    hipMemcpyAsync(w.cvec + (size_t)b * G, (hp.n_spks > 1 && sid_host) ? m->dur_spk_emb + (size_t)sid_host[b] * G : m->zero_vec, sizeof(float) * G,
                   hipMemcpyDeviceToDevice, s->stream);
  stts_modulations(s, m->enc, w.cvec, B, H, G, w.tmp, w.mods);
  for (int i = 0; i < hp.enc_layers; ++i)
    stts_dit_block(s, m, m->enc[i], w.h, w.mods + (size_t)i * B * 6 * H, d_len, B, H, F, hp.enc_heads, hp.enc_kernel, T, w.sc, "enc");
  P = conv_params(m->enc_proj, w.h, d_mu, B, T, 1, 0);  // mu_x = proj(x) * x_mask (text_encoder.py:45)
  P.out_mask = 1; P.len = d_len;
  launch_conv(s, P, EPI_STORE, "enc.proj");
}

// ---- estimator state for one synthesis: everything that does not depend on the Euler step is computed once
struct SttsEst {
  int nb, T, n_steps;
  float *cat, *h, *h2, *dphi, *film, *mods, *lsc[3 + 8], *a1, *a2;
  DitScratch sc;
  int* len;   // [nb] valid frames per item (masks)
  int* lenT;  // [nb] frames per item rounded up to a multiple of 4: where a single-utterance run's tensors END, i.e. where the
              // unmasked convs (cond_proj, long-skip) see zero padding; equals T for one utterance
  std::vector<float> host_sinus;
  bool tables_ready = false;  // the per-step time tables (sinus, t1, temb, film: functions of n_steps only) are already resident
};
static size_t stts_est_bytes(const stts_hparams& hp, int nb, int T, int n_steps) {
  const size_t H = hp.dec_hidden, F = hp.dec_filter, NF = hp.n_feats, NL = hp.dec_layers;
  size_t fl = (size_t)nb * T * ((NF + H) + H + NF + H + 3 * H + H + F + (NL / 2) * 2 * H + 2 * F + hp.enc_hidden);
  fl += (size_t)n_steps * (H + F + H + NL * 2 * H) + (size_t)nb * (hp.spk_emb_dim + H + NL * 6 * H);
  return fl * sizeof(float) + 256 * 1024;
}
// c [nb][G] device, mu [nb][enc_hidden][T] device (item 1 = fake content when nb == 2), tvals host [n_steps]
static void stts_est_setup(vits_session* s, const stts_model* m, SttsEst& E, const float* d_c, const float* d_mu, const int* d_len, const float* tvals,
                           const int* d_lenT = nullptr) {
  const stts_hparams& hp = m->hp;
  const int H = hp.dec_hidden, F = hp.dec_filter, NF = hp.n_feats, NL = hp.dec_layers, G = hp.spk_emb_dim, K = hp.dec_kernel;
  const int nb = E.nb, T = E.T, n = E.n_steps;
  E.len = const_cast<int*>(d_len);
  E.lenT = const_cast<int*>(d_lenT);
  E.cat = bump<float>(s, (size_t)nb * (NF + H) * T); E.h = bump<float>(s, (size_t)nb * H * T); E.dphi = bump<float>(s, (size_t)nb * NF * T);
  E.sc.hn = bump<float>(s, (size_t)nb * H * T); E.sc.qkv = bump<float>(s, (size_t)nb * 3 * H * T); E.sc.att = bump<float>(s, (size_t)nb * H * T);
  E.sc.ffh = bump<float>(s, (size_t)nb * F * T);
  for (int j = 0; j < NL / 2; ++j) E.lsc[j] = bump<float>(s, (size_t)nb * H * T);
  E.h2 = bump<float>(s, (size_t)nb * H * T);
  E.a1 = bump<float>(s, (size_t)nb * F * T); E.a2 = bump<float>(s, (size_t)nb * F * T);
  // time embeddings of every step: SinusoidalPosEmb(H)(t, scale=1000) on the host (n_steps * H values), TimestepEmbedding
  // and the FiLM projections of all blocks as GEMVs over the n_steps columns (components/decoder.py:35-62,15-33)
  float* sinus = bump<float>(s, (size_t)n * H); float* t1 = bump<float>(s, (size_t)n * F); float* temb = bump<float>(s, (size_t)n * H);
  E.film = bump<float>(s, (size_t)NL * n * 2 * H);
  if (!E.tables_ready) {
    std::vector<float>& hs = E.host_sinus;  // must outlive the asynchronous copy below: owned by the call's SttsEst
    hs.resize((size_t)n * H);
    const int half = H / 2;
    const float lg = logf(10000.0f) / (float)(half - 1);
    for (int k = 0; k < n; ++k)
      for (int j = 0; j < half; ++j) {
        const float a = 1000.0f * tvals[k] * expf((float)j * -lg);
        hs[(size_t)k * H + j] = sinf(a); hs[(size_t)k * H + half + j] = cosf(a);
      }
    hipMemcpyAsync(sinus, hs.data(), sizeof(float) * hs.size(), hipMemcpyHostToDevice, s->stream);
    stts_gemv(s, m->t0w, m->t0b, sinus, H, t1, F, F, H, n, 1);
    stts_gemv(s, m->t2w, m->t2b, t1, F, temb, H, H, F, n, 0);
    for (int i = 0; i < NL; ++i) stts_gemv(s, m->film_w[i], m->film_b[i], temb, H, E.film + (size_t)i * n * 2 * H, 2 * H, 2 * H, H, n, 0);
  }
  float* tmp = bump<float>(s, (size_t)nb * H);
  E.mods = bump<float>(s, (size_t)NL * nb * 6 * H);
  stts_modulations(s, m->dec, d_c, nb, H, G, tmp, E.mods);
  // mu = cond_proj(mu): conv -> SiLU -> conv -> SiLU -> conv, no masks (decoder.py:82-88,121); lands in rows [NF, NF+H) of cat
  // in a batch every item must see zeros beyond ITS OWN padded length (in_mask with lenT) to equal its single-utterance run
  ConvParams P = conv_params(m->cp0, d_mu, E.a1, nb, T, 1, K / 2); P.relu = 2;
  if (E.lenT) { P.in_mask = 1; P.len = E.lenT; }
  stts_skip(s, P, E.lenT);
  launch_conv(s, P, EPI_STORE, "cfm.cond_proj");
  P = conv_params(m->cp2, E.a1, E.a2, nb, T, 1, K / 2); P.relu = 2;
  if (E.lenT) { P.in_mask = 1; P.len = E.lenT; }
  stts_skip(s, P, E.lenT);
  launch_conv(s, P, EPI_STORE, "cfm.cond_proj");
  P = conv_params(m->cp4, E.a2, E.cat + (size_t)NF * T, nb, T, 1, K / 2); P.y_bstride = (long long)(NF + H) * T;
  if (E.lenT) { P.in_mask = 1; P.len = E.lenT; }
  stts_skip(s, P, E.lenT);
  launch_conv(s, P, EPI_STORE, "cfm.cond_proj");
}
// one Decoder.forward (decoder.py:105-138) over the nb batch items; state x = rows [0,NF) of E.cat; result in E.dphi
static void stts_est_step(vits_session* s, const stts_model* m, SttsEst& E, int step) {
  const stts_hparams& hp = m->hp;
  const int H = hp.dec_hidden, F = hp.dec_filter, NL = hp.dec_layers, K = hp.dec_kernel;
  const int nb = E.nb, T = E.T, n = E.n_steps;
  ConvParams P = conv_params(m->in_proj, E.cat, E.h, nb, T, 1, 0);
  stts_skip(s, P, E.len);
  launch_conv(s, P, EPI_STORE, "cfm.in_proj");
  // Buffer choreography without copies: FiLM runs out of place, so for the first NL/2 blocks the tensor it READ is left
  // untouched and IS the long-skip output (lsc_outputs.append(x)); the second half consumes them in reverse through the
  // channel-split conv (cat((x, skip), dim=1) never exists).
  float* cur = E.h;                 // in_proj output
  float* skips[8];
  int sp = 0, nfree = 0;
  float* freeb[12];
  for (int j = 0; j < NL / 2; ++j) freeb[nfree++] = E.lsc[j];
  freeb[nfree++] = E.h2;
  for (int idx = 0; idx < NL; ++idx) {
    if (idx >= NL / 2) {  // x = lsc_layers[idx - NL/2](cat((x, lsc_outputs.pop()), dim=1)): channels [0,H) from x, [H,2H) from the skip
      float* skip = skips[--sp];
      float* out = freeb[--nfree];
      P = conv_params(m->lsc[idx - NL / 2], cur, out, nb, T, 1, K / 2);
      P.x_bstride = (long long)H * T;  // each of the two inputs is a dense [nb, H, T] tensor
      P.g[0].x2 = skip; P.x_split = H;
      if (E.lenT) { P.in_mask = 1; P.len = E.lenT; }
      stts_skip(s, P, E.lenT);
      launch_conv(s, P, EPI_STORE, "cfm.lsc");
      freeb[nfree++] = cur; freeb[nfree++] = skip;
      cur = out;
    }
    float* fo = freeb[--nfree];  // FiLM(cur) * mask lands here (inside the block's first LayerNorm launch)
    stts_dit_block(s, m, m->dec[idx], fo, E.mods + (size_t)idx * nb * 6 * H, E.len, nb, H, F, hp.dec_heads, K, T, E.sc, "cfm",
                   E.film + ((size_t)idx * n + step) * 2 * H, cur);
    if (idx < NL / 2) skips[sp++] = cur; else freeb[nfree++] = cur;
    cur = fo;
  }
  P = conv_params(m->final_proj, cur, E.dphi, nb, T, 1, 0);  // final_proj(x * mask) * mask
  P.in_mask = 1; P.out_mask = 1; P.len = E.len;
  stts_skip(s, P, E.len);
  launch_conv(s, P, EPI_STORE, "cfm.final_proj");
}

// t_span of BASECFM.forward (flow_matching.py:53-54) and the (t, dt) sequence of solve_euler (:84-106), in fp32 like torch
static void stts_time_grid(int n, std::vector<float>& tv, std::vector<float>& dtv) {
  std::vector<float> ts(n + 1);
  for (int i = 0; i <= n; ++i) {
    const float step = 1.0f / (float)n;
    const float lin = i < (n + 1) / 2 ? step * (float)i : 1.0f - step * (float)(n - i);  // torch.linspace
    ts[i] = 1.0f - cosf(lin * 0.5f * 3.14159265358979323846f);
  }
  tv.resize(n); dtv.resize(n);
  float t = ts[0], dt = ts[1] - ts[0];
  for (int k = 1; k <= n; ++k) {
    tv[k - 1] = t; dtv[k - 1] = dt;
    t = t + dt;
    if (k < n) dt = ts[k + 1] - t;
  }
}

struct SttsCall {  // pooled session + temporaries of one entry-point call
  stts_model* m; vits_session* s = nullptr; std::vector<void*> tmp;
  explicit SttsCall(stts_model* m_) : m(m_) {}
  ~SttsCall() {
    if (s) {
      hipStreamSynchronize(s->stream);
      if (!tmp.empty()) {  // the staging area overflowed during this call: grow it once, for the next one
        const size_t want = s->stage_used + (s->stage_used >> 2) + (1 << 20);
        if (s->stage) hipFree(s->stage);
        s->stage = nullptr; s->stage_bytes = 0;
        void* p = nullptr;
        if (hipMalloc(&p, want) == hipSuccess) { s->stage = static_cast<char*>(p); s->stage_bytes = want; }
      }
      s->stage_used = 0;
      pool_release(&m->base, s);
    }
    for (void* p : tmp) hipFree(p);
  }
  int begin(size_t arena_bytes) {
    hipError_t e = hipSetDevice(m->base.device);
    if (e != hipSuccess) return fail(VITS_ERR_DEVICE, "hipSetDevice failed: %s", hipGetErrorString(e));
    TRY(pool_acquire(&m->base, &s));
    if (g_poison && s->stage) hipMemsetAsync(s->stage, 0xFF, s->stage_bytes, s->stream);
    return stts_arena(s, arena_bytes);
  }
  // inputs / outputs of the call: bump-allocated from the pooled session's staging area (no hipMalloc / hipFree in the
  // steady state); falls back to hipMalloc, freed at the end of the call, when the area is too small
  template <typename T> T* dev(size_t n) {
    const size_t bytes = align_up((n ? n : 1) * sizeof(T), 256);
    const size_t off = s->stage_used;
    s->stage_used += bytes;
    if (off + bytes <= s->stage_bytes) return reinterpret_cast<T*>(s->stage + off);
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    if (g_poison) hipMemsetAsync(d, 0xFF, bytes, s->stream);
    tmp.push_back(d);
    return static_cast<T*>(d);
  }
  template <typename T> T* up(const T* h, size_t n) {
    if (!h) return nullptr;
    T* d = dev<T>(n);
    if (d) hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s->stream);
    return d;
  }
};

static int stts_check_sid(const stts_model* m, const int64_t* sid, int B) {
  if (m->hp.n_spks <= 1) return VITS_OK;
  if (!sid) return fail(VITS_ERR_ARG, "sid required");
  for (int b = 0; b < B; ++b) if (sid[b] < 0 || sid[b] >= m->hp.n_spks) return fail(VITS_ERR_ARG, "speaker id out of range");
  return VITS_OK;
}

// matcha_tts.py:144-158 on the host (T_x * dp_out values; the host round trip is needed for T_y anyway)
static void stts_durations_host(const stts_hparams& hp, const float* mu_dp, int B, int T, float length_scale, const float* pde, int32_t* dur, int64_t* ylen) {
  const int K = hp.dp_out;
  for (int b = 0; b < B; ++b) {
    int64_t tot = 0;
    for (int t = 0; t < T; ++t) {
      float lw = 0.f;
      for (int k = 0; k < K; ++k) lw += 1.0f / (1.0f + expf(-mu_dp[((size_t)b * K + k) * T + t]));
      if (pde && pde[(size_t)b * T + t] != 0.f) lw = pde[(size_t)b * T + t];
      float w = rintf(lw * length_scale);  // torch.round: half to even
      if (w < 1.f) w = 1.f;
      dur[(size_t)b * T + t] = (int32_t)w;
      tot += (int64_t)w;
    }
    ylen[b] = tot;
  }
}

// BASECFM.forward + solve_euler with guidance (flow_matching.py:36-108,177-189) for one utterance, all on device.
// d_mu2 [nb][enc_hidden][T] (item 1 pre-filled with fake_content), result: state rows of E.cat item 0
// B utterances; E.nb = B (no guidance) or 2B (items [B,2B) = the unconditional branch of items [0,B))
static void stts_run_cfm(vits_session* s, const stts_model* m, SttsEst& E, const float* d_c, const float* d_mu2, const int* d_len, const float* d_noise,
                         long long nstride, float temperature, uint64_t seed, int B = 1, const int* d_lenT = nullptr, const SttsDev* dv = nullptr,
                         bool setup_only = false, const unsigned long long* d_item_seeds = nullptr) {
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats, H = hp.dec_hidden, T = E.T, cfg = E.nb > B ? 1 : 0;
  std::vector<float> tv, dtv;
  stts_time_grid(E.n_steps, tv, dtv);
  stts_est_setup(s, m, E, d_c, d_mu2, d_len, tv.data(), d_lenT);
  if (setup_only) return;
  hipLaunchKernelGGL(cfm_init_kernel, dim3(cdiv(T, 64), NF, B), dim3(64), 0, s->stream, E.cat, (long long)(NF + H) * T, d_noise, nstride, temperature, seed, NF, T, B, cfg, dv, d_item_seeds);
  for (int k = 0; k < E.n_steps; ++k) {
    stts_est_step(s, m, E, k);
    hipLaunchKernelGGL(cfm_euler_kernel, dim3(cdiv(T, 64), NF, B), dim3(64), 0, s->stream, E.cat, (long long)(NF + H) * T, E.dphi, dtv[k], hp.guidance_scale, NF, T, B, cfg);
  }
}

extern "C" {

const char* stts_last_error(void) { return g_err; }

int stts_create(const void* blob, size_t bytes, vits_model* vocoder, int device, stts_model** out) {
  if (!blob || !out || bytes < 16 + sizeof(stts_hparams)) return fail(VITS_ERR_ARG, "bad blob argument");
  const unsigned char* p = static_cast<const unsigned char*>(blob);
  if (memcmp(p, "STTSW001", 8) != 0) return fail(VITS_ERR_BLOB, "bad magic");
  uint32_t hb;
  memcpy(&hb, p + 8, 4);
  if (hb != sizeof(stts_hparams)) return fail(VITS_ERR_BLOB, "hparams size %u != %zu", hb, sizeof(stts_hparams));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VITS_ERR_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(VITS_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
  if (vocoder && (vocoder->device != device || vocoder->acoustic || vocoder->hp.inter_channels != reinterpret_cast<const stts_hparams*>(p + 12)->n_feats))
    return fail(VITS_ERR_ARG, "vocoder must be a vocoder-only model with n_feats input channels on the same device");
  HIP_TRY(hipSetDevice(device));
  stts_model* m = new stts_model();
  memcpy(&m->hp, p + 12, sizeof(stts_hparams));
  if (m->hp.abi_version != STTS_ABI_VERSION) { delete m; return fail(VITS_ERR_BLOB, "abi version mismatch"); }
  vits_model* b = &m->base;
  b->device = device; b->acoustic = false;
  b->blob = p; b->blob_bytes = bytes;
  memcpy(&b->n_entries, p + 12 + hb, 4);
  b->entries = reinterpret_cast<const vits_blob_entry*>(p + 16 + hb);
  int rc = VITS_OK;
  if (16 + hb + (size_t)b->n_entries * sizeof(vits_blob_entry) > bytes) rc = fail(VITS_ERR_BLOB, "truncated table");
  for (uint32_t i = 0; rc == VITS_OK && i < b->n_entries; ++i)
    if (b->entries[i].offset > bytes || b->entries[i].nelem > (bytes - b->entries[i].offset) / 4) rc = fail(VITS_ERR_BLOB, "truncated data");  // overflow-safe
  if (rc == VITS_OK) rc = stts_load(m);
  b->blob = nullptr; b->entries = nullptr;
  if (rc != VITS_OK) { for (void* a : b->allocs) hipFree(a); delete m; return rc; }
  m->vocoder = vocoder;
  hipDeviceSynchronize();
  *out = m;
  return VITS_OK;
}

void stts_destroy(stts_model* m) {
  if (!m) return;
  hipSetDevice(m->base.device);
  for (auto& kv : m->fronts) stts_front_free(kv.second);
  for (vits_session* s : m->base.pool) session_free(s);
  for (void* a : m->base.allocs) hipFree(a);
  delete m;
}

int stts_get_hparams(const stts_model* m, stts_hparams* out) {
  if (!m || !out) return fail(VITS_ERR_ARG, "null argument");
  *out = m->hp;
  return VITS_OK;
}

int stts_stage_encoder(stts_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T, const int64_t* sid, const float* bert,
                       float* x, float* mu_dp) {
  if (!m || !ids || !lengths || !x || !mu_dp || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  for (int b = 0; b < B; ++b) if (lengths[b] < 0 || lengths[b] > T) return fail(VITS_ERR_ARG, "length out of range");
  TRY(stts_check_sid(m, sid, B));
  const stts_hparams& hp = m->hp;
  SttsCall c(m);
  TRY(c.begin(stts_enc_bytes(hp, B, T)));
  vits_session* s = c.s;
  int64_t* d_ids = c.up(ids, (size_t)B * 5 * T);
  std::vector<int> l32(B);
  for (int b = 0; b < B; ++b) l32[b] = (int)lengths[b];
  int* d_len = c.up(l32.data(), B);
  float* d_bert = bert ? c.up(bert, (size_t)B * hp.bert_dim * T) : c.dev<float>((size_t)B * hp.bert_dim * T);
  float* d_x = c.dev<float>((size_t)B * hp.enc_hidden * T);
  float* d_mu = c.dev<float>((size_t)B * hp.dp_out * T);
  if (!d_ids || !d_len || !d_bert || !d_x || !d_mu) return fail(VITS_ERR_NOMEM, "device alloc failed");
  if (!bert) HIP_TRY(hipMemsetAsync(d_bert, 0, sizeof(float) * (size_t)B * hp.bert_dim * T, s->stream));
  stts_run_encoder(s, m, d_ids, d_len, sid, B, T, d_bert, d_x, d_mu);
  HIP_TRY(hipMemcpyAsync(x, d_x, sizeof(float) * (size_t)B * hp.enc_hidden * T, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(mu_dp, d_mu, sizeof(float) * (size_t)B * hp.dp_out * T, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

int stts_stage_durations(stts_model* m, const float* mu_dp, int32_t B, int32_t T, float length_scale, const float* pde, int32_t* durations,
                         int64_t* y_lengths) {
  if (!m || !mu_dp || !durations || !y_lengths || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  stts_durations_host(m->hp, mu_dp, B, T, length_scale, pde, durations, y_lengths);
  return VITS_OK;
}

int stts_stage_estimator(stts_model* m, const float* x, const float* mu, const int64_t* y_lengths, int32_t B, int32_t T, float t, const float* c,
                         float* out) {
  if (!m || !x || !mu || !y_lengths || !c || !out || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats, H = hp.dec_hidden;
  SttsCall call(m);
  TRY(call.begin(stts_est_bytes(hp, B, T, 1)));
  vits_session* s = call.s;
  std::vector<int> l32(B);
  for (int b = 0; b < B; ++b) l32[b] = (int)y_lengths[b];
  int* d_len = call.up(l32.data(), B);
  float* d_c = call.up(c, (size_t)B * hp.spk_emb_dim);
  float* d_mu = call.up(mu, (size_t)B * hp.enc_hidden * T);
  float* d_x = call.up(x, (size_t)B * NF * T);
  if (!d_len || !d_c || !d_mu || !d_x) return fail(VITS_ERR_NOMEM, "device alloc failed");
  SttsEst E; E.nb = B; E.T = T; E.n_steps = 1;
  stts_est_setup(s, m, E, d_c, d_mu, d_len, &t);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(T, 256), NF, B), dim3(256), 0, s->stream, d_x, (long long)NF * T, E.cat, (long long)(NF + H) * T, T);
  stts_est_step(s, m, E, 0);
  HIP_TRY(hipMemcpyAsync(out, E.dphi, sizeof(float) * (size_t)B * NF * T, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

int stts_stage_cfm(stts_model* m, const float* mu_y, int64_t y_length, int32_t T, int64_t sid, const float* noise, float temperature,
                   int32_t n_timesteps, float* out) {
  if (!m || !mu_y || !noise || !out || T <= 0 || y_length < 0 || y_length > T) return fail(VITS_ERR_ARG, "bad argument");
  TRY(stts_check_sid(m, &sid, 1));
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats, CC = hp.enc_hidden, G = hp.spk_emb_dim, H = hp.dec_hidden;
  const int n = n_timesteps > 0 ? n_timesteps : hp.n_timesteps, nb = hp.guidance_scale > 0.f ? 2 : 1;
  SttsCall call(m);
  TRY(call.begin(stts_est_bytes(hp, nb, T, n)));
  vits_session* s = call.s;
  const int l2[2] = {(int)y_length, (int)y_length};
  int* d_len = call.up(l2, 2);
  float* d_noise = call.up(noise, (size_t)NF * T);
  float* d_mu2 = call.dev<float>((size_t)nb * CC * T);
  float* d_c = call.dev<float>((size_t)nb * G);
  if (!d_len || !d_noise || !d_mu2 || !d_c) return fail(VITS_ERR_NOMEM, "device alloc failed");
  HIP_TRY(hipMemcpyAsync(d_mu2, mu_y, sizeof(float) * (size_t)CC * T, hipMemcpyHostToDevice, s->stream));
  HIP_TRY(hipMemcpyAsync(d_c, hp.n_spks > 1 ? m->spk_emb + (size_t)sid * G : m->zero_vec, sizeof(float) * G, hipMemcpyDeviceToDevice, s->stream));
  if (nb == 2) {  // fake_content.repeat(1, 1, T), fake_speaker (flow_matching.py:183-185)
    hipLaunchKernelGGL(fill_rows_kernel, dim3(cdiv(T, 64), CC), dim3(64), 0, s->stream, d_mu2 + (size_t)CC * T, m->fake_content, T, CC);
    HIP_TRY(hipMemcpyAsync(d_c + G, m->fake_speaker, sizeof(float) * G, hipMemcpyDeviceToDevice, s->stream));
  }
  SttsEst E; E.nb = nb; E.T = T; E.n_steps = n;
  stts_run_cfm(s, m, E, d_c, d_mu2, d_len, d_noise, T, temperature, 0);
  HIP_TRY(hipMemcpyAsync(out, E.cat, sizeof(float) * (size_t)NF * T, hipMemcpyDeviceToHost, s->stream));  // state rows of item 0
  (void)H;
  return check_err(s);
}

}  // extern "C"

// ---- fast path of stts_synthesize -----------------------------------------------------------------------------------
// One utterance costs ~310 launches; issued eagerly the host needs longer than the GPU (m2: 5.6 ms wall for ~3 ms of kernels, and
// 8 ms on a slower host).  Same recipe as the VITS entry point (engine.hip "fast path"): per-call scalars in a device block
// (SttsDev), inputs through a pinned block copied by a memcpy node, shapes bucketed (T_x to a multiple of 8, frames to a
// multiple of 32) and two captured graphs around the one host round trip the path needs (durations -> T_y):
//   FRONT (T_x bucket): H2D, text encoder, duration logits D2H.       BACK (frame bucket, n_steps): H2D, expand, speaker
//   vectors, cond_proj, n Euler steps of the estimator (CFG twins as a batch of 2), mel, vocoder (bucketed single utterance:
//   zeros beyond the item's own end at every stage), clamp, D2H.
// Bucketing is exact by the argument of stts_synthesize_batch: a padded item equals its own exact-size call because every
// masked op masks per item and the unmasked convs read zeros beyond the item's own length rounded up to 4 (lenT).  The time
// tables of the estimator (functions of n_steps only) are computed once when a BACK context is created.  Calls that inject a
// noise tensor (parity tests) and VITS_NO_FASTPATH / vits_debug_fast_path(0) take the eager path.
struct SttsBack {
  vits_session* s = nullptr;   // estimator workspace (arena, error word); launches go onto the front's stream
  vits_session* sv = nullptr;  // vocoder workspace (a session of the vocoder model owned by this context)
  int TB = 0, n = 0, nb = 0;
  SttsEst E;
  char* buf = nullptr;         // d_mu2 | d_pau | d_c | d_mel | d_audio
  float *d_mu2 = nullptr, *d_pau = nullptr, *d_c = nullptr, *d_mel = nullptr, *d_audio = nullptr;
  float* out_h = nullptr;      // pinned: audio [TB * hop] | mel [NF * TB]
  size_t audio_elems = 0;
  hipGraphExec_t g[2] = {nullptr, nullptr};  // [with vocoder]
  uint64_t last_use = 0;
};
struct SttsFront {
  vits_session* s = nullptr;   // stream + encoder workspace
  int TxB = 0;
  char *io_h = nullptr, *io_d = nullptr;  // pinned host mirror / device copy of the per-call inputs (region A: phase 1, region B: phase 2)
  size_t io_bytes = 0, a_bytes = 0, o_len = 0, o_sid = 0, o_bert = 0, o_dev = 0, o_cum = 0, o_pde = 0, o_len2 = 0, o_lenT = 0;
  float *d_x = nullptr, *d_mu = nullptr;
  float* mu_h = nullptr;       // pinned [dp_out * TxB] + one int error word behind it
  hipGraphExec_t g1 = nullptr;
  std::map<std::pair<int, int>, SttsBack*> backs;  // (frame bucket, n_steps)
  uint64_t last_use = 0, clock = 0;
};

static void stts_back_free(SttsBack* b) {
  if (!b) return;
  for (int i = 0; i < 2; ++i) if (b->g[i]) hipGraphExecDestroy(b->g[i]);
  if (b->buf) hipFree(b->buf);
  if (b->out_h) hipHostFree(b->out_h);
  if (b->sv) session_free(b->sv);
  if (b->s) { b->s->stream = nullptr; session_free(b->s); }
  delete b;
}
static void stts_front_free(SttsFront* f) {
  if (!f) return;
  if (f->s && f->s->stream) hipStreamSynchronize(f->s->stream);
  for (auto& kv : f->backs) stts_back_free(kv.second);
  if (f->g1) hipGraphExecDestroy(f->g1);
  if (f->io_h) hipHostFree(f->io_h);
  if (f->io_d) hipFree(f->io_d);
  if (f->d_x) hipFree(f->d_x);
  if (f->d_mu) hipFree(f->d_mu);
  if (f->mu_h) hipHostFree(f->mu_h);
  if (f->s) session_free(f->s);
  delete f;
}

static int stts_front_acquire(stts_model* m, int TxB, SttsFront** out) {
  {
    std::lock_guard<std::mutex> g(m->base.pool_mu);
    auto it = m->fronts.find(TxB);
    if (it != m->fronts.end()) { *out = it->second; m->fronts.erase(it); return VITS_OK; }
  }
  const stts_hparams& hp = m->hp;
  SttsFront* f = new SttsFront();
  f->TxB = TxB;
  int rc = session_new(&m->base, &f->s);
  if (rc == VITS_OK) rc = stts_arena(f->s, stts_enc_bytes(hp, 1, TxB));
  // region A: [ids int64 [5, TxB] | len int | sid int64 | bert float [bert_dim, TxB]]   region B: [SttsDev | cum int [TxB] | pde float [TxB] | len [2] | lenT [2]]
  f->o_len = align_up(sizeof(int64_t) * 5 * (size_t)TxB, 64);
  f->o_sid = f->o_len + 64;
  f->o_bert = f->o_sid + 64;
  f->a_bytes = f->o_bert + align_up(sizeof(float) * (size_t)hp.bert_dim * TxB, 256);
  f->o_dev = f->a_bytes;
  f->o_cum = f->o_dev + align_up(sizeof(SttsDev), 64);
  f->o_pde = f->o_cum + align_up(sizeof(int) * (size_t)TxB, 64);
  f->o_len2 = f->o_pde + align_up(sizeof(float) * (size_t)TxB, 64);
  f->o_lenT = f->o_len2 + 64;
  f->io_bytes = f->o_lenT + 64;
  const size_t nmu = (size_t)hp.dp_out * TxB;
  if (rc == VITS_OK && (hipHostMalloc((void**)&f->io_h, f->io_bytes) != hipSuccess || hipMalloc((void**)&f->io_d, f->io_bytes) != hipSuccess ||
                        hipMalloc((void**)&f->d_x, sizeof(float) * (size_t)hp.enc_hidden * TxB) != hipSuccess ||
                        hipMalloc((void**)&f->d_mu, sizeof(float) * nmu) != hipSuccess ||
                        hipHostMalloc((void**)&f->mu_h, sizeof(float) * nmu + 64) != hipSuccess))
    rc = fail(VITS_ERR_NOMEM, "fast-path staging buffers");
  if (rc == VITS_OK) {
    memset(f->io_h, 0, f->io_bytes);
    if (hipMemsetAsync(f->io_d, 0, f->io_bytes, f->s->stream) != hipSuccess) rc = fail(VITS_ERR_DEVICE, "memset failed");
  }
  if (rc != VITS_OK) { stts_front_free(f); return rc; }
  *out = f;
  return VITS_OK;
}
static void stts_front_release(stts_model* m, SttsFront* f) {
  std::vector<SttsFront*> evict;
  {
    std::lock_guard<std::mutex> g(m->base.pool_mu);
    f->last_use = ++m->use_clock;
    auto it = m->fronts.find(f->TxB);
    if (it != m->fronts.end()) { evict.push_back(it->second); m->fronts.erase(it); }  // a concurrent call built the same bucket: keep the newer
    m->fronts[f->TxB] = f;
    while (m->fronts.size() > 16) {
      auto lru = m->fronts.begin();
      for (auto jt = m->fronts.begin(); jt != m->fronts.end(); ++jt) if (jt->second->last_use < lru->second->last_use) lru = jt;
      evict.push_back(lru->second);
      m->fronts.erase(lru);
    }
  }
  for (SttsFront* e : evict) stts_front_free(e);
}

// frees every idle front context (with its back contexts): the answer to a failed allocation on the request path (the cache is
// bounded by count -- 16 text buckets x 4 frame buckets -- not by bytes)
static void stts_fronts_evict_all(stts_model* m) {
  std::vector<SttsFront*> evict;
  {
    std::lock_guard<std::mutex> g(m->base.pool_mu);
    for (auto& kv : m->fronts) evict.push_back(kv.second);
    m->fronts.clear();
  }
  for (SttsFront* e : evict) stts_front_free(e);
  (void)hipGetLastError();
}

static int stts_back_get(stts_model* m, SttsFront* F, int TB, int n, SttsBack** out) {
  const auto key = std::make_pair(TB, n);
  auto it = F->backs.find(key);
  if (it != F->backs.end()) { it->second->last_use = ++F->clock; *out = it->second; return VITS_OK; }
  if (F->backs.size() >= 4) {
    auto lru = F->backs.begin();
    for (auto jt = F->backs.begin(); jt != F->backs.end(); ++jt) if (jt->second->last_use < lru->second->last_use) lru = jt;
    hipStreamSynchronize(F->s->stream);
    stts_back_free(lru->second);
    F->backs.erase(lru);
  }
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats, CC = hp.enc_hidden, G = hp.spk_emb_dim;
  SttsBack* b = new SttsBack();
  b->TB = TB; b->n = n; b->nb = hp.guidance_scale > 0.f ? 2 : 1;
  b->s = new vits_session();
  b->s->m = &m->base; b->s->stream = F->s->stream; b->s->own_stream = false;
  int rc = VITS_OK;
  if (hipMalloc((void**)&b->s->d_err, sizeof(int)) != hipSuccess || hipMemsetAsync(b->s->d_err, 0, sizeof(int), F->s->stream) != hipSuccess)
    rc = fail(VITS_ERR_NOMEM, "back context");
  if (rc == VITS_OK) rc = stts_arena(b->s, stts_est_bytes(hp, b->nb, TB, n));
  const int hop = m->vocoder ? m->vocoder->hp.hop_length : 0;
  b->audio_elems = (size_t)TB * hop;
  const size_t o_pau = align_up(sizeof(float) * (size_t)b->nb * CC * TB, 256), o_c = o_pau + align_up(sizeof(float) * (size_t)TB, 256),
               o_mel = o_c + align_up(sizeof(float) * (size_t)b->nb * G, 256), o_audio = o_mel + align_up(sizeof(float) * (size_t)NF * TB, 256),
               total = o_audio + align_up(sizeof(float) * (b->audio_elems + 1), 256);
  if (rc == VITS_OK && (hipMalloc((void**)&b->buf, total) != hipSuccess || hipMemsetAsync(b->buf, 0, total, F->s->stream) != hipSuccess ||
                        hipHostMalloc((void**)&b->out_h, sizeof(float) * (b->audio_elems + (size_t)NF * TB + 1)) != hipSuccess))
    rc = fail(VITS_ERR_NOMEM, "fast-path buffers of frame bucket %d", TB);
  if (rc == VITS_OK) {
    b->d_mu2 = reinterpret_cast<float*>(b->buf); b->d_pau = reinterpret_cast<float*>(b->buf + o_pau); b->d_c = reinterpret_cast<float*>(b->buf + o_c);
    b->d_mel = reinterpret_cast<float*>(b->buf + o_mel); b->d_audio = reinterpret_cast<float*>(b->buf + o_audio);
    if (m->vocoder) {
      rc = session_new(m->vocoder, &b->sv);
      if (rc == VITS_OK) rc = session_reserve(b->sv, 1, 1, TB);
    }
  }
  if (rc == VITS_OK) {
    // time tables of the estimator on the final arena layout (the captured run finds them resident)
    b->E.nb = b->nb; b->E.T = TB; b->E.n_steps = n; b->E.tables_ready = false;
    b->s->arena_used = 0;
    stts_run_cfm(b->s, m, b->E, b->d_c, b->d_mu2, reinterpret_cast<const int*>(F->io_d + F->o_len2), nullptr, TB, 0.f, 0, 1,
                 reinterpret_cast<const int*>(F->io_d + F->o_lenT), nullptr, true);
    if (hipStreamSynchronize(F->s->stream) != hipSuccess) rc = fail(VITS_ERR_DEVICE, "estimator table setup failed");
    b->E.tables_ready = true;
  }
  if (rc != VITS_OK) { stts_back_free(b); return rc; }
  b->last_use = ++F->clock;
  F->backs[key] = b;
  *out = b;
  return VITS_OK;
}

static int stts_phase1(stts_model* m, SttsFront* F) {
  vits_session* s = F->s;
  if (!F->g1) {
    const stts_hparams& hp = m->hp;
    const size_t nmu = (size_t)hp.dp_out * F->TxB;
    s->arena_used = 0;
    HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    CaptureGuard cg(s->stream);
    hipMemcpyAsync(F->io_d, F->io_h, F->a_bytes, hipMemcpyHostToDevice, s->stream);
    stts_run_encoder(s, m, reinterpret_cast<const int64_t*>(F->io_d), reinterpret_cast<const int*>(F->io_d + F->o_len), nullptr, 1, F->TxB,
                     reinterpret_cast<const float*>(F->io_d + F->o_bert), F->d_x, F->d_mu, reinterpret_cast<const int64_t*>(F->io_d + F->o_sid));
    hipMemcpyAsync(F->mu_h, F->d_mu, sizeof(float) * nmu, hipMemcpyDeviceToHost, s->stream);
    hipMemcpyAsync(F->mu_h + nmu, s->d_err, sizeof(int), hipMemcpyDeviceToHost, s->stream);
    TRY(capture_end(s, &F->g1, &cg));
  }
  HIP_TRY(hipGraphLaunch(F->g1, s->stream));
  return VITS_OK;
}

static int stts_phase2(stts_model* m, SttsFront* F, SttsBack* Bk, bool audio) {
  const int gi = audio ? 1 : 0;
  hipStream_t st = F->s->stream;
  if (!Bk->g[gi]) {
    const stts_hparams& hp = m->hp;
    const int NF = hp.n_feats, CC = hp.enc_hidden, G = hp.spk_emb_dim, H = hp.dec_hidden, TB = Bk->TB, nb = Bk->nb;
    vits_session* s = Bk->s;
    const int* d_cum = reinterpret_cast<const int*>(F->io_d + F->o_cum);
    const float* d_pde = reinterpret_cast<const float*>(F->io_d + F->o_pde);
    const int* d_len = reinterpret_cast<const int*>(F->io_d + F->o_len2);
    const int* d_lenT = reinterpret_cast<const int*>(F->io_d + F->o_lenT);
    const int64_t* d_sid = reinterpret_cast<const int64_t*>(F->io_d + F->o_sid);
    const SttsDev* dv = reinterpret_cast<const SttsDev*>(F->io_d + F->o_dev);
    s->arena_used = 0;
    HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CaptureGuard cg(st);
    hipMemcpyAsync(F->io_d + F->a_bytes, F->io_h + F->a_bytes, F->io_bytes - F->a_bytes, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(stts_expand_kernel, dim3(cdiv(TB, 64), CC, 1), dim3(64), 0, st, F->d_x, d_cum, F->TxB, Bk->d_mu2, CC, TB, d_pde, Bk->d_pau);
    if (hp.n_spks > 1) hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(G, 64), 1), dim3(64), 0, st, Bk->d_c, m->spk_emb, d_sid, G, hp.n_spks, s->d_err);
    else hipMemcpyAsync(Bk->d_c, m->zero_vec, sizeof(float) * G, hipMemcpyDeviceToDevice, st);
    if (nb == 2) {
      hipLaunchKernelGGL(fill_rows_kernel, dim3(cdiv(TB, 64), CC), dim3(64), 0, st, Bk->d_mu2 + (size_t)CC * TB, m->fake_content, TB, CC);
      hipMemcpyAsync(Bk->d_c + G, m->fake_speaker, sizeof(float) * G, hipMemcpyDeviceToDevice, st);
    }
    stts_run_cfm(s, m, Bk->E, Bk->d_c, Bk->d_mu2, d_len, nullptr, TB, 0.f, 0, 1, d_lenT, dv);
    hipLaunchKernelGGL(stts_mel_kernel, dim3(cdiv(TB, 64), NF, 1), dim3(64), 0, st, Bk->E.cat, (long long)(NF + H) * TB, TB, Bk->d_pau, Bk->d_mel, NF, TB, d_len,
                       hp.mel_std, hp.mel_mean);
    if (audio) {
      vits_session* sv = Bk->sv;
      hipStream_t own = sv->stream;
      sv->stream = st;
      sv->tile_keys.clear();
      sv->ragged = true; sv->rag_b1 = true;
      hipMemcpyAsync(sv->len_y, d_len, sizeof(int), hipMemcpyDeviceToDevice, st);
      run_decoder(sv, Bk->d_mel, false, 1, TB, Bk->d_audio, (long long)Bk->audio_elems, nullptr, true, 0);
      sv->ragged = false; sv->rag_b1 = false;
      sv->stream = own;
      hipLaunchKernelGGL(clamp_kernel, dim3(cdiv((int)Bk->audio_elems, 256)), dim3(256), 0, st, Bk->d_audio, (long long)Bk->audio_elems);
      hipMemcpyAsync(Bk->out_h, Bk->d_audio, sizeof(float) * Bk->audio_elems, hipMemcpyDeviceToHost, st);
    }
    hipMemcpyAsync(Bk->out_h + Bk->audio_elems, Bk->d_mel, sizeof(float) * (size_t)NF * TB, hipMemcpyDeviceToHost, st);
    TRY(capture_end(F->s, &Bk->g[gi], &cg));
  }
  HIP_TRY(hipGraphLaunch(Bk->g[gi], st));
  return VITS_OK;
}

static int stts_synth_fast(stts_model* m, const int64_t* ids, int32_t Tx, const float* scales, int64_t sid, const float* bert, const float* pde,
                           const stts_synth_opts* opts, float** out_audio, int64_t* out_samples, float** out_mel, int64_t* out_frames) {
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats;
  const int n = (opts && opts->n_timesteps > 0) ? opts->n_timesteps : hp.n_timesteps;
  HIP_TRY(hipSetDevice(m->base.device));
  const int TxB = (Tx + 7) / 8 * 8;
  SttsFront* F = nullptr;
  {
    int rc = stts_front_acquire(m, TxB, &F);
    if (rc == VITS_ERR_NOMEM) { stts_fronts_evict_all(m); rc = stts_front_acquire(m, TxB, &F); }
    if (rc != VITS_OK) return rc;
  }
  struct Rel { stts_model* m; SttsFront* f; ~Rel() { stts_front_release(m, f); } } rel{m, F};
  hipStream_t st = F->s->stream;
  // ---- inputs -> pinned block (rows re-strided to the bucket, padding zero)
  int64_t* h_ids = reinterpret_cast<int64_t*>(F->io_h);
  for (int r = 0; r < 5; ++r) {
    memcpy(h_ids + (size_t)r * TxB, ids + (size_t)r * Tx, sizeof(int64_t) * Tx);
    for (int t = Tx; t < TxB; ++t) h_ids[(size_t)r * TxB + t] = 0;
  }
  *reinterpret_cast<int*>(F->io_h + F->o_len) = Tx;
  *reinterpret_cast<int64_t*>(F->io_h + F->o_sid) = sid;
  float* h_bert = reinterpret_cast<float*>(F->io_h + F->o_bert);
  if (bert) {
    for (int r = 0; r < hp.bert_dim; ++r) {
      memcpy(h_bert + (size_t)r * TxB, bert + (size_t)r * Tx, sizeof(float) * Tx);
      for (int t = Tx; t < TxB; ++t) h_bert[(size_t)r * TxB + t] = 0.f;
    }
  } else {
    memset(h_bert, 0, sizeof(float) * (size_t)hp.bert_dim * TxB);
  }
  // ---- phase 1 and the one host round trip
  TRY(stts_phase1(m, F));
  HIP_TRY(hipStreamSynchronize(st));
  const size_t nmu = (size_t)hp.dp_out * TxB;
  {
    int e = 0;
    memcpy(&e, F->mu_h + nmu, sizeof(int));
    if (e) {
      hipMemsetAsync(F->s->d_err, 0, sizeof(int), st);
      return fail(VITS_ERR_ARG, (e & 2) ? "speaker id out of range" : "token id out of range");
    }
  }
  float* h_pde = reinterpret_cast<float*>(F->io_h + F->o_pde);
  for (int t = 0; t < TxB; ++t) h_pde[t] = (pde && t < Tx) ? pde[t] : 0.f;
  std::vector<int32_t> dur(TxB);
  int64_t ytot = 0;
  stts_durations_host(hp, F->mu_h, 1, TxB, scales[1], pde ? h_pde : nullptr, dur.data(), &ytot);
  int* h_cum = reinterpret_cast<int*>(F->io_h + F->o_cum);
  int ylen = 0;
  for (int j = 0; j < TxB; ++j) { if (j < Tx) ylen += dur[j]; h_cum[j] = ylen; }  // only the utterance's own tokens count
  if (ylen > (1 << 18)) return fail(VITS_ERR_ARG, "T_y unreasonably large");  // (also keeps every per-item [C <= 2048, T] tensor below the 2 GiB the conv kernels' 32-bit offsets address)
  const int T4 = (ylen + 3) / 4 * 4;  // fix_len_compatibility (utils/model.py:14-20): where the exact-size run's tensors end
  const int TB = (T4 + 31) / 32 * 32;
  SttsDev* hv = reinterpret_cast<SttsDev*>(F->io_h + F->o_dev);
  hv->temperature = scales[0]; hv->pad = 0.f; hv->seed = opts ? opts->seed : 0;
  int* h_len2 = reinterpret_cast<int*>(F->io_h + F->o_len2);
  int* h_lenT = reinterpret_cast<int*>(F->io_h + F->o_lenT);
  h_len2[0] = h_len2[1] = ylen;
  h_lenT[0] = h_lenT[1] = T4;
  // ---- phase 2
  SttsBack* Bk = nullptr;
  {
    int rc = stts_back_get(m, F, TB, n, &Bk);
    if (rc == VITS_ERR_NOMEM) {
      stts_fronts_evict_all(m);
      hipStreamSynchronize(F->s->stream);
      for (auto& kv : F->backs) stts_back_free(kv.second);
      F->backs.clear();
      rc = stts_back_get(m, F, TB, n, &Bk);
    }
    if (rc != VITS_OK) return rc;
  }
  const bool audio = out_audio != nullptr;
  TRY(stts_phase2(m, F, Bk, audio));
  const int64_t S = audio ? (int64_t)ylen * m->vocoder->hp.hop_length : 0;
  float* h_audio = audio ? static_cast<float*>(malloc(sizeof(float) * (size_t)(S ? S : 1))) : nullptr;
  float* h_mel = out_mel ? static_cast<float*>(malloc(sizeof(float) * (size_t)NF * (ylen ? ylen : 1))) : nullptr;
  if ((audio && !h_audio) || (out_mel && !h_mel)) { hipStreamSynchronize(st); free(h_audio); free(h_mel); return fail(VITS_ERR_NOMEM, "host alloc failed"); }
  hipError_t se = hipStreamSynchronize(st);
  hipError_t le = hipGetLastError();
  if (se != hipSuccess || le != hipSuccess) {
    free(h_audio); free(h_mel);
    return fail(VITS_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(se != hipSuccess ? se : le));
  }
  if (audio) { memcpy(h_audio, Bk->out_h, sizeof(float) * (size_t)S); *out_audio = h_audio; *out_samples = S; }
  if (out_mel) {
    const float* hm = Bk->out_h + Bk->audio_elems;
    for (int c = 0; c < NF; ++c) memcpy(h_mel + (size_t)c * ylen, hm + (size_t)c * TB, sizeof(float) * (size_t)ylen);
    *out_mel = h_mel; *out_frames = ylen;
  }
  return VITS_OK;
}

extern "C" {

int stts_synthesize(stts_model* m, const int64_t* ids, int32_t Tx, const float* scales, int64_t sid, const float* bert, const float* pde,
                    const stts_synth_opts* opts, float** out_audio, int64_t* out_samples, float** out_mel, int64_t* out_frames) {
  if (!m || !ids || !scales || Tx <= 0 || (out_audio && !out_samples) || (out_mel && !out_frames)) return fail(VITS_ERR_ARG, "bad argument");
  if (out_audio && !m->vocoder) return fail(VITS_ERR_ARG, "no vocoder attached");
  TRY(stts_check_sid(m, &sid, 1));
  {
    static const bool env_off = getenv("VITS_NO_FASTPATH") != nullptr;
    if (g_fast_path && !env_off && !(opts && opts->noise))
      return stts_synth_fast(m, ids, Tx, scales, sid, bert, pde, opts, out_audio, out_samples, out_mel, out_frames);
  }
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats, CC = hp.enc_hidden, G = hp.spk_emb_dim, H = hp.dec_hidden;
  const float temperature = scales[0], length_scale = scales[1];
  const int n = (opts && opts->n_timesteps > 0) ? opts->n_timesteps : hp.n_timesteps, nb = hp.guidance_scale > 0.f ? 2 : 1;
  SttsCall call(m);
  TRY(call.begin(stts_enc_bytes(hp, 1, Tx)));
  vits_session* s = call.s;
  // ---- text encoder + durations
  int64_t* d_ids = call.up(ids, (size_t)5 * Tx);
  const int lx = Tx;
  int* d_lenx = call.up(&lx, 1);
  float* d_bert = bert ? call.up(bert, (size_t)hp.bert_dim * Tx) : call.dev<float>((size_t)hp.bert_dim * Tx);
  float* d_x = call.dev<float>((size_t)CC * Tx);
  float* d_mu = call.dev<float>((size_t)hp.dp_out * Tx);
  if (!d_ids || !d_lenx || !d_bert || !d_x || !d_mu) return fail(VITS_ERR_NOMEM, "device alloc failed");
  if (!bert) HIP_TRY(hipMemsetAsync(d_bert, 0, sizeof(float) * (size_t)hp.bert_dim * Tx, s->stream));
  stts_run_encoder(s, m, d_ids, d_lenx, &sid, 1, Tx, d_bert, d_x, d_mu);
  std::vector<float> mu_dp((size_t)hp.dp_out * Tx);
  HIP_TRY(hipMemcpyAsync(mu_dp.data(), d_mu, sizeof(float) * mu_dp.size(), hipMemcpyDeviceToHost, s->stream));
  TRY(check_err(s));  // also surfaces bad token ids
  std::vector<int32_t> dur(Tx);
  int64_t ylen = 0;
  stts_durations_host(hp, mu_dp.data(), 1, Tx, length_scale, pde, dur.data(), &ylen);
  if (ylen > (1 << 18)) return fail(VITS_ERR_ARG, "T_y unreasonably large");  // (also keeps every per-item [C <= 2048, T] tensor below the 2 GiB the conv kernels' 32-bit offsets address)
  const int T = (int)((ylen + 3) / 4 * 4);  // fix_len_compatibility (utils/model.py:14-20)
  std::vector<int> cum(Tx);
  for (int j = 0, a = 0; j < Tx; ++j) { a += dur[j]; cum[j] = a; }
  // ---- flow-matching decoder
  TRY(stts_arena(s, stts_est_bytes(hp, nb, T, n)));
  int* d_cum = call.up(cum.data(), Tx);
  float* d_pde = pde ? call.up(pde, Tx) : nullptr;
  const int l2[2] = {(int)ylen, (int)ylen};
  int* d_len = call.up(l2, 2);
  float* d_mu2 = call.dev<float>((size_t)nb * CC * T);
  float* d_pau = call.dev<float>(T);
  float* d_c = call.dev<float>((size_t)nb * G);
  float* d_mel = call.dev<float>((size_t)NF * (ylen ? ylen : 1));
  if (!d_cum || !d_len || !d_mu2 || !d_pau || !d_c || !d_mel) return fail(VITS_ERR_NOMEM, "device alloc failed");
  hipLaunchKernelGGL(stts_expand_kernel, dim3(cdiv(T, 64), CC, 1), dim3(64), 0, s->stream, d_x, d_cum, Tx, d_mu2, CC, T, d_pde, d_pau);
  HIP_TRY(hipMemcpyAsync(d_c, hp.n_spks > 1 ? m->spk_emb + (size_t)sid * G : m->zero_vec, sizeof(float) * G, hipMemcpyDeviceToDevice, s->stream));
  if (nb == 2) {
    hipLaunchKernelGGL(fill_rows_kernel, dim3(cdiv(T, 64), CC), dim3(64), 0, s->stream, d_mu2 + (size_t)CC * T, m->fake_content, T, CC);
    HIP_TRY(hipMemcpyAsync(d_c + G, m->fake_speaker, sizeof(float) * G, hipMemcpyDeviceToDevice, s->stream));
  }
  float* d_noise = nullptr;
  long long nstride = T;
  if (opts && opts->noise) {
    if (opts->noise_stride < T) return fail(VITS_ERR_ARG, "noise stride %lld < %d", (long long)opts->noise_stride, T);
    nstride = opts->noise_stride;
    d_noise = call.up(opts->noise, (size_t)NF * nstride);
    if (!d_noise) return fail(VITS_ERR_NOMEM, "device alloc failed");
  }
  SttsEst E; E.nb = nb; E.T = T; E.n_steps = n;
  stts_run_cfm(s, m, E, d_c, d_mu2, d_len, d_noise, nstride, temperature, opts ? opts->seed : 0);
  hipLaunchKernelGGL(stts_mel_kernel, dim3(cdiv((int)ylen, 64), NF, 1), dim3(64), 0, s->stream, E.cat, (long long)(NF + H) * T, T, d_pau, d_mel, NF, (int)ylen, d_len,
                     hp.mel_std, hp.mel_mean);
  float* h_mel = nullptr;
  if (out_mel) {
    h_mel = static_cast<float*>(malloc(sizeof(float) * (size_t)NF * ylen));
    if (!h_mel) return fail(VITS_ERR_NOMEM, "host alloc failed");
    hipMemcpyAsync(h_mel, d_mel, sizeof(float) * (size_t)NF * ylen, hipMemcpyDeviceToHost, s->stream);
  }
  float* h_audio = nullptr;
  int64_t S = 0;
  int rc = VITS_OK;
  if (out_audio) {  // vocoder.decode(mel).clamp(-1, 1) (onnx/export.py:28-31): the vocoder-only model's decoder stage
    vits_model* v = m->vocoder;
    S = ylen * v->hp.hop_length;
    float* d_audio = call.dev<float>((size_t)S);
    h_audio = static_cast<float*>(malloc(sizeof(float) * (size_t)(S ? S : 1)));
    vits_session* sv = nullptr;
    if (!d_audio || !h_audio) rc = fail(VITS_ERR_NOMEM, "alloc failed");
    if (rc == VITS_OK) rc = pool_acquire(v, &sv);
    if (rc == VITS_OK) rc = session_reserve(sv, 1, 1, (int)ylen);
    if (rc == VITS_OK) {
      // the vocoder's launches go onto THIS call's stream (its session only lends the decoder workspace): the mel is
      // consumed in stream order, no host synchronisation between the two halves
      hipStream_t own = sv->stream;
      sv->stream = s->stream;
      run_decoder(sv, d_mel, false, 1, (int)ylen, d_audio, S, nullptr);
      sv->stream = own;
      hipLaunchKernelGGL(clamp_kernel, dim3(cdiv((int)S, 256)), dim3(256), 0, s->stream, d_audio, (long long)S);
      hipMemcpyAsync(h_audio, d_audio, sizeof(float) * (size_t)S, hipMemcpyDeviceToHost, s->stream);
      rc = check_err(s);
    }
    if (sv) pool_release(v, sv);
  } else {
    rc = check_err(s);
  }
  if (rc != VITS_OK) { free(h_mel); free(h_audio); return rc; }
  if (out_audio) { *out_audio = h_audio; *out_samples = S; }
  if (out_mel) { *out_mel = h_mel; *out_frames = ylen; }
  (void)H;
  return VITS_OK;
}

int stts_stream_open(stts_model* m, const int64_t* ids, int32_t Tx, const float* scales, int64_t sid, const float* bert, const float* pde,
                     const stts_synth_opts* opts, int32_t chunk_frames, vits_stream** out, int64_t* total_samples) {
  if (!m || !out || chunk_frames <= 0) return fail(VITS_ERR_ARG, "bad argument");
  if (!m->vocoder) return fail(VITS_ERR_ARG, "no vocoder attached");
  float* mel = nullptr;
  int64_t frames = 0;
  TRY(stts_synthesize(m, ids, Tx, scales, sid, bert, pde, opts, nullptr, nullptr, &mel, &frames));
  // the mel crosses the host once (80 x T_y floats): the acoustic context that produced it is cached and reused by other calls
  int rc = frames > 0 ? vits_stream_open_latent(m->vocoder, mel, (int32_t)frames, chunk_frames, 1u, out, total_samples)
                      : fail(VITS_ERR_ARG, "empty utterance");
  free(mel);
  return rc;
}

int stts_synthesize_batch(stts_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                          const int64_t* sid, const float* bert, const float* pde, const stts_synth_opts* opts, float** out_audio,
                          int64_t* out_samples, int64_t* out_lengths) {
  if (!m || !ids || !lengths || !scales || !out_audio || !out_samples || !out_lengths || B <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  if (!m->vocoder) return fail(VITS_ERR_ARG, "no vocoder attached");
  if (opts && opts->noise) return fail(VITS_ERR_ARG, "injected noise is a single-utterance option");
  for (int b = 0; b < B; ++b) if (lengths[b] <= 0 || lengths[b] > Tx) return fail(VITS_ERR_ARG, "length out of range");
  TRY(stts_check_sid(m, sid, B));
  const stts_hparams& hp = m->hp;
  const int NF = hp.n_feats, CC = hp.enc_hidden, G = hp.spk_emb_dim, H = hp.dec_hidden;
  const float temperature = scales[0], length_scale = scales[1];
  const int n = (opts && opts->n_timesteps > 0) ? opts->n_timesteps : hp.n_timesteps, cfg = hp.guidance_scale > 0.f ? 1 : 0, nb = cfg ? 2 * B : B;
  SttsCall call(m);
  TRY(call.begin(stts_enc_bytes(hp, B, Tx)));
  vits_session* s = call.s;
  // ---- text encoder (masked per item) + durations on the host
  int64_t* d_ids = call.up(ids, (size_t)B * 5 * Tx);
  std::vector<int> lx(B);
  for (int b = 0; b < B; ++b) lx[b] = (int)lengths[b];
  int* d_lenx = call.up(lx.data(), B);
  float* d_bert = bert ? call.up(bert, (size_t)B * hp.bert_dim * Tx) : call.dev<float>((size_t)B * hp.bert_dim * Tx);
  float* d_x = call.dev<float>((size_t)B * CC * Tx);
  float* d_mu = call.dev<float>((size_t)B * hp.dp_out * Tx);
  if (!d_ids || !d_lenx || !d_bert || !d_x || !d_mu) return fail(VITS_ERR_NOMEM, "device alloc failed");
  if (!bert) HIP_TRY(hipMemsetAsync(d_bert, 0, sizeof(float) * (size_t)B * hp.bert_dim * Tx, s->stream));
  stts_run_encoder(s, m, d_ids, d_lenx, sid, B, Tx, d_bert, d_x, d_mu);
  std::vector<float> mu_dp((size_t)B * hp.dp_out * Tx);
  HIP_TRY(hipMemcpyAsync(mu_dp.data(), d_mu, sizeof(float) * mu_dp.size(), hipMemcpyDeviceToHost, s->stream));
  TRY(check_err(s));
  std::vector<int32_t> dur((size_t)B * Tx);
  std::vector<int64_t> ylen(B);
  stts_durations_host(hp, mu_dp.data(), B, Tx, length_scale, pde, dur.data(), ylen.data());
  std::vector<int> cum((size_t)B * Tx), l2((size_t)2 * B), lT((size_t)2 * B);
  int T = 4, Tm = 1;
  for (int b = 0; b < B; ++b) {  // only the item's own tokens count (a single-utterance call has no padded tokens)
    int a = 0;
    for (int j = 0; j < Tx; ++j) { if (j < lengths[b]) a += dur[(size_t)b * Tx + j]; cum[(size_t)b * Tx + j] = a; }
    ylen[b] = a;
    if (a > (1 << 22)) return fail(VITS_ERR_ARG, "T_y unreasonably large");
    const int Tb = (a + 3) / 4 * 4;  // fix_len_compatibility per item
    l2[b] = l2[B + b] = a; lT[b] = lT[B + b] = Tb;
    if (Tb > T) T = Tb;
    if (a > Tm) Tm = a;
  }
  // ---- flow matching over all items (and their CFG twins) at once
  TRY(stts_arena(s, stts_est_bytes(hp, nb, T, n)));
  int* d_cum = call.up(cum.data(), (size_t)B * Tx);
  float* d_pde = pde ? call.up(pde, (size_t)B * Tx) : nullptr;
  int* d_len = call.up(l2.data(), (size_t)2 * B);
  int* d_lenT = call.up(lT.data(), (size_t)2 * B);
  float* d_mu2 = call.dev<float>((size_t)nb * CC * T);
  float* d_pau = call.dev<float>((size_t)B * T);
  float* d_c = call.dev<float>((size_t)nb * G);
  float* d_mel = call.dev<float>((size_t)B * NF * Tm);
  if (!d_cum || !d_len || !d_lenT || !d_mu2 || !d_pau || !d_c || !d_mel) return fail(VITS_ERR_NOMEM, "device alloc failed");
  hipLaunchKernelGGL(stts_expand_kernel, dim3(cdiv(T, 64), CC, B), dim3(64), 0, s->stream, d_x, d_cum, Tx, d_mu2, CC, T, d_pde, d_pau);
  for (int b = 0; b < B; ++b) {
    HIP_TRY(hipMemcpyAsync(d_c + (size_t)b * G, hp.n_spks > 1 ? m->spk_emb + (size_t)sid[b] * G : m->zero_vec, sizeof(float) * G, hipMemcpyDeviceToDevice, s->stream));
    if (cfg) HIP_TRY(hipMemcpyAsync(d_c + (size_t)(B + b) * G, m->fake_speaker, sizeof(float) * G, hipMemcpyDeviceToDevice, s->stream));
  }
  if (cfg) hipLaunchKernelGGL(fill_rows_kernel, dim3(cdiv(T, 64), B * CC), dim3(64), 0, s->stream, d_mu2 + (size_t)B * CC * T, m->fake_content, T, CC);
  SttsEst E; E.nb = nb; E.T = T; E.n_steps = n;
  // ragged batch: compact tile maps so that no kernel dispatches the padding of the shorter items
  s->tile_tabs = call.dev<int>((size_t)32 * (nb + 1));
  if (!s->tile_tabs) return fail(VITS_ERR_NOMEM, "device alloc failed");
  s->tile_keys.clear();
  s->B = nb;
  s->ragged = B > 1;
  struct RaggedOff { vits_session* s; ~RaggedOff() { s->ragged = false; s->tile_tabs = nullptr; s->tile_keys.clear(); } } ragged_off{s};
  const unsigned long long* d_seeds = nullptr;
  if (opts && (opts->flags & STTS_FLAG_ITEM_SEEDS) && opts->item_seeds) {
    d_seeds = call.up(reinterpret_cast<const unsigned long long*>(opts->item_seeds), (size_t)B);
    if (!d_seeds) return fail(VITS_ERR_NOMEM, "device alloc failed");
  }
  stts_run_cfm(s, m, E, d_c, d_mu2, d_len, nullptr, T, temperature, opts ? opts->seed : 0, B, d_lenT, nullptr, false, d_seeds);
  hipLaunchKernelGGL(stts_mel_kernel, dim3(cdiv(Tm, 64), NF, B), dim3(64), 0, s->stream, E.cat, (long long)(NF + H) * T, T, d_pau, d_mel, NF, Tm, d_len,
                     hp.mel_std, hp.mel_mean);
  // ---- vocoder over the ragged batch, every item decoded as if alone (rag halo 0)
  vits_model* v = m->vocoder;
  const int64_t S = (int64_t)Tm * v->hp.hop_length;
  float* d_audio = call.dev<float>((size_t)B * S);
  float* h_audio = static_cast<float*>(malloc(sizeof(float) * (size_t)B * S));
  vits_session* sv = nullptr;
  int rc = (!d_audio || !h_audio) ? fail(VITS_ERR_NOMEM, "alloc failed") : VITS_OK;
  if (rc == VITS_OK) rc = pool_acquire(v, &sv);
  if (rc == VITS_OK) rc = session_reserve(sv, B, 1, Tm);
  if (rc == VITS_OK) {
    hipStream_t own = sv->stream;
    sv->stream = s->stream;
    sv->tile_keys.clear();
    sv->ragged = B > 1;
    hipMemcpyAsync(sv->len_y, d_len, sizeof(int) * B, hipMemcpyDeviceToDevice, s->stream);
    run_decoder(sv, d_mel, false, B, Tm, d_audio, S, nullptr, true, 0);
    sv->ragged = false;
    sv->stream = own;
    hipLaunchKernelGGL(clamp_kernel, dim3(cdiv((int)(B * S), 256)), dim3(256), 0, s->stream, d_audio, (long long)B * S);
    hipMemcpyAsync(h_audio, d_audio, sizeof(float) * (size_t)B * S, hipMemcpyDeviceToHost, s->stream);
    rc = check_err(s);
  }
  if (sv) pool_release(v, sv);
  if (rc != VITS_OK) { free(h_audio); return rc; }
  *out_audio = h_audio;
  *out_samples = S;
  for (int b = 0; b < B; ++b) out_lengths[b] = ylen[b] * v->hp.hop_length;
  return VITS_OK;
}

}  // extern "C"

// ================================================================== word-embedding BERT encoder (include/stts_mi355.h)
// transformers.BertModel up to hidden_states[-3] on the same kernels: tokens are columns ([H, T] channel-major), every
// Linear is a 1x1 conv launch (q/k/v fused, GELU in the intermediate conv's epilogue), self-attention is the MFMA flash
// kernel without relative tables, residual + LayerNorm(eps) is layernorm_c_kernel(a + b).
struct BertLayerW { ConvW qkv, o, c1, c2; float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr; };
// One captured forward per token-count bucket (round 5): get_word_bert runs in front of EVERY request of a BERT-conditioned voice
// (vosk_tts/synth.py:25-44), and 73 eager launches + two hipMalloc / hipFree per call were 0.9 ms of such a request.  A context owns its
// session (stream + workspace laid out for the bucket), a pinned input block [ids | token types | length], the device copy, a pinned
// output block and the graph: memcpy node in, the launches, memcpy node out.  Bucket columns beyond the sentence are [PAD] tokens whose
// columns nothing valid reads: every op of the encoder is column-local except attention, which masks keys at the length.
struct BertCtx {
  vits_session* s = nullptr;
  int Tb = 0;
  char *io_h = nullptr, *io_d = nullptr;
  float *out_h = nullptr, *ot = nullptr;
  hipGraphExec_t g = nullptr;
  bool busy = false;
};
struct bert_model {
  vits_model base;
  bert_hparams hp;
  float *we = nullptr, *pe = nullptr, *te = nullptr, *eg = nullptr, *eb = nullptr;
  std::vector<BertLayerW> layers;
  std::mutex ctx_mu;
  std::map<int, BertCtx*> ctx;  // by bucket
};
static void bert_ctx_free(BertCtx* c) {
  if (!c) return;
  if (c->g) hipGraphExecDestroy(c->g);
  if (c->io_h) hipHostFree(c->io_h);
  if (c->io_d) hipFree(c->io_d);
  if (c->out_h) hipHostFree(c->out_h);
  if (c->s) session_free(c->s);
  delete c;
}

static ConvW bert_linear(vits_model* b, const char* name, int Cout, int Cin) {
  const float* w = tget(b, 2, Cout, Cin, -1, "%s.weight", name);
  const float* bias = tget(b, 1, Cout, -1, -1, "%s.bias", name);
  if (b->missing) return ConvW();
  return make_conv(b, Cout, Cin, 1, bias, [&](int r, int ci, int) { return w[(size_t)r * Cin + ci]; });
}

static int bert_load(bert_model* m) {
  vits_model* b = &m->base;
  const bert_hparams& hp = m->hp;
  const int H = hp.hidden, F = hp.intermediate;
  if (H % 32 || F % 32 || hp.n_heads <= 0 || H % hp.n_heads || H > 48 * LN_CG) return fail(VITS_ERR_UNSUPPORTED, "BERT geometry unsupported");
  const int dk = H / hp.n_heads;
  if (dk != 32 && dk != 64 && dk != 96) return fail(VITS_ERR_UNSUPPORTED, "head dim %d not in {32,64,96}", dk);
  if (hp.out_layers < 0 || hp.out_layers > hp.n_layers) return fail(VITS_ERR_UNSUPPORTED, "out_layers out of range");
  m->we = upload(b, tget(b, 2, hp.vocab_size, H, -1, "embeddings.word_embeddings.weight"), (size_t)hp.vocab_size * H);
  m->pe = upload(b, tget(b, 2, hp.max_position, H, -1, "embeddings.position_embeddings.weight"), (size_t)hp.max_position * H);
  m->te = upload(b, tget(b, 2, hp.type_vocab, H, -1, "embeddings.token_type_embeddings.weight"), (size_t)hp.type_vocab * H);
  m->eg = upload(b, tget(b, 1, H, -1, -1, "embeddings.LayerNorm.weight"), H);
  m->eb = upload(b, tget(b, 1, H, -1, -1, "embeddings.LayerNorm.bias"), H);
  char nm[200];
  for (int l = 0; l < hp.out_layers && !b->missing; ++l) {
    BertLayerW L;
    const float* wq = tget(b, 2, H, H, -1, "encoder.layer.%d.attention.self.query.weight", l);
    const float* wk = tget(b, 2, H, H, -1, "encoder.layer.%d.attention.self.key.weight", l);
    const float* wv = tget(b, 2, H, H, -1, "encoder.layer.%d.attention.self.value.weight", l);
    const float* bq = tget(b, 1, H, -1, -1, "encoder.layer.%d.attention.self.query.bias", l);
    const float* bk = tget(b, 1, H, -1, -1, "encoder.layer.%d.attention.self.key.bias", l);
    const float* bv = tget(b, 1, H, -1, -1, "encoder.layer.%d.attention.self.value.bias", l);
    if (b->missing) break;
    std::vector<float> bias((size_t)3 * H);
    memcpy(bias.data(), bq, sizeof(float) * H); memcpy(bias.data() + H, bk, sizeof(float) * H); memcpy(bias.data() + 2 * H, bv, sizeof(float) * H);
    L.qkv = make_conv(b, 3 * H, H, 1, bias.data(), [&](int r, int ci, int) { return (r < H ? wq : (r < 2 * H ? wk : wv))[(size_t)(r % H) * H + ci]; });
    snprintf(nm, sizeof nm, "encoder.layer.%d.attention.output.dense", l); L.o = bert_linear(b, nm, H, H);
    snprintf(nm, sizeof nm, "encoder.layer.%d.intermediate.dense", l); L.c1 = bert_linear(b, nm, F, H);
    snprintf(nm, sizeof nm, "encoder.layer.%d.output.dense", l); L.c2 = bert_linear(b, nm, H, F);
    L.g1 = upload(b, tget(b, 1, H, -1, -1, "encoder.layer.%d.attention.output.LayerNorm.weight", l), H);
    L.b1 = upload(b, tget(b, 1, H, -1, -1, "encoder.layer.%d.attention.output.LayerNorm.bias", l), H);
    L.g2 = upload(b, tget(b, 1, H, -1, -1, "encoder.layer.%d.output.LayerNorm.weight", l), H);
    L.b2 = upload(b, tget(b, 1, H, -1, -1, "encoder.layer.%d.output.LayerNorm.bias", l), H);
    m->layers.push_back(L);
  }
  return b->missing ? VITS_ERR_BLOB : VITS_OK;
}

extern "C" {

int stts_bert_create(const void* blob, size_t bytes, int device, bert_model** out) {
  if (!blob || !out || bytes < 16 + sizeof(bert_hparams)) return fail(VITS_ERR_ARG, "bad blob argument");
  const unsigned char* p = static_cast<const unsigned char*>(blob);
  if (memcmp(p, "BERTW001", 8) != 0) return fail(VITS_ERR_BLOB, "bad magic");
  uint32_t hb;
  memcpy(&hb, p + 8, 4);
  if (hb != sizeof(bert_hparams)) return fail(VITS_ERR_BLOB, "hparams size %u != %zu", hb, sizeof(bert_hparams));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(VITS_ERR_DEVICE, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(VITS_ERR_ARG, "device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  bert_model* m = new bert_model();
  memcpy(&m->hp, p + 12, sizeof(bert_hparams));
  if (m->hp.abi_version != BERT_ABI_VERSION) { delete m; return fail(VITS_ERR_BLOB, "abi version mismatch"); }
  vits_model* b = &m->base;
  b->device = device; b->acoustic = false;
  b->blob = p; b->blob_bytes = bytes;
  memcpy(&b->n_entries, p + 12 + hb, 4);
  b->entries = reinterpret_cast<const vits_blob_entry*>(p + 16 + hb);
  int rc = VITS_OK;
  if (16 + hb + (size_t)b->n_entries * sizeof(vits_blob_entry) > bytes) rc = fail(VITS_ERR_BLOB, "truncated table");
  for (uint32_t i = 0; rc == VITS_OK && i < b->n_entries; ++i)
    if (b->entries[i].offset > bytes || b->entries[i].nelem > (bytes - b->entries[i].offset) / 4) rc = fail(VITS_ERR_BLOB, "truncated data");  // overflow-safe
  if (rc == VITS_OK) rc = bert_load(m);
  b->blob = nullptr; b->entries = nullptr;
  if (rc != VITS_OK) { for (void* a : b->allocs) hipFree(a); delete m; return rc; }
  hipDeviceSynchronize();
  *out = m;
  return VITS_OK;
}

void stts_bert_destroy(bert_model* m) {
  if (!m) return;
  hipSetDevice(m->base.device);
  for (auto& kv : m->ctx) bert_ctx_free(kv.second);
  for (vits_session* s : m->base.pool) session_free(s);
  for (void* a : m->base.allocs) hipFree(a);
  delete m;
}

int stts_bert_get_hparams(const bert_model* m, bert_hparams* out) {
  if (!m || !out) return fail(VITS_ERR_ARG, "null argument");
  *out = m->hp;
  return VITS_OK;
}

// the encoder's launches on session s: ids / types int64 [T] and len int32 on the device -> ot [T][H]
static void bert_forward(bert_model* m, vits_session* s, const int64_t* d_ids, const int64_t* d_ty, const int* d_len, int T, float* x, float* y, float* att,
                         float* qkv, float* ff, float* ot) {
  const bert_hparams& hp = m->hp;
  const int H = hp.hidden, nh = hp.n_heads;
  hipLaunchKernelGGL(bert_embed_kernel, dim3(cdiv(T, 64), H), dim3(64), 0, s->stream, d_ids, d_ty, m->we, m->pe, m->te, y, H, T,
                     hp.vocab_size, hp.type_vocab, s->d_err);
  {
    LNParams P{y, nullptr, nullptr, x, m->eg, m->eb, nullptr, H, T, 0, 0, 0, 0, hp.ln_eps, nullptr, nullptr};
    launch_layernorm(s->stream, P, 1);
  }
  // (round 5) sentence-sized calls: the 3072 -> 768 matrix of the FFN is the one launch of a layer that ran on 24 CUs (21 us of a 75 us
  // layer, profiles/r5_bert_ffn2.txt); VITS_BERT_KSLICE=0: the single launch (A/B)
  static const bool kslice_env = !(getenv("VITS_BERT_KSLICE") && atoi(getenv("VITS_BERT_KSLICE")) == 0);
  const bool ffn2_slices = kslice_env && g_force_tile == 0 && T <= 64 && T >= 4 && hp.intermediate % (3 * CONV_CI_T) == 0 && hp.intermediate / 3 >= 8 * CONV_CI_T &&
                           H % 32 == 0 && !m->layers.empty() && m->layers[0].c2.K == 1;
  for (const BertLayerW& L : m->layers) {
    ConvParams P = conv_params(L.qkv, x, qkv, 1, T, 1, 0);
    launch_conv(s, P, EPI_STORE, "bert.qkv");
    launch_attention_raw(s, qkv, nullptr, nullptr, d_len, att, 1, H, T, nh, 4);
    P = conv_params(L.o, att, y, 1, T, 1, 0);
    launch_conv(s, P, EPI_STORE, "bert.o");
    { LNParams Q{y, x, nullptr, x, L.g1, L.b1, nullptr, H, T, 0, 0, 0, 0, hp.ln_eps, nullptr, nullptr}; launch_layernorm(s->stream, Q, 1); }  // LayerNorm(dense(ctx) + x)
    P = conv_params(L.c1, x, ff, 1, T, 1, 0); P.relu = 3;
    launch_conv(s, P, EPI_STORE, "bert.ffn1");
    P = conv_params(L.c2, ff, y, 1, T, 1, 0);
    if (ffn2_slices) {
      // K-sliced: three contiguous thirds of the 3072 input channels as the three groups of ONE conv_wp launch (72 workgroups of 8 waves,
      // each streaming 128 KB of weights) instead of the K-split kernel's 24 workgroups x 384 KB; the partial tensors land in the qkv
      // buffer (dead since the attention, exactly 3 x [H, T]) and the LayerNorm behind sums them
      const int Cs = hp.intermediate / 3;
      P.n_groups = 3; P.Cin = Cs; P.x_bstride = (long long)Cs * T;
      for (int j = 0; j < 3; ++j) {
        P.g[j] = P.g[0];
        P.g[j].x = ff + (size_t)j * Cs * T;
        P.g[j].w = L.c2.w + (size_t)j * (Cs / CONV_CI_T) * 2 * 64 * 4;  // packed [m-block][chunk][2 step groups][64 lanes][4]: the m-block stride (n_sg) stays the whole matrix's
        P.g[j].w16 = nullptr; P.g[j].wb = nullptr;
        P.g[j].bias = j == 0 ? L.c2.bias : nullptr;
        P.g[j].y = qkv + (size_t)j * H * T;
      }
      ProfScope ps(s, "bert.ffn2", 2.0 * H * hp.intermediate * (double)T);
      launch_conv_wp(s, P, ps);
    } else {
      launch_conv(s, P, EPI_STORE, "bert.ffn2");
    }
    {
      LNParams Q{ffn2_slices ? qkv : y, x, nullptr, x, L.g2, L.b2, nullptr, H, T, 0, 0, 0, 0, hp.ln_eps, nullptr, nullptr, ffn2_slices ? 3 : 0, (long long)H * T};
      launch_layernorm(s->stream, Q, 1);
    }
  }
  hipLaunchKernelGGL(transpose_ct_kernel, dim3(cdiv(H, 256), T), dim3(256), 0, s->stream, x, ot, H, T);
}

// graph-replayed form: returns 1 when it served the call, 0 when the caller should take the eager form (bucket busy in another
// thread, allocation failure), < 0 on an error of the call itself
static int bert_encode_graph(bert_model* m, const int64_t* ids, const int64_t* types, int T, float* out) {
  static const bool off = getenv("VITS_NO_FASTPATH") != nullptr;
  if (off || !g_fast_path) return 0;
  const bert_hparams& hp = m->hp;
  const int H = hp.hidden, F = hp.intermediate;
  int Tb = (T + 7) / 8 * 8;
  if (Tb > hp.max_position) Tb = hp.max_position;
  BertCtx* c = nullptr;
  {
    std::lock_guard<std::mutex> g(m->ctx_mu);
    auto it = m->ctx.find(Tb);
    if (it != m->ctx.end()) {
      if (it->second->busy) return 0;
      c = it->second;
    } else {
      if (m->ctx.size() >= 64) return 0;
      c = new BertCtx();
      c->Tb = Tb;
      const size_t io = (size_t)2 * Tb * sizeof(int64_t) + 64;
      bool ok = session_new(&m->base, &c->s) == VITS_OK;
      ok = ok && stts_arena(c->s, ((size_t)Tb * (H * 4 + 3 * H + F) + 64) * sizeof(float) + 64 * 1024) == VITS_OK;
      ok = ok && hipHostMalloc((void**)&c->io_h, io) == hipSuccess && hipMalloc((void**)&c->io_d, io) == hipSuccess;
      ok = ok && hipHostMalloc((void**)&c->out_h, sizeof(float) * (size_t)Tb * H) == hipSuccess;
      if (!ok) { bert_ctx_free(c); (void)hipGetLastError(); return 0; }
      m->ctx[Tb] = c;
    }
    c->busy = true;
  }
  struct Done { bert_model* m; BertCtx* c; ~Done() { std::lock_guard<std::mutex> g(m->ctx_mu); c->busy = false; } } done{m, c};
  vits_session* s = c->s;
  int64_t* h_ids = reinterpret_cast<int64_t*>(c->io_h);
  int64_t* h_ty = h_ids + Tb;
  int* h_len = reinterpret_cast<int*>(h_ty + Tb);
  for (int t = 0; t < Tb; ++t) { h_ids[t] = t < T ? ids[t] : 0; h_ty[t] = (t < T && types) ? types[t] : 0; }
  *h_len = T;
  if (!c->g) {
    s->arena_used = 0;
    float* x = bump<float>(s, (size_t)H * Tb); float* y = bump<float>(s, (size_t)H * Tb); float* att = bump<float>(s, (size_t)H * Tb);
    float* qkv = bump<float>(s, (size_t)3 * H * Tb); float* ff = bump<float>(s, (size_t)F * Tb);
    c->ot = bump<float>(s, (size_t)H * Tb);
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return 0; }
    CaptureGuard cg(s->stream);
    hipMemcpyAsync(c->io_d, c->io_h, (size_t)2 * Tb * sizeof(int64_t) + 64, hipMemcpyHostToDevice, s->stream);
    const int64_t* d_ids = reinterpret_cast<const int64_t*>(c->io_d);
    bert_forward(m, s, d_ids, d_ids + Tb, reinterpret_cast<const int*>(d_ids + 2 * Tb), Tb, x, y, att, qkv, ff, c->ot);
    hipMemcpyAsync(c->out_h, c->ot, sizeof(float) * (size_t)Tb * H, hipMemcpyDeviceToHost, s->stream);
    if (capture_end(s, &c->g, &cg) != VITS_OK) { c->g = nullptr; return 0; }
  }
  { const hipError_t le = hipGraphLaunch(c->g, s->stream); if (le != hipSuccess) return -fail(VITS_ERR_DEVICE, "hipGraphLaunch failed: %s", hipGetErrorString(le)); }
  const int rc = check_err(s);  // (synchronises the stream; error codes are positive: handed back negated)
  if (rc != VITS_OK) return -rc;
  memcpy(out, c->out_h, sizeof(float) * (size_t)T * H);
  return 1;
}

int stts_bert_encode(bert_model* m, const int64_t* ids, const int64_t* types, int32_t T, float* out) {
  if (!m || !ids || !out || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const bert_hparams& hp = m->hp;
  if (T > hp.max_position) return fail(VITS_ERR_ARG, "%d tokens exceed max_position %d", T, hp.max_position);
  const int H = hp.hidden, F = hp.intermediate;
  HIP_TRY(hipSetDevice(m->base.device));
  {
    const int gr = bert_encode_graph(m, ids, types, T, out);
    if (gr == 1) return VITS_OK;
    if (gr < 0) return -gr;
  }
  vits_session* s = nullptr;
  TRY(pool_acquire(&m->base, &s));
  struct Rel { vits_model* b; vits_session* s; std::vector<void*> tmp; ~Rel() { hipStreamSynchronize(s->stream); pool_release(b, s); for (void* p : tmp) hipFree(p); } } rel{&m->base, s, {}};
  TRY(stts_arena(s, ((size_t)T * (H * 4 + 3 * H + F) + 64) * sizeof(float) + 64 * 1024));
  float* x = bump<float>(s, (size_t)H * T); float* y = bump<float>(s, (size_t)H * T); float* att = bump<float>(s, (size_t)H * T);
  float* qkv = bump<float>(s, (size_t)3 * H * T); float* ff = bump<float>(s, (size_t)F * T); float* ot = bump<float>(s, (size_t)H * T);
  int* d_len = bump<int>(s, 1);
  void* d_ids = nullptr; void* d_ty = nullptr;
  if (hipMalloc(&d_ids, sizeof(int64_t) * T) != hipSuccess) return fail(VITS_ERR_NOMEM, "device alloc failed");
  rel.tmp.push_back(d_ids);
  if (types) { if (hipMalloc(&d_ty, sizeof(int64_t) * T) != hipSuccess) return fail(VITS_ERR_NOMEM, "device alloc failed"); rel.tmp.push_back(d_ty); }
  HIP_TRY(hipMemcpyAsync(d_ids, ids, sizeof(int64_t) * T, hipMemcpyHostToDevice, s->stream));
  if (types) HIP_TRY(hipMemcpyAsync(d_ty, types, sizeof(int64_t) * T, hipMemcpyHostToDevice, s->stream));
  const int tl = T;
  HIP_TRY(hipMemcpyAsync(d_len, &tl, sizeof(int), hipMemcpyHostToDevice, s->stream));
  bert_forward(m, s, (const int64_t*)d_ids, (const int64_t*)d_ty, d_len, T, x, y, att, qkv, ff, ot);
  HIP_TRY(hipMemcpyAsync(out, ot, sizeof(float) * (size_t)T * H, hipMemcpyDeviceToHost, s->stream));
  return check_err(s);
}

}  // extern "C"
