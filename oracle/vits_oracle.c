/*
 * vits_oracle.c — TEST INFRASTRUCTURE.  CPU restatement (plain C, fp32) of the
 * reference's VITS2 inference arithmetic, used ONLY as the parity checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * the product path (vosk_tts_amd/) links, loads or calls this file.
 *
 * Pinned against the reference: tests/test_oracle_golden.py compares every
 * stage with fixtures in tests/golden/ that oracle/gen_golden.py produced by
 * running the reference's own PyTorch modules (imported from
 * /root/reference/training/vits2 in the build container).  The reference ships
 * no tests or golden vectors for this path, so those generated
 * fixtures are the pin.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/training/vits2/).  Exports the ABI of include/vits_mi355.h
 * with the prefix vitsref_ instead of vits_.
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "../include/vits_mi355.h"

#define API(name) vitsref_##name
#define PI_D 3.14159265358979323846

static __thread char g_err[512];
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

struct vits_model {
  vits_hparams hp;
  unsigned char* blob;
  size_t blob_bytes;
  uint32_t n_entries;
  const vits_blob_entry* entries;
  float* istft_basis; /* [n_fft+2][n_fft]  (stft.py:191-214 inverse_basis) */
  float* pqmf_syn;    /* [subbands][taps+1] (pqmf.py:64-75 synthesis_filter) */
  int missing;        /* set when a tensor lookup failed */
};

/* ------------------------------------------------------------------ blob */

static const float* tget(vits_model* m, int ndim, int d0, int d1, int d2, const char* fmt, ...) {
  char name[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(name, sizeof name, fmt, ap);
  va_end(ap);
  for (uint32_t i = 0; i < m->n_entries; ++i) {
    const vits_blob_entry* e = &m->entries[i];
    if (strncmp(e->name, name, sizeof e->name) == 0) {
      int want[3] = {d0, d1, d2};
      if ((int)e->ndim != ndim) { m->missing = 1; fail(VITS_ERR_BLOB, "tensor %s: ndim %u != %d", name, e->ndim, ndim); return NULL; }
      for (int k = 0; k < ndim && k < 3; ++k)
        if (want[k] >= 0 && (int)e->dims[k] != want[k]) {
          m->missing = 1;
          fail(VITS_ERR_BLOB, "tensor %s: dim %d is %u, expected %d", name, k, e->dims[k], want[k]);
          return NULL;
        }
      return (const float*)(m->blob + e->offset);
    }
  }
  m->missing = 1;
  fail(VITS_ERR_BLOB, "tensor %s missing from blob", name);
  return NULL;
}

/* optional tensors (e.g. the conv_post bias of the StableTTS HiFi-GAN): presence test without raising `missing` */
static int thas(const vits_model* m, const char* name) {
  for (uint32_t i = 0; i < m->n_entries; ++i)
    if (strncmp(m->entries[i].name, name, sizeof m->entries[i].name) == 0) return 1;
  return 0;
}

static float* falloc(size_t n) {
  float* p = (float*)calloc(n ? n : 1, sizeof(float));
  if (!p) { fprintf(stderr, "vits_oracle: out of memory\n"); abort(); }
  return p;
}

/* --------------------------------------------------------- primitive ops */

/* torch.nn.Conv1d / F.conv1d with explicit left pad; reads outside [0,T) are 0.
 * x [B,Cin,T], w [Cout,Cin,K], y [B,Cout,Tout]. */
static void conv1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int K, int dil,
                   int pad_l, int Tout, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float* yo = y + ((size_t)b * Cout + co) * Tout;
      float bv = bias ? bias[co] : 0.f;
      for (int t = 0; t < Tout; ++t) yo[t] = bv;
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xi = x + ((size_t)b * Cin + ci) * T;
        const float* wk = w + ((size_t)co * Cin + ci) * K;
        for (int k = 0; k < K; ++k) {
          int off = k * dil - pad_l;
          int t0 = off < 0 ? -off : 0;
          int t1 = T - off < Tout ? T - off : Tout;
          float wv = wk[k];
          for (int t = t0; t < t1; ++t) yo[t] += wv * xi[t + off];
        }
      }
    }
}

/* depthwise Conv1d (groups == channels), "same" padding (modules.py:86-91) */
static void dwconv1d(const float* x, int B, int C, int T, const float* w, const float* bias, int K, int dil, float* y) {
  int pad = (K * dil - dil) / 2;
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float* xi = x + ((size_t)b * C + c) * T;
      float* yo = y + ((size_t)b * C + c) * T;
      for (int t = 0; t < T; ++t) {
        float a = bias[c];
        for (int k = 0; k < K; ++k) {
          int s = t + k * dil - pad;
          if (s >= 0 && s < T) a += w[c * K + k] * xi[s];
        }
        yo[t] = a;
      }
    }
}

/* torch.nn.ConvTranspose1d: x [B,Cin,T], w [Cin,Cout,K]; Tout=(T-1)*s-2p+K */
static void conv_transpose1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int K,
                             int stride, int pad, float* y) {
  int Tout = (T - 1) * stride - 2 * pad + K;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float* yo = y + ((size_t)b * Cout + co) * Tout;
      float bv = bias ? bias[co] : 0.f;
      for (int t = 0; t < Tout; ++t) yo[t] = bv;
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xi = x + ((size_t)b * Cin + ci) * T;
        const float* wk = w + ((size_t)ci * Cout + co) * K;
        for (int k = 0; k < K; ++k) {
          float wv = wk[k];
          /* t = i*stride + k - pad */
          for (int i = 0; i < T; ++i) {
            int t = i * stride + k - pad;
            if (t >= 0 && t < Tout) yo[t] += wv * xi[i];
          }
        }
      }
    }
}

/* modules.LayerNorm (modules.py:20-32): F.layer_norm over the channel dim, eps 1e-5 */
static void layer_norm_c(float* x, int B, int C, int T, const float* gamma, const float* beta) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      float* p = x + (size_t)b * C * T + t;
      float mean = 0.f;
      for (int c = 0; c < C; ++c) mean += p[(size_t)c * T];
      mean /= (float)C;
      float var = 0.f;
      for (int c = 0; c < C; ++c) { float d = p[(size_t)c * T] - mean; var += d * d; }
      var /= (float)C;
      float rstd = 1.0f / sqrtf(var + 1e-5f);
      for (int c = 0; c < C; ++c) p[(size_t)c * T] = (p[(size_t)c * T] - mean) * rstd * gamma[c] + beta[c];
    }
}

static inline float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
static inline float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
static inline float softplusf_(float v) { return v > 20.f ? v : log1pf(expf(v)); } /* F.softplus, threshold 20 */

static void mul_mask(float* x, int B, int C, int T, const int64_t* len) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      float* p = x + ((size_t)b * C + c) * T;
      for (int t = (int)len[b] < T ? (int)len[b] : T; t < T; ++t) p[t] = 0.f;
    }
}

/* ------------------------------------------ attentions.py: MHA / FFN / Encoder */

/* MultiHeadAttention.forward + attention (attentions.py:155-213) with the
 * relative-position terms (attentions.py:216-260) in band form (SURVEY.md A1,
 * verified bit-exact against the pad/reshape skew).  x [B,H,T] -> y [B,H,T]. */
static void mha(vits_model* m, const char* pfx, const float* x, int B, int H, int T, const int64_t* len, float* y) {
  const vits_hparams* hp = &m->hp;
  int nh = hp->n_heads, dk = H / nh, W = hp->window_size, NW = 2 * W + 1;
  const float* wq = tget(m, 3, H, H, 1, "%s.conv_q.weight", pfx);
  const float* bq = tget(m, 1, H, -1, -1, "%s.conv_q.bias", pfx);
  const float* wk = tget(m, 3, H, H, 1, "%s.conv_k.weight", pfx);
  const float* bk = tget(m, 1, H, -1, -1, "%s.conv_k.bias", pfx);
  const float* wv = tget(m, 3, H, H, 1, "%s.conv_v.weight", pfx);
  const float* bv = tget(m, 1, H, -1, -1, "%s.conv_v.bias", pfx);
  const float* wo = tget(m, 3, H, H, 1, "%s.conv_o.weight", pfx);
  const float* bo = tget(m, 1, H, -1, -1, "%s.conv_o.bias", pfx);
  const float* ek = tget(m, 3, 1, NW, dk, "%s.emb_rel_k", pfx); /* heads_share (attentions.py:142-145) */
  const float* ev = tget(m, 3, 1, NW, dk, "%s.emb_rel_v", pfx);
  if (m->missing) return;
  size_t n = (size_t)B * H * T;
  float* q = falloc(n); float* k = falloc(n); float* v = falloc(n); float* o = falloc(n);
  conv1d(x, B, H, T, wq, bq, H, 1, 1, 0, T, q);
  conv1d(x, B, H, T, wk, bk, H, 1, 1, 0, T, k);
  conv1d(x, B, H, T, wv, bv, H, 1, 1, 0, T, v);
  float scale = 1.0f / sqrtf((float)dk);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < nh; ++h) {
      float* s = (float*)malloc(sizeof(float) * (size_t)T);
      const float* qh = q + ((size_t)b * H + (size_t)h * dk) * T;
      const float* kh = k + ((size_t)b * H + (size_t)h * dk) * T;
      const float* vh = v + ((size_t)b * H + (size_t)h * dk) * T;
      float* oh = o + ((size_t)b * H + (size_t)h * dk) * T;
      int L = (int)len[b];
      for (int i = 0; i < T; ++i) {
        float mx = -INFINITY;
        for (int j = 0; j < T; ++j) {
          float a = 0.f;
          for (int d = 0; d < dk; ++d) a += (qh[(size_t)d * T + i] * scale) * kh[(size_t)d * T + j];
          int r = j - i;
          if (r >= -W && r <= W) { /* attentions.py:175-178 */
            float e = 0.f;
            for (int d = 0; d < dk; ++d) e += (qh[(size_t)d * T + i] * scale) * ek[(r + W) * dk + d];
            a += e;
          }
          if (!(i < L && j < L)) a = -1e4f; /* masked_fill(mask==0,-1e4) attentions.py:183 */
          s[j] = a;
          if (a > mx) mx = a;
        }
        float sum = 0.f;
        for (int j = 0; j < T; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
        float inv = 1.0f / sum;
        for (int d = 0; d < dk; ++d) {
          float a = 0.f;
          for (int j = 0; j < T; ++j) a += (s[j] * inv) * vh[(size_t)d * T + j];
          for (int r = -W; r <= W; ++r) { /* attentions.py:191-194 */
            int j = i + r;
            if (j >= 0 && j < T) a += (s[j] * inv) * ev[(r + W) * dk + d];
          }
          oh[(size_t)d * T + i] = a;
        }
      }
      free(s);
    }
  conv1d(o, B, H, T, wo, bo, H, 1, 1, 0, T, y);
  free(q); free(k); free(v); free(o);
}

/* FFN.forward (attentions.py:294-320), activation=None -> relu, "same" padding */
static void ffn(vits_model* m, const char* pfx, const float* x, int B, int H, int F, int K, int T, const int64_t* len,
                float* y) {
  const float* w1 = tget(m, 3, F, H, K, "%s.conv_1.weight", pfx);
  const float* b1 = tget(m, 1, F, -1, -1, "%s.conv_1.bias", pfx);
  const float* w2 = tget(m, 3, H, F, K, "%s.conv_2.weight", pfx);
  const float* b2 = tget(m, 1, H, -1, -1, "%s.conv_2.bias", pfx);
  if (m->missing) return;
  float* xm = falloc((size_t)B * H * T);
  memcpy(xm, x, sizeof(float) * (size_t)B * H * T);
  mul_mask(xm, B, H, T, len);
  float* h = falloc((size_t)B * F * T);
  conv1d(xm, B, H, T, w1, b1, F, K, 1, (K - 1) / 2, T, h);
  for (size_t i = 0; i < (size_t)B * F * T; ++i) h[i] = h[i] > 0.f ? h[i] : 0.f;
  mul_mask(h, B, F, T, len);
  conv1d(h, B, F, T, w2, b2, H, K, 1, (K - 1) / 2, T, y);
  mul_mask(y, B, H, T, len);
  free(xm); free(h);
}

/* attentions.Encoder.forward (attentions.py:48-65).  x in/out [B,H,T].
 * g [B,G] or NULL (speaker add before layer cond_layer, attentions.py:52-56). */
static void encoder(vits_model* m, const char* pfx, float* x, int B, int H, int F, int K, int n_layers, int T,
                    const int64_t* len, const float* g, int G, int cond_layer) {
  size_t n = (size_t)B * H * T;
  float* y = falloc(n);
  char sub[200];
  mul_mask(x, B, H, T, len);
  for (int i = 0; i < n_layers && !m->missing; ++i) {
    if (g && i == cond_layer) {
      const float* lw = tget(m, 2, H, G, -1, "%s.spk_emb_linear.weight", pfx);
      const float* lb = tget(m, 1, H, -1, -1, "%s.spk_emb_linear.bias", pfx);
      if (m->missing) break;
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < H; ++c) {
          float a = lb[c];
          for (int j = 0; j < G; ++j) a += lw[c * G + j] * g[b * G + j];
          float* p = x + ((size_t)b * H + c) * T;
          for (int t = 0; t < T; ++t) p[t] += a;
        }
      mul_mask(x, B, H, T, len);
    }
    snprintf(sub, sizeof sub, "%s.attn_layers.%d", pfx, i);
    mha(m, sub, x, B, H, T, len, y);
    for (size_t e = 0; e < n; ++e) x[e] += y[e];
    layer_norm_c(x, B, H, T, tget(m, 1, H, -1, -1, "%s.norm_layers_1.%d.gamma", pfx, i),
                 tget(m, 1, H, -1, -1, "%s.norm_layers_1.%d.beta", pfx, i));
    snprintf(sub, sizeof sub, "%s.ffn_layers.%d", pfx, i);
    ffn(m, sub, x, B, H, F, K, T, len, y);
    for (size_t e = 0; e < n; ++e) x[e] += y[e];
    layer_norm_c(x, B, H, T, tget(m, 1, H, -1, -1, "%s.norm_layers_2.%d.gamma", pfx, i),
                 tget(m, 1, H, -1, -1, "%s.norm_layers_2.%d.beta", pfx, i));
  }
  mul_mask(x, B, H, T, len);
  free(y);
}

/* speaker embedding g = emb_g(sid) (models.py:1680-1681); out [B,G] */
static int speaker_g(vits_model* m, const int64_t* sid, int B, float* g) {
  const vits_hparams* hp = &m->hp;
  int G = hp->gin_channels;
  if (hp->n_speakers <= 1) { memset(g, 0, sizeof(float) * (size_t)B * G); return VITS_OK; }
  const float* e = tget(m, 2, hp->n_speakers, G, -1, "emb_g.weight");
  if (!e) return VITS_ERR_BLOB;
  for (int b = 0; b < B; ++b) {
    int64_t s = sid ? sid[b] : 0;
    if (s < 0 || s >= hp->n_speakers) return fail(VITS_ERR_ARG, "speaker id %lld out of range", (long long)s);
    memcpy(g + (size_t)b * G, e + (size_t)s * G, sizeof(float) * G);
  }
  return VITS_OK;
}

/* ------------------------------------------------------------ a2 TextEncoder */

/* bert (optional, [B, bert_dim, T]): the "bert" feed of the BERT-conditioned flavours (vosk_tts/synth.py:88-99,113-120).  Their
 * text encoder is not in the reference tree; the build defines the wiring as a 1x1 projection added to the
 * scaled embedding, x = (emb(ids) * sqrt(H) + bert_proj(bert)) * mask, and this function is its CPU statement. */
static int text_encoder_impl(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T,
                             const int64_t* sid, const float* bert, float* x, float* m_p, float* logs_p);
int API(stage_text_encoder)(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T,
                            const int64_t* sid, float* x, float* m_p, float* logs_p) {
  return text_encoder_impl(m, ids, lengths, B, T, sid, NULL, x, m_p, logs_p);
}
static int text_encoder_impl(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T,
                             const int64_t* sid, const float* bert, float* x, float* m_p, float* logs_p) {
  if (!m || !ids || !lengths || !x || !m_p || !logs_p || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const vits_hparams* hp = &m->hp;
  int H = hp->hidden_channels, I = hp->inter_channels, G = hp->gin_channels;
  m->missing = 0;
  const float* emb = tget(m, 2, hp->n_vocab, H, -1, "enc_p.emb.weight");
  if (!emb) return VITS_ERR_BLOB;
  for (int b = 0; b < B; ++b) {
    if (lengths[b] < 0 || lengths[b] > T) return fail(VITS_ERR_ARG, "length out of range");
    for (int t = 0; t < T; ++t) {
      int64_t id = ids[(size_t)b * T + t];
      if (t >= lengths[b]) id = 0;
      if (id < 0 || id >= hp->n_vocab) return fail(VITS_ERR_ARG, "token id %lld out of range", (long long)id);
      float sc = sqrtf((float)H); /* models.py:318 */
      for (int c = 0; c < H; ++c) x[((size_t)b * H + c) * T + t] = emb[(size_t)id * H + c] * sc;
    }
  }
  if (hp->bert_dim > 0) {
    if (!bert) return fail(VITS_ERR_ARG, "this voice is BERT-conditioned: the bert feed is required");
    const float* bw = tget(m, 3, H, hp->bert_dim, 1, "enc_p.bert_proj.weight");
    const float* bb = tget(m, 1, H, -1, -1, "enc_p.bert_proj.bias");
    if (m->missing) return VITS_ERR_BLOB;
    float* pr = falloc((size_t)B * H * T);
    conv1d(bert, B, hp->bert_dim, T, bw, bb, H, 1, 1, 0, T, pr);
    mul_mask(pr, B, H, T, lengths);
    for (size_t i = 0; i < (size_t)B * H * T; ++i) x[i] += pr[i];
    free(pr);
  } else if (bert) {
    return fail(VITS_ERR_ARG, "the bert feed was given but this voice has no BERT projection");
  }
  float* g = falloc((size_t)B * G);
  int rc = speaker_g(m, sid, B, g);
  if (rc) { free(g); return rc; }
  int use_g = hp->enc_cond_layer >= 0 && G > 0 && hp->n_speakers > 1;
  encoder(m, "enc_p.encoder", x, B, H, hp->filter_channels, hp->kernel_size, hp->n_layers, T, lengths,
          use_g ? g : NULL, G, hp->enc_cond_layer);
  free(g);
  const float* pw = tget(m, 3, 2 * I, H, 1, "enc_p.proj.weight");
  const float* pb = tget(m, 1, 2 * I, -1, -1, "enc_p.proj.bias");
  if (m->missing) return VITS_ERR_BLOB;
  float* stats = falloc((size_t)B * 2 * I * T);
  conv1d(x, B, H, T, pw, pb, 2 * I, 1, 1, 0, T, stats);
  mul_mask(stats, B, 2 * I, T, lengths);
  for (int b = 0; b < B; ++b) {
    memcpy(m_p + (size_t)b * I * T, stats + (size_t)b * 2 * I * T, sizeof(float) * (size_t)I * T);
    memcpy(logs_p + (size_t)b * I * T, stats + ((size_t)b * 2 * I + I) * T, sizeof(float) * (size_t)I * T);
  }
  free(stats);
  return VITS_OK;
}

/* ------------------------------------------------- a6-a9 duration predictor */

/* DDSConv.forward (modules.py:96-108).  x in/out [B,C,T]; g [B,C,T] or NULL */
static void ddsconv(vits_model* m, const char* pfx, float* x, int B, int C, int K, int n_layers, int T,
                    const int64_t* len, const float* g) {
  size_t n = (size_t)B * C * T;
  if (g) for (size_t i = 0; i < n; ++i) x[i] += g[i];
  float* xm = falloc(n); float* y = falloc(n); float* y2 = falloc(n);
  int dil = 1;
  for (int i = 0; i < n_layers && !m->missing; ++i) {
    memcpy(xm, x, sizeof(float) * n);
    mul_mask(xm, B, C, T, len);
    const float* sw = tget(m, 3, C, 1, K, "%s.convs_sep.%d.weight", pfx, i);
    const float* sb = tget(m, 1, C, -1, -1, "%s.convs_sep.%d.bias", pfx, i);
    const float* pw = tget(m, 3, C, C, 1, "%s.convs_1x1.%d.weight", pfx, i);
    const float* pb = tget(m, 1, C, -1, -1, "%s.convs_1x1.%d.bias", pfx, i);
    const float* g1 = tget(m, 1, C, -1, -1, "%s.norms_1.%d.gamma", pfx, i);
    const float* b1 = tget(m, 1, C, -1, -1, "%s.norms_1.%d.beta", pfx, i);
    const float* g2 = tget(m, 1, C, -1, -1, "%s.norms_2.%d.gamma", pfx, i);
    const float* b2 = tget(m, 1, C, -1, -1, "%s.norms_2.%d.beta", pfx, i);
    if (m->missing) break;
    dwconv1d(xm, B, C, T, sw, sb, K, dil, y);
    layer_norm_c(y, B, C, T, g1, b1);
    for (size_t e = 0; e < n; ++e) y[e] = gelu_erf(y[e]);
    conv1d(y, B, C, T, pw, pb, C, 1, 1, 0, T, y2);
    layer_norm_c(y2, B, C, T, g2, b2);
    for (size_t e = 0; e < n; ++e) x[e] += gelu_erf(y2[e]);
    dil *= K; /* dilation = kernel_size ** i (modules.py:84) */
  }
  mul_mask(x, B, C, T, len);
  free(xm); free(y); free(y2);
}

/* rational_quadratic_spline(inverse=True) inside unconstrained_... with linear
 * tails (transforms.py:55-177), one element.  uw,uh: [nb] already divided by
 * sqrt(filter_channels) (modules.py:372-373); ud: [nb-1]. */
static float rqs_inverse(float y, const float* uw, const float* uh, const float* ud, int nb, float bound) {
  if (!(y >= -bound && y <= bound)) return y; /* transforms.py:65-77 */
  const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
  float w[32], cw[33], h[32], ch[33], d[33];
  /* widths (transforms.py:117-123) */
  float mx = uw[0];
  for (int i = 1; i < nb; ++i) if (uw[i] > mx) mx = uw[i];
  float sum = 0.f;
  for (int i = 0; i < nb; ++i) { w[i] = expf(uw[i] - mx); sum += w[i]; }
  float acc = 0.f;
  cw[0] = -bound;
  for (int i = 0; i < nb; ++i) {
    float wi = min_w + (1.f - min_w * nb) * (w[i] / sum);
    acc += wi;
    cw[i + 1] = (bound - (-bound)) * acc + (-bound);
  }
  cw[0] = -bound; cw[nb] = bound;
  for (int i = 0; i < nb; ++i) w[i] = cw[i + 1] - cw[i];
  /* derivatives (transforms.py:68-71,125) */
  float cst = (float)log(exp(1.0 - (double)min_d) - 1.0);
  for (int i = 0; i <= nb; ++i) {
    float u = (i == 0 || i == nb) ? cst : ud[i - 1];
    d[i] = min_d + softplusf_(u);
  }
  /* heights (transforms.py:127-134) */
  mx = uh[0];
  for (int i = 1; i < nb; ++i) if (uh[i] > mx) mx = uh[i];
  sum = 0.f;
  for (int i = 0; i < nb; ++i) { h[i] = expf(uh[i] - mx); sum += h[i]; }
  acc = 0.f;
  ch[0] = -bound;
  for (int i = 0; i < nb; ++i) {
    float hi = min_h + (1.f - min_h * nb) * (h[i] / sum);
    acc += hi;
    ch[i + 1] = (bound - (-bound)) * acc + (-bound);
  }
  ch[0] = -bound; ch[nb] = bound;
  for (int i = 0; i < nb; ++i) h[i] = ch[i + 1] - ch[i];
  /* searchsorted on cumheights, last knot +1e-6 (transforms.py:47-52,136-137) */
  int bin = -1;
  for (int i = 0; i <= nb; ++i) {
    float loc = ch[i] + (i == nb ? 1e-6f : 0.f);
    if (y >= loc) bin++;
  }
  if (bin < 0) bin = 0;
  if (bin > nb - 1) bin = nb - 1;
  float in_cw = cw[bin], in_w = w[bin], in_ch = ch[bin], in_h = h[bin];
  float delta = h[bin] / w[bin];
  float d0 = d[bin], d1 = d[bin + 1];
  /* transforms.py:152-167 */
  float t1 = (y - in_ch) * (d0 + d1 - 2.f * delta);
  float a = t1 + in_h * (delta - d0);
  float b = in_h * d0 - t1;
  float c = -delta * (y - in_ch);
  float disc = b * b - 4.f * a * c;
  float root = (2.f * c) / (-b - sqrtf(disc));
  return root * in_w + in_cw;
}

/* ConvFlow.forward(reverse=True) (modules.py:363-390).  z [B,2,T] in place; c = conditioning [B,D,T] */
static void convflow_reverse(vits_model* m, const char* pfx, float* z, int B, int T, const int64_t* len, const float* c) {
  const vits_hparams* hp = &m->hp;
  int D = hp->dp_filter_channels, nb = hp->dp_num_bins, P = 3 * nb - 1;
  const float* pw = tget(m, 3, D, 1, 1, "%s.pre.weight", pfx);
  const float* pb = tget(m, 1, D, -1, -1, "%s.pre.bias", pfx);
  const float* jw = tget(m, 3, P, D, 1, "%s.proj.weight", pfx);
  const float* jb = tget(m, 1, P, -1, -1, "%s.proj.bias", pfx);
  if (m->missing) return;
  float* h = falloc((size_t)B * D * T);
  for (int b = 0; b < B; ++b)
    for (int ch = 0; ch < D; ++ch)
      for (int t = 0; t < T; ++t) h[((size_t)b * D + ch) * T + t] = pw[ch] * z[((size_t)b * 2 + 0) * T + t] + pb[ch];
  char sub[200];
  snprintf(sub, sizeof sub, "%s.convs", pfx);
  ddsconv(m, sub, h, B, D, hp->dp_kernel_size, hp->dp_dds_layers, T, len, c);
  float* pr = falloc((size_t)B * P * T);
  conv1d(h, B, D, T, jw, jb, P, 1, 1, 0, T, pr);
  mul_mask(pr, B, P, T, len);
  float inv = 1.0f / sqrtf((float)D);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      float uw[32], uh[32], ud[32];
      for (int i = 0; i < nb; ++i) {
        uw[i] = pr[((size_t)b * P + i) * T + t] * inv;
        uh[i] = pr[((size_t)b * P + nb + i) * T + t] * inv;
      }
      for (int i = 0; i < nb - 1; ++i) ud[i] = pr[((size_t)b * P + 2 * nb + i) * T + t];
      float* x1 = &z[((size_t)b * 2 + 1) * T + t];
      *x1 = rqs_inverse(*x1, uw, uh, ud, nb, hp->dp_tail_bound);
    }
  mul_mask(z, B, 2, T, len);
  free(h); free(pr);
}

/* StochasticDurationPredictor.forward(reverse=True) (models.py:56-63,93-101) */
int API(stage_duration)(vits_model* m, const float* x, const int64_t* lengths, int32_t B, int32_t T,
                        const int64_t* sid, const float* noise, float noise_scale_w, float* logw) {
  if (!m || !x || !lengths || !noise || !logw || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const vits_hparams* hp = &m->hp;
  int H = hp->hidden_channels, D = hp->dp_filter_channels, G = hp->gin_channels;
  m->missing = 0;
  const float* pw = tget(m, 3, D, H, 1, "dp.pre.weight");
  const float* pb = tget(m, 1, D, -1, -1, "dp.pre.bias");
  const float* jw = tget(m, 3, D, D, 1, "dp.proj.weight");
  const float* jb = tget(m, 1, D, -1, -1, "dp.proj.bias");
  if (m->missing) return VITS_ERR_BLOB;
  float* h = falloc((size_t)B * D * T);
  conv1d(x, B, H, T, pw, pb, D, 1, 1, 0, T, h);
  if (G > 0 && hp->n_speakers > 1) { /* x = x + cond(g), models.py:58-60 */
    float* g = falloc((size_t)B * G);
    int rc = speaker_g(m, sid, B, g);
    if (rc) { free(g); free(h); return rc; }
    const float* cw = tget(m, 3, D, G, 1, "dp.cond.weight");
    const float* cb = tget(m, 1, D, -1, -1, "dp.cond.bias");
    if (m->missing) { free(g); free(h); return VITS_ERR_BLOB; }
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < D; ++c) {
        float a = cb[c];
        for (int j = 0; j < G; ++j) a += cw[c * G + j] * g[b * G + j];
        float* p = h + ((size_t)b * D + c) * T;
        for (int t = 0; t < T; ++t) p[t] += a;
      }
    free(g);
  }
  ddsconv(m, "dp.convs", h, B, D, hp->dp_kernel_size, hp->dp_dds_layers, T, lengths, NULL);
  float* c = falloc((size_t)B * D * T);
  conv1d(h, B, D, T, jw, jb, D, 1, 1, 0, T, c);
  mul_mask(c, B, D, T, lengths);
  free(h);
  float* z = falloc((size_t)B * 2 * T);
  for (size_t i = 0; i < (size_t)B * 2 * T; ++i) z[i] = noise[i] * noise_scale_w; /* models.py:96 */
  /* reversed(flows)[:-2] + [flows[0]]: Flip, CF_{n}, Flip, ..., CF_2, Flip, EA (models.py:94-95) */
  char sub[64];
  for (int k = hp->dp_n_flows; k >= 2 && !m->missing; --k) {
    for (int b = 0; b < B; ++b) /* Flip (modules.py:270-277) */
      for (int t = 0; t < T; ++t) {
        float a = z[((size_t)b * 2) * T + t];
        z[((size_t)b * 2) * T + t] = z[((size_t)b * 2 + 1) * T + t];
        z[((size_t)b * 2 + 1) * T + t] = a;
      }
    snprintf(sub, sizeof sub, "dp.flows.%d", 2 * k - 1);
    convflow_reverse(m, sub, z, B, T, lengths, c);
  }
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      float a = z[((size_t)b * 2) * T + t];
      z[((size_t)b * 2) * T + t] = z[((size_t)b * 2 + 1) * T + t];
      z[((size_t)b * 2 + 1) * T + t] = a;
    }
  const float* em = tget(m, 2, 2, 1, -1, "dp.flows.0.m");
  const float* el = tget(m, 2, 2, 1, -1, "dp.flows.0.logs");
  if (m->missing) { free(c); free(z); return VITS_ERR_BLOB; }
  for (int b = 0; b < B; ++b) /* ElementwiseAffine reverse (modules.py:293-295); logw = z0 */
    for (int t = 0; t < T; ++t) {
      float v = (z[((size_t)b * 2) * T + t] - em[0]) * expf(-el[0]);
      logw[(size_t)b * T + t] = t < lengths[b] ? v : 0.f;
    }
  free(c); free(z);
  return VITS_OK;
}

/* --------------------------------------- a10/a11 length regulator + prior */

int API(stage_regulate)(vits_model* m, const float* logw, const int32_t* forced, const int64_t* lengths, int32_t B,
                        int32_t T, float length_scale, const float* m_p, const float* logs_p, const float* noise,
                        float noise_scale, int32_t Tcap, int32_t* durations, int64_t* y_lengths, float* z_p) {
  if (!m || !lengths || !durations || !y_lengths || (!logw && !forced)) return fail(VITS_ERR_ARG, "bad argument");
  int I = m->hp.inter_channels;
  for (int b = 0; b < B; ++b) {
    int64_t tot = 0;
    for (int t = 0; t < T; ++t) {
      int32_t d;
      if (t >= lengths[b]) d = 0;
      else if (forced) d = forced[(size_t)b * T + t];
      else d = (int32_t)ceilf(expf(logw[(size_t)b * T + t]) * length_scale); /* models.py:1689-1690 */
      if (d < 0) d = 0;
      durations[(size_t)b * T + t] = d;
      tot += d;
    }
    y_lengths[b] = tot < 1 ? 1 : tot; /* clamp_min(...,1) models.py:1691 */
  }
  if (!z_p) return VITS_OK;
  if (!m_p || !logs_p) return fail(VITS_ERR_ARG, "m_p/logs_p required");
  for (int b = 0; b < B; ++b) {
    if (y_lengths[b] > Tcap) return fail(VITS_ERR_ARG, "T_y %lld exceeds capacity %d", (long long)y_lengths[b], Tcap);
    /* generate_path (commons.py:128-143): frame f belongs to token j iff cum[j-1] <= f < cum[j] */
    int j = 0;
    int64_t cum = durations[(size_t)b * T];
    int64_t dsum = 0;
    for (int t = 0; t < T; ++t) dsum += durations[(size_t)b * T + t];
    for (int f = 0; f < Tcap; ++f) {
      int tok = -1;
      if (f < dsum && f < y_lengths[b]) {
        while (f >= cum && j + 1 < T) { ++j; cum += durations[(size_t)b * T + j]; }
        tok = j;
      }
      for (int c = 0; c < I; ++c) {
        float mu = tok >= 0 ? m_p[((size_t)b * I + c) * T + tok] : 0.f;
        float ls = tok >= 0 ? logs_p[((size_t)b * I + c) * T + tok] : 0.f;
        float e = noise ? noise[((size_t)b * I + c) * Tcap + f] : 0.f;
        z_p[((size_t)b * I + c) * Tcap + f] = mu + e * expf(ls) * noise_scale; /* models.py:1700 */
      }
    }
  }
  return VITS_OK;
}

/* ------------------------------------------------------------ a12-a14 flow */

/* WN.forward (modules.py:148-176) + fused_add_tanh_sigmoid_multiply (commons.py:100-107).
 * x in [B,H,T] (destroyed), out [B,H,T]; g [B,G] or NULL */
static void wn(vits_model* m, const char* pfx, float* x, int B, int H, int T, const int64_t* len, const float* g, int G,
               float* out) {
  const vits_hparams* hp = &m->hp;
  int L = hp->flow_wn_layers, K = hp->flow_kernel_size;
  size_t n = (size_t)B * H * T;
  memset(out, 0, sizeof(float) * n);
  float* gl = NULL;
  if (g) {
    const float* cw = tget(m, 3, 2 * H * L, G, 1, "%s.cond_layer.weight", pfx);
    const float* cb = tget(m, 1, 2 * H * L, -1, -1, "%s.cond_layer.bias", pfx);
    if (m->missing) return;
    gl = falloc((size_t)B * 2 * H * L);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < 2 * H * L; ++c) {
        float a = cb[c];
        for (int j = 0; j < G; ++j) a += cw[(size_t)c * G + j] * g[b * G + j];
        gl[(size_t)b * 2 * H * L + c] = a;
      }
  }
  float* xin = falloc((size_t)B * 2 * H * T);
  float* acts = falloc(n);
  float* rs = falloc((size_t)B * 2 * H * T);
  int dil = 1;
  for (int i = 0; i < L && !m->missing; ++i) {
    const float* iw = tget(m, 3, 2 * H, H, K, "%s.in_layers.%d.weight", pfx, i);
    const float* ib = tget(m, 1, 2 * H, -1, -1, "%s.in_layers.%d.bias", pfx, i);
    int RS = i < L - 1 ? 2 * H : H;
    const float* rw = tget(m, 3, RS, H, 1, "%s.res_skip_layers.%d.weight", pfx, i);
    const float* rb = tget(m, 1, RS, -1, -1, "%s.res_skip_layers.%d.bias", pfx, i);
    if (m->missing) break;
    conv1d(x, B, H, T, iw, ib, 2 * H, K, dil, (K * dil - dil) / 2, T, xin);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < H; ++c) {
        float ga = gl ? gl[(size_t)b * 2 * H * L + (size_t)i * 2 * H + c] : 0.f;
        float gb = gl ? gl[(size_t)b * 2 * H * L + (size_t)i * 2 * H + H + c] : 0.f;
        const float* pa = xin + ((size_t)b * 2 * H + c) * T;
        const float* pb2 = xin + ((size_t)b * 2 * H + H + c) * T;
        float* po = acts + ((size_t)b * H + c) * T;
        for (int t = 0; t < T; ++t) po[t] = tanhf(pa[t] + ga) * sigmoidf_(pb2[t] + gb);
      }
    conv1d(acts, B, H, T, rw, rb, RS, 1, 1, 0, T, rs);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < H; ++c) {
        float* px = x + ((size_t)b * H + c) * T;
        float* po = out + ((size_t)b * H + c) * T;
        int Lb = (int)len[b];
        if (i < L - 1) {
          const float* pr = rs + ((size_t)b * 2 * H + c) * T;
          const float* ps = rs + ((size_t)b * 2 * H + H + c) * T;
          for (int t = 0; t < T; ++t) { px[t] = t < Lb ? px[t] + pr[t] : 0.f; po[t] += ps[t]; }
        } else {
          const float* ps = rs + ((size_t)b * H + c) * T;
          for (int t = 0; t < T; ++t) po[t] += ps[t];
        }
      }
    dil *= hp->flow_dilation_rate;
  }
  mul_mask(out, B, H, T, len);
  free(xin); free(acts); free(rs); free(gl);
}

/* ResidualCouplingTransformersLayer2.forward(reverse=True) (models.py:374-393), mean_only */
static void coupling_reverse(vits_model* m, const char* pfx, float* x, int B, int T, const int64_t* len, const float* g) {
  const vits_hparams* hp = &m->hp;
  int I = hp->inter_channels, half = I / 2, H = hp->hidden_channels, G = hp->gin_channels;
  const float* pw = tget(m, 3, H, half, 1, "%s.pre.weight", pfx);
  const float* pb = tget(m, 1, H, -1, -1, "%s.pre.bias", pfx);
  const float* ow = tget(m, 3, half, H, 1, "%s.post.weight", pfx);
  const float* ob = tget(m, 1, half, -1, -1, "%s.post.bias", pfx);
  if (m->missing) return;
  size_t n = (size_t)B * H * T;
  float* x0 = falloc((size_t)B * half * T);
  for (int b = 0; b < B; ++b) memcpy(x0 + (size_t)b * half * T, x + (size_t)b * I * T, sizeof(float) * (size_t)half * T);
  float* h = falloc(n);
  conv1d(x0, B, half, T, pw, pb, H, 1, 1, 0, T, h);
  mul_mask(h, B, H, T, len);
  float* e = falloc(n);
  memcpy(e, h, sizeof(float) * n);
  char sub[200];
  snprintf(sub, sizeof sub, "%s.pre_transformer", pfx);
  encoder(m, sub, e, B, H, H, hp->flow_kernel_size, 1, T, len, NULL, 0, -1);
  for (size_t i = 0; i < n; ++i) h[i] += e[i]; /* models.py:377 */
  snprintf(sub, sizeof sub, "%s.enc", pfx);
  wn(m, sub, h, B, H, T, len, (G > 0 && hp->n_speakers > 1) ? g : NULL, G, e);
  float* mu = falloc((size_t)B * half * T);
  conv1d(e, B, H, T, ow, ob, half, 1, 1, 0, T, mu);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < half; ++c) {
      float* p1 = x + ((size_t)b * I + half + c) * T;
      const float* pm = mu + ((size_t)b * half + c) * T;
      for (int t = 0; t < T; ++t) p1[t] = t < len[b] ? (p1[t] - pm[t]) : 0.f; /* (x1 - m*mask)*mask, logs == 0 */
    }
  free(x0); free(h); free(e); free(mu);
}

static void flip_channels(float* x, int B, int C, int T) {
  float* tmp = falloc((size_t)T);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C / 2; ++c) {
      float* a = x + ((size_t)b * C + c) * T;
      float* z = x + ((size_t)b * C + (C - 1 - c)) * T;
      memcpy(tmp, a, sizeof(float) * T); memcpy(a, z, sizeof(float) * T); memcpy(z, tmp, sizeof(float) * T);
    }
  free(tmp);
}

/* ResidualCouplingTransformersBlock.forward(reverse=True) (models.py:750-757) */
int API(stage_flow)(vits_model* m, const float* z_p, const int64_t* y_lengths, int32_t B, int32_t T, const int64_t* sid,
                    float* z) {
  if (!m || !z_p || !y_lengths || !z || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const vits_hparams* hp = &m->hp;
  int I = hp->inter_channels, G = hp->gin_channels;
  m->missing = 0;
  float* g = falloc((size_t)B * (G > 0 ? G : 1));
  int rc = speaker_g(m, sid, B, g);
  if (rc) { free(g); return rc; }
  if (z != z_p) memcpy(z, z_p, sizeof(float) * (size_t)B * I * T);
  char sub[64];
  for (int f = hp->flow_n_flows - 1; f >= 0 && !m->missing; --f) {
    flip_channels(z, B, I, T);
    snprintf(sub, sizeof sub, "flow.flows.%d", 2 * f);
    coupling_reverse(m, sub, z, B, T, y_lengths, g);
  }
  free(g);
  return m->missing ? VITS_ERR_BLOB : VITS_OK;
}

/* --------------------------------------------------------- a15-a20 decoder */

static double bessel_i0(double x) {
  double s = 1.0, term = 1.0, q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) { term *= q / ((double)k * k); s += term; if (term < 1e-18 * s) break; }
  return s;
}

/* OnnxSTFT.__init__ inverse_basis (stft.py:191-214): pinv(scale*[Re F; Im F]).T * hann.
 * For the [2(N/2+1)] x N real DFT matrix the Moore-Penrose pinv is the irfft
 * synthesis matrix (it is a left inverse that vanishes on the two all-zero rows
 * Im F_0, Im F_{N/2}, i.e. on range(F)^perp). */
static void build_istft_basis(vits_model* m) {
  int N = m->hp.istft_n_fft, hop = m->hp.istft_hop, cut = N / 2 + 1;
  double scale = (double)N / hop;
  m->istft_basis = falloc((size_t)2 * cut * N);
  for (int n = 0; n < N; ++n) {
    double win = 0.5 - 0.5 * cos(2.0 * PI_D * n / N); /* scipy get_window('hann', N, fftbins=True) */
    for (int k = 0; k < cut; ++k) {
      double wk = (k == 0 || k == N / 2) ? 1.0 : 2.0;
      double th = 2.0 * PI_D * k * n / N;
      m->istft_basis[(size_t)k * N + n] = (float)((float)(wk * cos(th) / N / scale) * (float)win);
      m->istft_basis[(size_t)(cut + k) * N + n] = (float)((float)(-wk * sin(th) / N / scale) * (float)win);
    }
  }
}

/* PQMF synthesis filter (pqmf.py:15-43 design_prototype_filter, :64-75) */
static void build_pqmf(vits_model* m) {
  int taps = m->hp.pqmf_taps, S = m->hp.subbands, L = taps + 1;
  double cutoff = (double)m->hp.pqmf_cutoff, beta = (double)m->hp.pqmf_beta;
  double* h = (double*)malloc(sizeof(double) * L);
  double omega_c = PI_D * cutoff;
  for (int n = 0; n < L; ++n) {
    double xx = n - 0.5 * taps;
    double hi = (n == taps / 2) ? cutoff : sin(omega_c * xx) / (PI_D * xx);
    double r = (n - (L - 1) / 2.0) / ((L - 1) / 2.0);
    double arg = 1.0 - r * r;
    double w = bessel_i0(beta * sqrt(arg < 0 ? 0 : arg)) / bessel_i0(beta); /* scipy.signal.windows.kaiser */
    h[n] = hi * w;
  }
  m->pqmf_syn = falloc((size_t)S * L);
  for (int k = 0; k < S; ++k)
    for (int n = 0; n < L; ++n) {
      double sign = (k % 2 == 0) ? 1.0 : -1.0;
      m->pqmf_syn[(size_t)k * L + n] =
          (float)(2.0 * h[n] * cos((2 * k + 1) * (PI_D / (2.0 * S)) * (n - ((taps - 1) / 2.0)) - sign * PI_D / 4.0));
    }
  free(h);
}

static void lrelu_inplace(float* x, size_t n, float slope) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}

/* ResBlock1.forward (modules.py:210-223): x in place [B,C,T] */
static void resblock1(vits_model* m, int idx, float* x, int B, int C, int T, int K, const int32_t* dils, int nd) {
  size_t n = (size_t)B * C * T;
  float* xt = falloc(n); float* y = falloc(n);
  for (int i = 0; i < nd && !m->missing; ++i) {
    const float* w1 = tget(m, 3, C, C, K, "dec.resblocks.%d.convs1.%d.weight", idx, i);
    const float* b1 = tget(m, 1, C, -1, -1, "dec.resblocks.%d.convs1.%d.bias", idx, i);
    const float* w2 = tget(m, 3, C, C, K, "dec.resblocks.%d.convs2.%d.weight", idx, i);
    const float* b2 = tget(m, 1, C, -1, -1, "dec.resblocks.%d.convs2.%d.bias", idx, i);
    if (m->missing) break;
    memcpy(xt, x, sizeof(float) * n);
    lrelu_inplace(xt, n, 0.1f); /* LRELU_SLOPE modules.py:17 */
    int d = dils[i];
    conv1d(xt, B, C, T, w1, b1, C, K, d, (K * d - d) / 2, T, y); /* get_padding commons.py:14-15 */
    lrelu_inplace(y, n, 0.1f);
    conv1d(y, B, C, T, w2, b2, C, K, 1, (K - 1) / 2, T, xt);
    for (size_t e = 0; e < n; ++e) x[e] += xt[e];
  }
  free(xt); free(y);
}

/* Multiband_iSTFT_Generator.forward (models.py:1016-1054) / Generator.forward (models.py:872-891) */
int API(stage_decoder)(vits_model* m, const float* z, int32_t B, int32_t T, const int64_t* sid, float* audio, float* audio_mb) {
  if (!m || !z || !audio || B <= 0 || T <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const vits_hparams* hp = &m->hp;
  int I = hp->inter_channels, C = hp->dec_initial_channel;
  m->missing = 0;
  const float* w = tget(m, 3, C, I, 7, "dec.conv_pre.weight");
  const float* bi = tget(m, 1, C, -1, -1, "dec.conv_pre.bias");
  if (m->missing) return VITS_ERR_BLOB;
  float* x = falloc((size_t)B * C * T);
  conv1d(z, B, I, T, w, bi, C, 7, 1, 3, T, x);
  if (hp->dec_type == 1 && hp->gin_channels > 0 && hp->n_speakers > 1) { /* x = x + cond(g) (models.py:873-875) */
    int G = hp->gin_channels;
    const float* cw = tget(m, 3, C, G, 1, "dec.cond.weight");
    const float* cb = tget(m, 1, C, -1, -1, "dec.cond.bias");
    float* g = falloc((size_t)B * G);
    int rc = m->missing ? VITS_ERR_BLOB : speaker_g(m, sid, B, g);
    if (rc) { free(g); free(x); return rc; }
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) {
        float a = cb[c];
        for (int j = 0; j < G; ++j) a += cw[(size_t)c * G + j] * g[(size_t)b * G + j];
        float* p = x + ((size_t)b * C + c) * T;
        for (int t = 0; t < T; ++t) p[t] += a;
      }
    free(g);
  }
  int Tc = T;
  for (int i = 0; i < hp->n_ups && !m->missing; ++i) {
    int u = hp->up_rates[i], k = hp->up_kernels[i], Co = C / 2;
    const float* uw = tget(m, 3, C, Co, k, "dec.ups.%d.weight", i);
    const float* ub = tget(m, 1, Co, -1, -1, "dec.ups.%d.bias", i);
    if (m->missing) break;
    lrelu_inplace(x, (size_t)B * C * Tc, 0.1f);
    int To = (Tc - 1) * u - 2 * ((k - u) / 2) + k;
    float* y = falloc((size_t)B * Co * To);
    conv_transpose1d(x, B, C, Tc, uw, ub, Co, k, u, (k - u) / 2, y);
    free(x);
    C = Co; Tc = To;
    size_t n = (size_t)B * C * Tc;
    float* xs = falloc(n); float* r = falloc(n);
    for (int j = 0; j < hp->n_resk; ++j) { /* models.py:1030-1036 */
      memcpy(r, y, sizeof(float) * n);
      resblock1(m, i * hp->n_resk + j, r, B, C, Tc, hp->res_kernels[j], hp->res_dilations[j], hp->n_resd);
      for (size_t e = 0; e < n; ++e) xs[e] += r[e];
    }
    for (size_t e = 0; e < n; ++e) xs[e] /= (float)hp->n_resk;
    free(r); free(y);
    x = xs;
  }
  if (m->missing) { free(x); return VITS_ERR_BLOB; }
  size_t n = (size_t)B * C * Tc;
  if (hp->dec_type == 1) { /* plain HiFi-GAN tail: leaky_relu -> conv_post -> tanh (models.py:887-889) */
    const float* pw = tget(m, 3, 1, C, 7, "dec.conv_post.weight");
    /* VITS' Generator has no conv_post bias (models.py:866); the HiFi-GAN bundled with StableTTS does
     * (training/stabletts/matcha/hifigan/models.py:176) */
    const float* pb = thas(m, "dec.conv_post.bias") ? tget(m, 1, 1, -1, -1, "dec.conv_post.bias") : NULL;
    if (!pw) { free(x); return VITS_ERR_BLOB; }
    lrelu_inplace(x, n, 0.01f);
    conv1d(x, B, C, Tc, pw, pb, 1, 7, 1, 3, Tc, audio);
    for (size_t e = 0; e < (size_t)B * Tc; ++e) audio[e] = tanhf(audio[e]);
    free(x);
    return VITS_OK;
  }
  int S = hp->subbands, N = hp->istft_n_fft, hop = hp->istft_hop, cut = N / 2 + 1, P = S * (N + 2);
  const float* pw = tget(m, 3, P, C, 7, "dec.subband_conv_post.weight");
  if (!pw) { free(x); return VITS_ERR_BLOB; }
  lrelu_inplace(x, n, 0.01f); /* F.leaky_relu default slope, models.py:1038 */
  int Tp = Tc + 1;            /* ReflectionPad1d((1,0)), models.py:1039 */
  float* xp = falloc((size_t)B * C * Tp);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c) {
      const float* s = x + ((size_t)b * C + c) * Tc;
      float* d = xp + ((size_t)b * C + c) * Tp;
      d[0] = s[Tc > 1 ? 1 : 0];
      memcpy(d + 1, s, sizeof(float) * Tc);
    }
  free(x);
  float* post = falloc((size_t)B * P * Tp);
  conv1d(xp, B, C, Tp, pw, NULL, P, 7, 1, 3, Tp, post);
  free(xp);
  /* spec = exp(x[:, :, :cut]), phase = pi*sin(x[:, :, cut:]) (models.py:1043-1044);
   * OnnxSTFT.inverse (stft.py:246-262) */
  int Tm = (Tp - 1) * hop + N - N; /* after trimming N/2 each side: (Tp-1)*hop */
  float* mb = falloc((size_t)B * S * Tm);
  float* rec = falloc((size_t)2 * cut * Tp);
  float* full = falloc((size_t)(Tp - 1) * hop + N);
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      const float* ps = post + ((size_t)b * P + (size_t)s * (N + 2)) * Tp;
      for (int k = 0; k < cut; ++k)
        for (int t = 0; t < Tp; ++t) {
          float mag = expf(ps[(size_t)k * Tp + t]);
          float ph = (float)PI_D * sinf(ps[(size_t)(cut + k) * Tp + t]);
          rec[(size_t)k * Tp + t] = mag * cosf(ph);
          rec[(size_t)(cut + k) * Tp + t] = mag * sinf(ph);
        }
      size_t fl = (size_t)(Tp - 1) * hop + N;
      memset(full, 0, sizeof(float) * fl);
      for (int c = 0; c < 2 * cut; ++c)
        for (int t = 0; t < Tp; ++t) {
          float v = rec[(size_t)c * Tp + t];
          for (int j = 0; j < N; ++j) full[(size_t)t * hop + j] += v * m->istft_basis[(size_t)c * N + j];
        }
      float sc = (float)N / (float)hop;
      for (int t = 0; t < Tm; ++t) mb[((size_t)b * S + s) * Tm + t] = full[t + N / 2] * sc;
    }
  free(rec); free(full); free(post);
  if (audio_mb) memcpy(audio_mb, mb, sizeof(float) * (size_t)B * S * Tm);
  /* PQMF.synthesis (pqmf.py:105-116): zero-stuff x S (gain S), pad taps/2, FIR */
  int L = hp->pqmf_taps + 1, To = Tm * S, padl = hp->pqmf_taps / 2;
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < To; ++t) {
      float a = 0.f;
      int j0 = ((padl - t) % S + S) % S; /* first tap whose input index t+j-padl is a multiple of S */
      for (int s = 0; s < S; ++s)
        for (int j = j0; j < L; j += S) {
          int u = t + j - padl;
          if (u >= 0 && u < To) a += m->pqmf_syn[(size_t)s * L + j] * (mb[((size_t)b * S + s) * Tm + u / S] * (float)S);
        }
      audio[(size_t)b * To + t] = a;
    }
  free(mb);
  return VITS_OK;
}

/* ------------------------------------------------------------- lifecycle */

int API(create)(const void* blob, size_t bytes, int device, vits_model** out) {
  (void)device;
  if (!blob || !out || bytes < 16 + sizeof(vits_hparams)) return fail(VITS_ERR_ARG, "bad blob argument");
  const unsigned char* p = (const unsigned char*)blob;
  if (memcmp(p, "VITSW001", 8) != 0) return fail(VITS_ERR_BLOB, "bad magic");
  uint32_t hb; memcpy(&hb, p + 8, 4);
  if (hb != sizeof(vits_hparams)) return fail(VITS_ERR_BLOB, "hparams size %u != %zu", hb, sizeof(vits_hparams));
  vits_model* m = (vits_model*)calloc(1, sizeof *m);
  memcpy(&m->hp, p + 12, sizeof(vits_hparams));
  if (m->hp.abi_version != VITS_ABI_VERSION) { free(m); return fail(VITS_ERR_BLOB, "abi version mismatch"); }
  m->blob = (unsigned char*)malloc(bytes);
  memcpy(m->blob, blob, bytes);
  m->blob_bytes = bytes;
  memcpy(&m->n_entries, m->blob + 12 + hb, 4);
  m->entries = (const vits_blob_entry*)(m->blob + 16 + hb);
  if (16 + hb + (size_t)m->n_entries * sizeof(vits_blob_entry) > bytes) { free(m->blob); free(m); return fail(VITS_ERR_BLOB, "truncated table"); }
  for (uint32_t i = 0; i < m->n_entries; ++i)
    if (m->entries[i].offset + m->entries[i].nelem * 4 > bytes) { free(m->blob); free(m); return fail(VITS_ERR_BLOB, "truncated data"); }
  if (m->hp.dp_num_bins > 30 || m->hp.n_ups > VITS_MAX_UPS || m->hp.n_resk > VITS_MAX_RESK) { free(m->blob); free(m); return fail(VITS_ERR_UNSUPPORTED, "hparams out of range"); }
  build_istft_basis(m);
  build_pqmf(m);
  *out = m;
  return VITS_OK;
}

void API(destroy)(vits_model* m) {
  if (!m) return;
  free(m->blob); free(m->istft_basis); free(m->pqmf_syn); free(m);
}

const char* API(last_error)(void) { return g_err; }
int API(get_hparams)(const vits_model* m, vits_hparams* out) {
  if (!m || !out) return VITS_ERR_ARG;
  *out = m->hp;
  return VITS_OK;
}
int API(is_device_backend)(void) { return 0; }
int API(num_threads)(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
/* bench.py's cpu_baseline leg picks the thread count that serves one utterance best (B = 1 has limited parallel grain:
 * all 128 hardware threads of the GPU box's host are slower than 16) */
void API(set_num_threads)(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
/* constants for pinning against the reference buffers */
const float* API(debug_istft_basis)(const vits_model* m) { return m->istft_basis; }
const float* API(debug_pqmf_filter)(const vits_model* m) { return m->pqmf_syn; }

/* ---------------------------------------------------- Philox normal stream */

static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
/* unit normal for element (row,t) of stream `stream` — same definition as the HIP library */
static float philox_normal(uint64_t seed, uint32_t stream, uint32_t row, uint32_t t) {
  uint32_t c[4] = {t, row, stream, 0};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float u1 = ((float)c[0] + 0.5f) * (1.0f / 4294967296.0f);
  float u2 = ((float)c[1] + 0.5f) * (1.0f / 4294967296.0f);
  if (u1 < 1e-12f) u1 = 1e-12f;
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}
float API(debug_philox_normal)(uint64_t seed, uint32_t stream, uint32_t row, uint32_t t) {
  return philox_normal(seed, stream, row, t);
}

/* ------------------------------------------------------- full path (a1) */

/* SynthesizerTrn.infer (models.py:1679-1704) == onnx_export.infer_forward (:61-74) */
int API(synthesize)(vits_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T, const float* scales,
                    const int64_t* sid, const vits_synth_opts* opts, float** out_audio, int64_t* out_samples,
                    int64_t* out_lengths) {
  if (!m || !ids || !lengths || !scales || !out_audio || !out_samples || B <= 0 || T <= 0)
    return fail(VITS_ERR_ARG, "bad argument");
  const vits_hparams* hp = &m->hp;
  int H = hp->hidden_channels, I = hp->inter_channels;
  float noise_scale = scales[0], length_scale = scales[1], noise_scale_w = scales[2];
  uint64_t seed = opts ? opts->seed : 0;
  int rc;
  float* x = falloc((size_t)B * H * T); float* m_p = falloc((size_t)B * I * T); float* logs_p = falloc((size_t)B * I * T);
  float* logw = falloc((size_t)B * T);
  int32_t* dur = (int32_t*)calloc((size_t)B * T, sizeof(int32_t));
  int64_t* ylen = (int64_t*)calloc((size_t)B, sizeof(int64_t));
  float *ndp = NULL, *npr = NULL, *z_p = NULL, *z = NULL, *audio = NULL;
  rc = text_encoder_impl(m, ids, lengths, B, T, sid, opts ? opts->bert : NULL, x, m_p, logs_p);
  if (rc) goto done;
  if (!(opts && opts->forced_durations)) {
    ndp = falloc((size_t)B * 2 * T);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < 2; ++c)
        for (int t = 0; t < T; ++t)
          ndp[((size_t)b * 2 + c) * T + t] =
              (opts && opts->noise_dp) ? opts->noise_dp[((size_t)b * 2 + c) * T + t] : philox_normal(seed, 1, (uint32_t)(b * 2 + c), (uint32_t)t);
    rc = API(stage_duration)(m, x, lengths, B, T, sid, ndp, noise_scale_w, logw);
    if (rc) goto done;
  }
  rc = API(stage_regulate)(m, logw, opts ? opts->forced_durations : NULL, lengths, B, T, length_scale, NULL, NULL, NULL,
                           0.f, 0, dur, ylen, NULL);
  if (rc) goto done;
  int64_t Ty = 1;
  for (int b = 0; b < B; ++b) if (ylen[b] > Ty) Ty = ylen[b];
  if (opts && opts->max_frames > 0 && Ty > opts->max_frames) { rc = fail(VITS_ERR_ARG, "T_y %lld exceeds max_frames", (long long)Ty); goto done; }
  npr = falloc((size_t)B * I * Ty);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < I; ++c)
      for (int t = 0; t < Ty; ++t)
        npr[((size_t)b * I + c) * Ty + t] =
            (opts && opts->noise_prior) ? opts->noise_prior[((size_t)b * I + c) * opts->noise_prior_stride + t]
                                        : philox_normal(seed, 2, (uint32_t)(b * I + c), (uint32_t)t);
  z_p = falloc((size_t)B * I * Ty);
  rc = API(stage_regulate)(m, logw, dur, lengths, B, T, length_scale, m_p, logs_p, npr, noise_scale, (int32_t)Ty, dur, ylen, z_p);
  if (rc) goto done;
  z = falloc((size_t)B * I * Ty);
  rc = API(stage_flow)(m, z_p, ylen, B, (int32_t)Ty, sid, z);
  if (rc) goto done;
  mul_mask(z, B, I, (int)Ty, ylen); /* (z * y_mask) models.py:1703 */
  int64_t up = hp->hop_length;
  audio = (float*)malloc(sizeof(float) * (size_t)B * Ty * up);
  rc = API(stage_decoder)(m, z, B, (int32_t)Ty, sid, audio, NULL);
  if (rc) { free(audio); audio = NULL; goto done; }
  *out_audio = audio;
  *out_samples = Ty * up;
  if (out_lengths) for (int b = 0; b < B; ++b) out_lengths[b] = ylen[b] * up;
done:
  free(x); free(m_p); free(logs_p); free(logw); free(dur); free(ylen); free(ndp); free(npr); free(z_p); free(z);
  return rc;
}
void API(free_output)(float* p) { free(p); }

int API(op_conv1d)(int device, const float* x, const float* w, const float* bias, int32_t B, int32_t Cin, int32_t Cout,
                   int32_t T, int32_t K, int32_t dil, float slope, float* y) {
  (void)device;
  size_t n = (size_t)B * Cin * T;
  float* xa = falloc(n);
  memcpy(xa, x, sizeof(float) * n);
  if (slope != 1.0f) lrelu_inplace(xa, n, slope);
  conv1d(xa, B, Cin, T, w, bias, Cout, K, dil, (K * dil - dil) / 2, T, y);
  free(xa);
  return VITS_OK;
}

/* Algorithmic FLOPs (2*MAC of every conv/matmul on the path + attention QK/PV
 * + banded relative terms), SURVEY.md §8a "alg" convention. */
double API(algorithmic_flops)(const vits_model* m, int32_t B, int32_t Tx, int32_t Ty) {
  const vits_hparams* hp = &m->hp;
  double H = hp->hidden_channels, I = hp->inter_channels, F = hp->filter_channels, D = hp->dp_filter_channels;
  double NW = 2 * hp->window_size + 1;
  /* per token */
  double enc_layer = 2 * (4 * H * H) + 2 * (2 * H * F * hp->kernel_size) + 2 * 2 * NW * H;
  double tok = hp->n_layers * enc_layer + 2 * H * 2 * I;
  double dds = hp->dp_dds_layers * (2 * D * hp->dp_kernel_size + 2 * D * D);
  tok += 2 * H * D + 2 * D * D + dds + (hp->dp_n_flows - 1) * (2 * D + dds + 2 * D * (3 * hp->dp_num_bins - 1));
  double tok_quad = hp->n_layers * 4 * H; /* QK^T + PV per (token,key) */
  /* per frame */
  double K5 = hp->flow_kernel_size;
  double fl = 2 * (I / 2) * H + (2 * (4 * H * H) + 2 * (2 * H * H * K5) + 2 * 2 * NW * H);
  for (int i = 0; i < hp->flow_wn_layers; ++i)
    fl += 2 * H * 2 * H * K5 + 2 * H * (i < hp->flow_wn_layers - 1 ? 2 * H : H);
  fl += 2 * H * (I / 2);
  double frame = hp->flow_n_flows * fl;
  double frame_quad = hp->flow_n_flows * 4 * H;
  double C = hp->dec_initial_channel, rate = 1;
  double dec = 2 * I * C * 7;
  for (int i = 0; i < hp->n_ups; ++i) {
    dec += rate * 2 * C * (C / 2) * hp->up_kernels[i]; /* per input position: K taps -> K/u per output x u outputs */
    rate *= hp->up_rates[i];
    C /= 2;
    for (int j = 0; j < hp->n_resk; ++j) dec += rate * hp->n_resd * 2 * (2 * C * C * hp->res_kernels[j]);
  }
  if (hp->dec_type == 0) {
    double P = hp->subbands * (hp->istft_n_fft + 2);
    dec += rate * 2 * C * P * 7;
    dec += rate * hp->subbands * 2 * (hp->istft_n_fft + 2) * hp->istft_n_fft;       /* iSTFT overlap-add */
    dec += rate * hp->subbands * hp->istft_hop * 2 * (hp->pqmf_taps + 1);          /* polyphase PQMF: taps+1 MAC per sample */
  } else {
    dec += rate * 2 * C * 7;
  }
  frame += dec;
  return (double)B * ((double)Tx * (tok + tok_quad * Tx) + (double)Ty * (frame + frame_quad * Ty));
}

/* ---- monotonic alignment search: monotonic_align/core.pyx:7-42 (maximum_path_each + the prange over items).
 * The Cython routine accumulates into `value` in place; this restatement works on a copy so the caller's
 * scores are preserved (paths are identical). */
int API(mas_maximum_path)(int device, const float* values, const int32_t* t_ys, const int32_t* t_xs, int32_t B, int32_t Ty,
                          int32_t Tx, int32_t* paths) {
  (void)device;
  if (!values || !t_ys || !t_xs || !paths || B <= 0 || Ty <= 0 || Tx <= 0) return fail(VITS_ERR_ARG, "bad argument");
  const float max_neg_val = -1e9f; /* core.pyx:7 */
  for (int b = 0; b < B; ++b)
    if (t_ys[b] < 0 || t_ys[b] > Ty || t_xs[b] < 0 || t_xs[b] > Tx) return fail(VITS_ERR_ARG, "extent out of range");
  memset(paths, 0, sizeof(int32_t) * (size_t)B * Ty * Tx);
  for (int b = 0; b < B; ++b) {
    const int t_y = t_ys[b], t_x = t_xs[b];
    if (t_y == 0 || t_x == 0) continue;
    float* v = (float*)malloc(sizeof(float) * (size_t)Ty * Tx);
    if (!v) return fail(VITS_ERR_NOMEM, "host alloc failed");
    memcpy(v, values + (size_t)b * Ty * Tx, sizeof(float) * (size_t)Ty * Tx);
    int32_t* path = paths + (size_t)b * Ty * Tx;
    for (int y = 0; y < t_y; ++y) { /* core.pyx:15-27 */
      const int x0 = t_x + y - t_y > 0 ? t_x + y - t_y : 0, x1 = t_x < y + 1 ? t_x : y + 1;
      for (int x = x0; x < x1; ++x) {
        const float v_cur = x == y ? max_neg_val : v[(size_t)(y - 1) * Tx + x];
        const float v_prev = x == 0 ? (y == 0 ? 0.f : max_neg_val) : v[(size_t)(y - 1) * Tx + x - 1];
        v[(size_t)y * Tx + x] += v_prev > v_cur ? v_prev : v_cur;
      }
    }
    int index = t_x - 1; /* core.pyx:29-32 */
    for (int y = t_y - 1; y >= 0; --y) {
      path[(size_t)y * Tx + index] = 1;
      if (index != 0 && (index == y || v[(size_t)(y - 1) * Tx + index] < v[(size_t)(y - 1) * Tx + index - 1])) index -= 1;
    }
    free(v);
  }
  return VITS_OK;
}
