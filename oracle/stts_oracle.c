/*
 * stts_oracle.c — TEST INFRASTRUCTURE.  CPU restatement (plain C, fp32) of the reference's StableTTS / Matcha
 * inference arithmetic (training/stabletts/matcha), the checker for the HIP path behind include/stts_mi355.h.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * Pinned against the reference: tests/test_oracle_golden.py compares every stage with tests/golden/stts_*.npz,
 * produced by oracle/gen_golden_stts.py from the reference's own modules (MatchaTTS.synthesise, the estimator,
 * the bundled HiFi-GAN) imported in the build container.  The reference has no tests of its own for this path.
 *
 * Citations are relative to /root/reference/training/stabletts/matcha/.  Exports include/stts_mi355.h with the
 * prefix sttsref_ ; the vocoder call goes to vitsref_stage_decoder (vits_oracle.c, same shared object).
 */
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/stts_mi355.h"

#define SAPI(name) sttsref_##name
int vitsref_stage_decoder(vits_model* m, const float* z, int32_t B, int32_t T, const int64_t* sid, float* audio, float* audio_mb);
int vitsref_get_hparams(const vits_model* m, vits_hparams* out);
const char* vitsref_last_error(void);

static __thread char s_err[512];
static int sfail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(s_err, sizeof s_err, fmt, ap);
  va_end(ap);
  return code;
}

struct stts_model {
  stts_hparams hp;
  unsigned char* blob;
  uint32_t n_entries;
  const vits_blob_entry* entries;
  vits_model* vocoder;
  int missing;
};

static const float* sget(stts_model* m, size_t nelem, const char* fmt, ...) {
  char name[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(name, sizeof name, fmt, ap);
  va_end(ap);
  for (uint32_t i = 0; i < m->n_entries; ++i) {
    const vits_blob_entry* e = &m->entries[i];
    if (strncmp(e->name, name, sizeof e->name) == 0) {
      if (e->nelem != nelem) { m->missing = 1; sfail(VITS_ERR_BLOB, "tensor %s: %llu elements, expected %zu", name, (unsigned long long)e->nelem, nelem); return NULL; }
      return (const float*)(m->blob + e->offset);
    }
  }
  m->missing = 1;
  sfail(VITS_ERR_BLOB, "tensor %s missing from blob", name);
  return NULL;
}

static float* fal(size_t n) { return (float*)calloc(n ? n : 1, sizeof(float)); }
static inline float siluf(float v) { return v / (1.0f + expf(-v)); }

/* nn.Conv1d, stride 1, zero 'same' padding K/2, no mask */
static void conv1d(const float* x, int B, int Cin, int T, const float* w, const float* bias, int Cout, int K, float* y) {
  const int pad = K / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float* yr = y + ((size_t)b * Cout + co) * T;
      for (int t = 0; t < T; ++t) yr[t] = bias ? bias[co] : 0.f;
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xr = x + ((size_t)b * Cin + ci) * T;
        for (int k = 0; k < K; ++k) {
          const float wv = w[((size_t)co * Cin + ci) * K + k];
          const int s0 = k - pad;
          const int lo = s0 < 0 ? -s0 : 0, hi = T - s0 < T ? T - s0 : T;
          for (int t = lo; t < hi; ++t) yr[t] += wv * xr[t + s0];
        }
      }
    }
}

/* nn.Linear on vectors: y[b] = W x[b] + bias */
static void linear(const float* x, int B, int Cin, const float* w, const float* bias, int Cout, float* y) {
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float a = bias ? bias[co] : 0.f;
      for (int ci = 0; ci < Cin; ++ci) a += w[(size_t)co * Cin + ci] * x[(size_t)b * Cin + ci];
      y[(size_t)b * Cout + co] = a;
    }
}

static void mask_inplace(float* x, int B, int C, int T, const int64_t* len) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int t = (int)len[b] < 0 ? 0 : (int)len[b]; t < T; ++t) x[((size_t)b * C + c) * T + t] = 0.f;
}

/* nn.LayerNorm(H, elementwise_affine=False) over channels, then modulate(x, shift, scale) = x*(1+scale)+shift
 * (diffusion_transformer.py:90,111,120-122) */
static void ln_modulate(const float* x, int B, int H, int T, const float* shift, const float* scale, float* y) {
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      float mean = 0.f, var = 0.f;
      for (int c = 0; c < H; ++c) mean += x[((size_t)b * H + c) * T + t];
      mean /= (float)H;
      for (int c = 0; c < H; ++c) { const float d = x[((size_t)b * H + c) * T + t] - mean; var += d * d; }
      const float rstd = 1.0f / sqrtf(var / (float)H + 1e-5f);
      for (int c = 0; c < H; ++c)
        y[((size_t)b * H + c) * T + t] = (x[((size_t)b * H + c) * T + t] - mean) * rstd * (1.0f + scale[(size_t)b * H + c]) + shift[(size_t)b * H + c];
    }
}

/* MultiHeadAttention.forward (diffusion_transformer.py:59-80): q,k,v 1x1 convs, RoPE on the first d = dk/2 features of q
 * and k (RotaryPositionalEmbeddings :124-198: pairs (j, j+d/2), theta_j = 10000^(-2j/d), position = frame index),
 * F.scaled_dot_product_attention with the additive mask of :107-108, conv_o. */
static void dit_attention(stts_model* m, const char* p, const float* x, int B, int H, int T, int heads, const int64_t* len, float* y) {
  const int dk = H / heads, d = (int)(dk * 0.5), d2 = d / 2;
  float* q = fal((size_t)B * H * T); float* k = fal((size_t)B * H * T); float* v = fal((size_t)B * H * T); float* o = fal((size_t)B * H * T);
  conv1d(x, B, H, T, sget(m, (size_t)H * H, "%s.conv_q.weight", p), sget(m, H, "%s.conv_q.bias", p), H, 1, q);
  conv1d(x, B, H, T, sget(m, (size_t)H * H, "%s.conv_k.weight", p), sget(m, H, "%s.conv_k.bias", p), H, 1, k);
  conv1d(x, B, H, T, sget(m, (size_t)H * H, "%s.conv_v.weight", p), sget(m, H, "%s.conv_v.bias", p), H, 1, v);
  for (int which = 0; which < 2; ++which) {
    float* a = which ? k : q;
    for (int b = 0; b < B; ++b)
      for (int h = 0; h < heads; ++h)
        for (int j = 0; j < d2; ++j) {
          const float theta = 1.0f / powf(10000.0f, (float)(2 * j) / (float)d);
          float* r0 = a + ((size_t)b * H + h * dk + j) * T;
          float* r1 = a + ((size_t)b * H + h * dk + j + d2) * T;
          for (int t = 0; t < T; ++t) {
            const float ang = (float)t * theta, cs = cosf(ang), sn = sinf(ang);
            const float x0 = r0[t], x1 = r1[t];
            r0[t] = x0 * cs - x1 * sn;  /* x_rope*cos + neg_half*sin, neg_half = [-x[d/2:], x[:d/2]] */
            r1[t] = x1 * cs + x0 * sn;
          }
        }
  }
  const float scale = 1.0f / sqrtf((float)dk);
#pragma omp parallel for collapse(2) schedule(static)
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < heads; ++h) {
      const int L = (int)len[b];
      float* sc = (float*)malloc(sizeof(float) * (size_t)T);
      for (int i = 0; i < T; ++i) {
        float* orow0 = o + ((size_t)b * H + h * dk) * T + i;
        /* masked query rows (i >= L): every score is -finfo.max -> uniform weights over all T keys (:107-108) */
        float mx = -3.0e38f;
        for (int j = 0; j < T; ++j) {
          float s = 0.f;
          if (i < L && j < L) {
            for (int c = 0; c < dk; ++c) s += q[((size_t)b * H + h * dk + c) * T + i] * k[((size_t)b * H + h * dk + c) * T + j];
            s *= scale;
          } else {
            s = -3.4028234663852886e38f;
          }
          sc[j] = s;
          if (s > mx) mx = s;
        }
        float den = 0.f;
        for (int j = 0; j < T; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
        for (int c = 0; c < dk; ++c) {
          float a = 0.f;
          for (int j = 0; j < T; ++j) a += sc[j] * v[((size_t)b * H + h * dk + c) * T + j];
          orow0[(size_t)c * T] = a / den;
        }
      }
      free(sc);
    }
  conv1d(o, B, H, T, sget(m, (size_t)H * H, "%s.conv_o.weight", p), sget(m, H, "%s.conv_o.bias", p), H, 1, y);
  free(q); free(k); free(v); free(o);
}

/* DiTConVBlock.forward (diffusion_transformer.py:99-118); x [B,H,T] in place, c [B,G] */
static void dit_block(stts_model* m, const char* p, float* x, const float* c, int B, int H, int F, int heads, int K, int T, const int64_t* len) {
  const int G = m->hp.spk_emb_dim;
  char q[200];
  float* h0 = fal((size_t)B * H); float* mod = fal((size_t)B * 6 * H);
  linear(c, B, G, sget(m, (size_t)H * G, "%s.adaLN_modulation.0.weight", p), sget(m, H, "%s.adaLN_modulation.0.bias", p), H, h0);
  for (size_t i = 0; i < (size_t)B * H; ++i) h0[i] = siluf(h0[i]);
  linear(h0, B, H, sget(m, (size_t)6 * H * H, "%s.adaLN_modulation.2.weight", p), sget(m, (size_t)6 * H, "%s.adaLN_modulation.2.bias", p), 6 * H, mod);
  if (m->missing) { free(h0); free(mod); return; }
  /* chunk(6, dim=1): shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp ; repack per chunk as [B,H] */
  float* ch[6];
  for (int j = 0; j < 6; ++j) { ch[j] = fal((size_t)B * H); for (int b = 0; b < B; ++b) memcpy(ch[j] + (size_t)b * H, mod + ((size_t)b * 6 + j) * H, sizeof(float) * H); }
  mask_inplace(x, B, H, T, len); /* x = x * x_mask */
  float* n = fal((size_t)B * H * T); float* a = fal((size_t)B * H * T);
  ln_modulate(x, B, H, T, ch[0], ch[1], n);
  snprintf(q, sizeof q, "%s.attn", p);
  dit_attention(m, q, n, B, H, T, heads, len, a);
  mask_inplace(a, B, H, T, len);
  for (int b = 0; b < B; ++b) for (int cc = 0; cc < H; ++cc) for (int t = 0; t < T; ++t) x[((size_t)b * H + cc) * T + t] += ch[2][(size_t)b * H + cc] * a[((size_t)b * H + cc) * T + t];
  /* FFN (diffusion_transformer.py:25-31): conv_1(x*mask) -> SiLU -> conv_2(.*mask) -> *mask */
  ln_modulate(x, B, H, T, ch[3], ch[4], n);
  mask_inplace(n, B, H, T, len);
  float* f = fal((size_t)B * F * T);
  conv1d(n, B, H, T, sget(m, (size_t)F * H * K, "%s.mlp.conv_1.weight", p), sget(m, F, "%s.mlp.conv_1.bias", p), F, K, f);
  for (size_t i = 0; i < (size_t)B * F * T; ++i) f[i] = siluf(f[i]);
  mask_inplace(f, B, F, T, len);
  conv1d(f, B, F, T, sget(m, (size_t)H * F * K, "%s.mlp.conv_2.weight", p), sget(m, H, "%s.mlp.conv_2.bias", p), H, K, a);
  mask_inplace(a, B, H, T, len);
  for (int b = 0; b < B; ++b) for (int cc = 0; cc < H; ++cc) for (int t = 0; t < T; ++t) x[((size_t)b * H + cc) * T + t] += ch[5][(size_t)b * H + cc] * a[((size_t)b * H + cc) * T + t];
  for (int j = 0; j < 6; ++j) free(ch[j]);
  free(h0); free(mod); free(n); free(a); free(f);
}

/* ---- TextEncoder.forward (text_encoder.py:111-139) */
int SAPI(stage_encoder)(stts_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t T, const int64_t* sid,
                        const float* bert, float* x, float* mu_dp) {
  if (!m || !ids || !lengths || !x || !mu_dp || B <= 0 || T <= 0) return sfail(VITS_ERR_ARG, "bad argument");
  const stts_hparams* hp = &m->hp;
  const int H = hp->enc_hidden, E = hp->emb_dim, Pd = hp->punc_dim, BP = hp->bert_proj_dim, G = hp->spk_emb_dim;
  if (E + 4 * Pd + BP != H) return sfail(VITS_ERR_UNSUPPORTED, "stream widths do not add up to enc_hidden");
  const float* emb = sget(m, (size_t)hp->n_vocab * E, "encoder.emb.weight");
  const float* pemb = sget(m, (size_t)hp->n_vocab * Pd, "encoder.punc_emb.weight");
  const float* bw = sget(m, (size_t)BP * hp->bert_dim, "encoder.bert_proj.1.weight");
  const float* bb = sget(m, BP, "encoder.bert_proj.1.bias");
  const float* dse = hp->n_spks > 1 ? sget(m, (size_t)hp->n_spks * G, "dur_spk_emb.weight") : NULL;
  if (m->missing) return VITS_ERR_BLOB;
  for (int b = 0; b < B; ++b) {
    if (lengths[b] < 0 || lengths[b] > T) return sfail(VITS_ERR_ARG, "length out of range");
    if (hp->n_spks > 1 && (!sid || sid[b] < 0 || sid[b] >= hp->n_spks)) return sfail(VITS_ERR_ARG, "speaker id out of range");
    for (int s = 0; s < 5; ++s)
      for (int t = 0; t < T; ++t) {
        const int64_t id = ids[((size_t)b * 5 + s) * T + t];
        if (id < 0 || id >= hp->n_vocab) return sfail(VITS_ERR_ARG, "token id out of range");
      }
  }
  const float es = sqrtf((float)E), ps = sqrtf((float)Pd);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t) {
      const int64_t i0 = ids[((size_t)b * 5 + 0) * T + t];
      for (int c = 0; c < E; ++c) x[((size_t)b * H + c) * T + t] = emb[(size_t)i0 * E + c] * es;
      for (int s = 1; s < 5; ++s) {
        const int64_t is = ids[((size_t)b * 5 + s) * T + t];
        for (int c = 0; c < Pd; ++c) x[((size_t)b * H + E + (s - 1) * Pd + c) * T + t] = pemb[(size_t)is * Pd + c] * ps;
      }
      for (int c = 0; c < BP; ++c) { /* bert_proj: Dropout (eval) + Linear(768, 32) on bert[:, :, t] */
        float a = bb[c];
        if (bert)
          for (int j = 0; j < hp->bert_dim; ++j) a += bw[(size_t)c * hp->bert_dim + j] * bert[((size_t)b * hp->bert_dim + j) * T + t];
        x[((size_t)b * H + E + 4 * Pd + c) * T + t] = a;
      }
    }
  /* x_dp, mu_dp = dp_encoder(x, dur_spks, x_mask) (text_encoder.py:137; Encoder.forward :40-47) */
  float* h = fal((size_t)B * H * T);
  memcpy(h, x, sizeof(float) * (size_t)B * H * T);
  float* c = fal((size_t)B * G);
  if (dse) for (int b = 0; b < B; ++b) memcpy(c + (size_t)b * G, dse + (size_t)sid[b] * G, sizeof(float) * G);
  char p[200];
  for (int i = 0; i < hp->enc_layers && !m->missing; ++i) {
    snprintf(p, sizeof p, "encoder.dp_encoder.encoder.%d", i);
    dit_block(m, p, h, c, B, H, hp->enc_filter, hp->enc_heads, hp->enc_kernel, T, lengths);
  }
  if (!m->missing) {
    conv1d(h, B, H, T, sget(m, (size_t)hp->dp_out * H, "encoder.dp_encoder.proj.weight"), sget(m, hp->dp_out, "encoder.dp_encoder.proj.bias"), hp->dp_out, 1, mu_dp);
    mask_inplace(mu_dp, B, hp->dp_out, T, lengths);
  }
  free(h); free(c);
  return m->missing ? VITS_ERR_BLOB : VITS_OK;
}

/* ---- durations (matcha_tts.py:144-158; DeterministicDurationPredictor returns its input * mask) */
int SAPI(stage_durations)(stts_model* m, const float* mu_dp, int32_t B, int32_t T, float length_scale, const float* pde,
                          int32_t* durations, int64_t* y_lengths) {
  if (!m || !mu_dp || !durations || !y_lengths || B <= 0 || T <= 0) return sfail(VITS_ERR_ARG, "bad argument");
  const int K = m->hp.dp_out;
  for (int b = 0; b < B; ++b) {
    int64_t tot = 0;
    for (int t = 0; t < T; ++t) {
      float lw = 0.f;
      for (int k = 0; k < K; ++k) lw += 1.0f / (1.0f + expf(-mu_dp[((size_t)b * K + k) * T + t])); /* sigmoid(logw).sum(axis=1) */
      if (pde && pde[(size_t)b * T + t] != 0.f) lw = pde[(size_t)b * T + t];                          /* torch.where(pde == 0, logw, pde) */
      float w = rintf(lw * length_scale);                                                              /* torch.round: half to even */
      if (w < 1.f) w = 1.f;
      durations[(size_t)b * T + t] = (int32_t)w;
      tot += (int64_t)w;
    }
    y_lengths[b] = tot;
  }
  return VITS_OK;
}

/* ---- one estimator call: Decoder.forward (components/decoder.py:105-138) */
static int estimator(stts_model* m, const float* x, const float* mu, const int64_t* ylen, int B, int T, float tval, const float* c, float* out) {
  const stts_hparams* hp = &m->hp;
  const int H = hp->dec_hidden, F = hp->dec_filter, NF = hp->n_feats, CC = hp->enc_hidden, K = hp->dec_kernel, NL = hp->dec_layers;
  const char* e = "decoder.estimator";
  /* SinusoidalPosEmb(H)(t, scale=1000) -> TimestepEmbedding (decoder.py:35-62) */
  const int half = H / 2;
  float* te = fal(H); float* t1 = fal(F); float* temb = fal(H);
  const float lg = logf(10000.0f) / (float)(half - 1);
  for (int j = 0; j < half; ++j) {
    const float a = 1000.0f * tval * expf((float)j * -lg);
    te[j] = sinf(a); te[half + j] = cosf(a);
  }
  linear(te, 1, H, sget(m, (size_t)F * H, "%s.time_mlp.layer.0.weight", e), sget(m, F, "%s.time_mlp.layer.0.bias", e), F, t1);
  for (int j = 0; j < F; ++j) t1[j] = siluf(t1[j]);
  linear(t1, 1, F, sget(m, (size_t)H * F, "%s.time_mlp.layer.2.weight", e), sget(m, H, "%s.time_mlp.layer.2.bias", e), H, temb);
  if (m->missing) { free(te); free(t1); free(temb); return VITS_ERR_BLOB; }
  /* mu = cond_proj(mu): conv k -> SiLU -> conv k -> SiLU -> conv k, no masks (decoder.py:82-88,121) */
  float* a1 = fal((size_t)B * F * T); float* a2 = fal((size_t)B * F * T); float* cat = fal((size_t)B * (NF + H) * T);
  conv1d(mu, B, CC, T, sget(m, (size_t)F * CC * K, "%s.cond_proj.0.weight", e), sget(m, F, "%s.cond_proj.0.bias", e), F, K, a1);
  for (size_t i = 0; i < (size_t)B * F * T; ++i) a1[i] = siluf(a1[i]);
  conv1d(a1, B, F, T, sget(m, (size_t)F * F * K, "%s.cond_proj.2.weight", e), sget(m, F, "%s.cond_proj.2.bias", e), F, K, a2);
  for (size_t i = 0; i < (size_t)B * F * T; ++i) a2[i] = siluf(a2[i]);
  float* muc = fal((size_t)B * H * T);
  conv1d(a2, B, F, T, sget(m, (size_t)H * F * K, "%s.cond_proj.4.weight", e), sget(m, H, "%s.cond_proj.4.bias", e), H, K, muc);
  for (int b = 0; b < B; ++b) { /* x = cat((x, mu), dim=1) ; in_proj */
    memcpy(cat + (size_t)b * (NF + H) * T, x + (size_t)b * NF * T, sizeof(float) * (size_t)NF * T);
    memcpy(cat + ((size_t)b * (NF + H) + NF) * T, muc + (size_t)b * H * T, sizeof(float) * (size_t)H * T);
  }
  float* h = fal((size_t)B * H * T);
  conv1d(cat, B, NF + H, T, sget(m, (size_t)H * (NF + H), "%s.in_proj.weight", e), sget(m, H, "%s.in_proj.bias", e), H, 1, h);
  free(a1); free(a2); free(muc); free(cat);
  float** stack = (float**)calloc(NL, sizeof(float*));
  int sp = 0;
  float* film = fal(2 * H); float* cat2 = fal((size_t)B * 2 * H * T);
  char p[200];
  for (int idx = 0; idx < NL && !m->missing; ++idx) {
    if (idx < NL / 2) { /* lsc_outputs.append(x) */
      stack[sp] = fal((size_t)B * H * T);
      memcpy(stack[sp++], h, sizeof(float) * (size_t)B * H * T);
    } else { /* x = lsc_layers[idx - n](cat((x, lsc_outputs.pop()), dim=1)) */
      float* s = stack[--sp];
      for (int b = 0; b < B; ++b) {
        memcpy(cat2 + (size_t)b * 2 * H * T, h + (size_t)b * H * T, sizeof(float) * (size_t)H * T);
        memcpy(cat2 + ((size_t)b * 2 * H + H) * T, s + (size_t)b * H * T, sizeof(float) * (size_t)H * T);
      }
      free(s);
      conv1d(cat2, B, 2 * H, T, sget(m, (size_t)H * 2 * H * K, "%s.lsc_layers.%d.weight", e, idx - NL / 2), sget(m, H, "%s.lsc_layers.%d.bias", e, idx - NL / 2), H, K, h);
    }
    /* DitWrapper.forward (decoder.py:15-18): x = FiLM(x, t) * mask ; block(x, c, mask) */
    linear(temb, 1, H, sget(m, (size_t)2 * H * H, "%s.blocks.%d.time_fusion.film.weight", e, idx), sget(m, (size_t)2 * H, "%s.blocks.%d.time_fusion.film.bias", e, idx), 2 * H, film);
    if (m->missing) break;
    for (int b = 0; b < B; ++b)
      for (int cc = 0; cc < H; ++cc)
        for (int t = 0; t < T; ++t) {
          float* v = &h[((size_t)b * H + cc) * T + t];
          *v = t < ylen[b] ? film[cc] * *v + film[H + cc] : 0.f;
        }
    snprintf(p, sizeof p, "%s.blocks.%d.block", e, idx);
    dit_block(m, p, h, c, B, H, F, hp->dec_heads, K, T, ylen);
  }
  while (sp > 0) free(stack[--sp]);
  free(stack); free(film); free(cat2);
  if (!m->missing) {
    mask_inplace(h, B, H, T, ylen); /* final_proj(x * mask) * mask */
    conv1d(h, B, H, T, sget(m, (size_t)NF * H, "%s.final_proj.weight", e), sget(m, NF, "%s.final_proj.bias", e), NF, 1, out);
    mask_inplace(out, B, NF, T, ylen);
  }
  free(h); free(te); free(t1); free(temb);
  return m->missing ? VITS_ERR_BLOB : VITS_OK;
}

int SAPI(stage_estimator)(stts_model* m, const float* x, const float* mu, const int64_t* y_lengths, int32_t B, int32_t T, float t,
                          const float* c, float* out) {
  if (!m || !x || !mu || !y_lengths || !c || !out || B <= 0 || T <= 0) return sfail(VITS_ERR_ARG, "bad argument");
  return estimator(m, x, mu, y_lengths, B, T, t, c, out);
}

/* ---- BASECFM.forward + solve_euler + func_dphi_dt (flow_matching.py:36-108,177-189), B = 1 */
static int cfm(stts_model* m, const float* mu_y, int64_t ylen, int T, const float* spk, const float* noise, int64_t nstride,
               float temperature, int n_steps, float* x) {
  const stts_hparams* hp = &m->hp;
  const int NF = hp->n_feats, CC = hp->enc_hidden, G = hp->spk_emb_dim;
  const float* fs = sget(m, G, "fake_speaker");
  const float* fc = sget(m, CC, "fake_content");
  if (m->missing) return VITS_ERR_BLOB;
  for (int c = 0; c < NF; ++c) for (int t = 0; t < T; ++t) x[(size_t)c * T + t] = noise[(size_t)c * nstride + t] * temperature;
  float* tspan = fal(n_steps + 1);
  for (int i = 0; i <= n_steps; ++i) { /* torch.linspace(0,1,n+1) then 1 - cos(t * 0.5 * pi) */
    const float lin = n_steps > 0 ? (i < (n_steps + 1) / 2 ? 0.0f + (1.0f / (float)n_steps) * (float)i : 1.0f - (1.0f / (float)n_steps) * (float)(n_steps - i)) : 0.f;
    tspan[i] = 1.0f - cosf(lin * 0.5f * 3.14159265358979323846f);
  }
  float* fmu = fal((size_t)CC * T); /* fake_content.repeat(1, 1, T) */
  for (int c = 0; c < CC; ++c) for (int t = 0; t < T; ++t) fmu[(size_t)c * T + t] = fc[c];
  float* d1 = fal((size_t)NF * T); float* d2 = fal((size_t)NF * T);
  float t = tspan[0], dt = n_steps > 0 ? tspan[1] - tspan[0] : 0.f;
  const float g = hp->guidance_scale;
  int rc = VITS_OK;
  for (int step = 1; step <= n_steps && rc == VITS_OK; ++step) {
    rc = estimator(m, x, mu_y, &ylen, 1, T, t, spk, d1);
    if (rc == VITS_OK && g > 0.f) {
      rc = estimator(m, x, fmu, &ylen, 1, T, t, fs, d2);
      for (size_t i = 0; i < (size_t)NF * T; ++i) d1[i] = d1[i] + g * (d1[i] - d2[i]);
    }
    for (size_t i = 0; i < (size_t)NF * T; ++i) x[i] = x[i] + dt * d1[i];
    t = t + dt;
    if (step < n_steps) dt = tspan[step + 1] - t;
  }
  free(tspan); free(fmu); free(d1); free(d2);
  return rc;
}

int SAPI(stage_cfm)(stts_model* m, const float* mu_y, int64_t y_length, int32_t T, int64_t sid, const float* noise, float temperature,
                    int32_t n_timesteps, float* out) {
  if (!m || !mu_y || !noise || !out || T <= 0 || y_length < 0 || y_length > T) return sfail(VITS_ERR_ARG, "bad argument");
  const int G = m->hp.spk_emb_dim;
  float* spk = fal(G);
  if (m->hp.n_spks > 1) {
    if (sid < 0 || sid >= m->hp.n_spks) { free(spk); return sfail(VITS_ERR_ARG, "speaker id out of range"); }
    const float* se = sget(m, (size_t)m->hp.n_spks * G, "spk_emb.weight");
    if (!se) { free(spk); return VITS_ERR_BLOB; }
    memcpy(spk, se + (size_t)sid * G, sizeof(float) * G);
  }
  int rc = cfm(m, mu_y, y_length, T, spk, noise, T, temperature, n_timesteps > 0 ? n_timesteps : m->hp.n_timesteps, out);
  free(spk);
  return rc;
}

/* ---- noise for the seeded path: the same Philox4x32-10 + Box-Muller stream as vits (stream 3, row = mel channel) */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
static float philox_normal(uint64_t seed, uint32_t stream, uint32_t row, uint32_t t) {
  uint32_t c[4] = {t, row, stream, 0};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

/* ---- the hot path: MatchaTTS.synthesise (matcha_tts.py:93-211) + vocoder.decode(mel).clamp(-1,1) (onnx/export.py:28-31) */
int SAPI(synthesize)(stts_model* m, const int64_t* ids, int32_t Tx, const float* scales, int64_t sid, const float* bert, const float* pde,
                     const stts_synth_opts* opts, float** out_audio, int64_t* out_samples, float** out_mel, int64_t* out_frames) {
  if (!m || !ids || !scales || Tx <= 0 || (out_audio && !out_samples) || (out_mel && !out_frames)) return sfail(VITS_ERR_ARG, "bad argument");
  if (out_audio && !m->vocoder) return sfail(VITS_ERR_ARG, "no vocoder attached");
  const stts_hparams* hp = &m->hp;
  const int CC = hp->enc_hidden, NF = hp->n_feats, G = hp->spk_emb_dim;
  const float temperature = scales[0], length_scale = scales[1];
  const int64_t len = Tx;
  float* x = fal((size_t)CC * Tx); float* mu_dp = fal((size_t)hp->dp_out * Tx);
  int32_t* dur = (int32_t*)calloc(Tx, sizeof(int32_t));
  int64_t ylen = 0;
  int rc = SAPI(stage_encoder)(m, ids, &len, 1, Tx, &sid, bert, x, mu_dp);
  if (rc == VITS_OK) rc = SAPI(stage_durations)(m, mu_dp, 1, Tx, length_scale, pde, dur, &ylen);
  float *mu_y = NULL, *z = NULL, *noise = NULL, *spk = NULL;
  if (rc == VITS_OK) {
    const int T = (int)((ylen + 3) / 4 * 4); /* fix_len_compatibility (utils/model.py:14-20) */
    mu_y = fal((size_t)CC * T); z = fal((size_t)NF * T); spk = fal(G);
    float* pau = fal(T);
    int t = 0;
    for (int j = 0; j < Tx; ++j) /* generate_path + matmul == gather (matcha_tts.py:163-174) */
      for (int r = 0; r < dur[j]; ++r, ++t) {
        for (int c = 0; c < CC; ++c) mu_y[(size_t)c * T + t] = x[(size_t)c * Tx + j];
        pau[t] = pde ? pde[j] : 0.f;
      }
    int64_t nstride = T;
    const float* nz = NULL;
    if (opts && opts->noise) {
      if (opts->noise_stride < T) rc = sfail(VITS_ERR_ARG, "noise stride %lld < %d", (long long)opts->noise_stride, T);
      nz = opts->noise; nstride = opts->noise_stride;
    } else {
      noise = fal((size_t)NF * T);
      for (int c = 0; c < NF; ++c) for (int tt = 0; tt < T; ++tt) noise[(size_t)c * T + tt] = philox_normal(opts ? opts->seed : 0, 3, (uint32_t)c, (uint32_t)tt);
      nz = noise;
    }
    if (rc == VITS_OK && hp->n_spks > 1) {
      const float* se = sget(m, (size_t)hp->n_spks * G, "spk_emb.weight");
      if (se) memcpy(spk, se + (size_t)sid * G, sizeof(float) * G); else rc = VITS_ERR_BLOB;
    }
    if (rc == VITS_OK) rc = cfm(m, mu_y, ylen, T, spk, nz, nstride, temperature, opts && opts->n_timesteps > 0 ? opts->n_timesteps : hp->n_timesteps, z);
    if (rc == VITS_OK) {
      /* decoder_outputs[:, :, :y_len]; frames of forced pauses take frame 0 of the output (matcha_tts.py:180-192);
       * denormalize (utils/model.py:73-90) */
      float* mel = (float*)malloc(sizeof(float) * (size_t)NF * ylen);
      for (int c = 0; c < NF; ++c) {
        const float sil = z[(size_t)c * T];
        for (int64_t tt = 0; tt < ylen; ++tt) mel[(size_t)c * ylen + tt] = (pau[tt] > 0.f ? sil : z[(size_t)c * T + tt]) * hp->mel_std + hp->mel_mean;
      }
      if (out_audio) {
        vits_hparams vh;
        vitsref_get_hparams(m->vocoder, &vh);
        const int64_t S = ylen * vh.hop_length;
        float* wav = (float*)malloc(sizeof(float) * (size_t)S);
        rc = vitsref_stage_decoder(m->vocoder, mel, 1, (int32_t)ylen, NULL, wav, NULL);
        if (rc != VITS_OK) { sfail(rc, "vocoder: %s", vitsref_last_error()); free(wav); }
        else {
          for (int64_t i = 0; i < S; ++i) wav[i] = wav[i] < -1.f ? -1.f : (wav[i] > 1.f ? 1.f : wav[i]);
          *out_audio = wav; *out_samples = S;
        }
      }
      if (rc == VITS_OK && out_mel) { *out_mel = mel; *out_frames = ylen; } else free(mel);
    }
    free(pau);
  }
  free(x); free(mu_dp); free(dur); free(mu_y); free(z); free(noise); free(spk);
  return rc;
}

/* ---- batch of independent utterances: by definition item b is the single-utterance path on its own inputs with
 * seed + b (include/stts_mi355.h); the oracle simply loops */
int SAPI(synthesize_batch)(stts_model* m, const int64_t* ids, const int64_t* lengths, int32_t B, int32_t Tx, const float* scales,
                           const int64_t* sid, const float* bert, const float* pde, const stts_synth_opts* opts, float** out_audio,
                           int64_t* out_samples, int64_t* out_lengths) {
  if (!m || !ids || !lengths || !scales || !out_audio || !out_samples || !out_lengths || B <= 0 || Tx <= 0) return sfail(VITS_ERR_ARG, "bad argument");
  if (opts && opts->noise) return sfail(VITS_ERR_ARG, "injected noise is a single-utterance option");
  float** wav = (float**)calloc(B, sizeof(float*));
  int64_t smax = 0;
  int rc = VITS_OK;
  for (int b = 0; b < B && rc == VITS_OK; ++b) {
    const int L = (int)lengths[b];
    if (L <= 0 || L > Tx) { rc = sfail(VITS_ERR_ARG, "length out of range"); break; }
    int64_t* idb = (int64_t*)malloc(sizeof(int64_t) * 5 * L);
    float* bb = bert ? (float*)malloc(sizeof(float) * (size_t)m->hp.bert_dim * L) : NULL;
    for (int s5 = 0; s5 < 5; ++s5) memcpy(idb + (size_t)s5 * L, ids + ((size_t)b * 5 + s5) * Tx, sizeof(int64_t) * L);
    if (bb) for (int c = 0; c < m->hp.bert_dim; ++c) memcpy(bb + (size_t)c * L, bert + ((size_t)b * m->hp.bert_dim + c) * Tx, sizeof(float) * L);
    stts_synth_opts o = {0};
    if (opts) o = *opts;
    o.seed = (opts && (opts->flags & STTS_FLAG_ITEM_SEEDS) && opts->item_seeds) ? opts->item_seeds[b] : (opts ? opts->seed : 0) + (uint64_t)b;
    o.item_seeds = NULL; o.flags &= ~STTS_FLAG_ITEM_SEEDS;
    int64_t ns = 0;
    rc = SAPI(synthesize)(m, idb, L, scales, sid ? sid[b] : 0, bb, pde ? pde + (size_t)b * Tx : NULL, &o, &wav[b], &ns, NULL, NULL);
    out_lengths[b] = ns;
    if (ns > smax) smax = ns;
    free(idb); free(bb);
  }
  if (rc == VITS_OK) {
    float* all = (float*)calloc((size_t)B * (smax ? smax : 1), sizeof(float));
    for (int b = 0; b < B; ++b) memcpy(all + (size_t)b * smax, wav[b], sizeof(float) * (size_t)out_lengths[b]);
    *out_audio = all; *out_samples = smax;
  }
  for (int b = 0; b < B; ++b) free(wav[b]);
  free(wav);
  return rc;
}

/* ---- lifecycle */
int SAPI(create)(const void* blob, size_t bytes, vits_model* vocoder, int device, stts_model** out) {
  (void)device;
  if (!blob || !out || bytes < 16 + sizeof(stts_hparams)) return sfail(VITS_ERR_ARG, "bad blob argument");
  const unsigned char* p = (const unsigned char*)blob;
  if (memcmp(p, "STTSW001", 8) != 0) return sfail(VITS_ERR_BLOB, "bad magic");
  uint32_t hb; memcpy(&hb, p + 8, 4);
  if (hb != sizeof(stts_hparams)) return sfail(VITS_ERR_BLOB, "hparams size %u != %zu", hb, sizeof(stts_hparams));
  stts_model* m = (stts_model*)calloc(1, sizeof *m);
  memcpy(&m->hp, p + 12, sizeof(stts_hparams));
  if (m->hp.abi_version != STTS_ABI_VERSION) { free(m); return sfail(VITS_ERR_BLOB, "abi version mismatch"); }
  m->blob = (unsigned char*)malloc(bytes);
  memcpy(m->blob, blob, bytes);
  memcpy(&m->n_entries, m->blob + 12 + hb, 4);
  m->entries = (const vits_blob_entry*)(m->blob + 16 + hb);
  if (16 + hb + (size_t)m->n_entries * sizeof(vits_blob_entry) > bytes) { free(m->blob); free(m); return sfail(VITS_ERR_BLOB, "truncated table"); }
  for (uint32_t i = 0; i < m->n_entries; ++i)
    if (m->entries[i].offset + m->entries[i].nelem * 4 > bytes) { free(m->blob); free(m); return sfail(VITS_ERR_BLOB, "truncated data"); }
  m->vocoder = vocoder;
  *out = m;
  return VITS_OK;
}
void SAPI(destroy)(stts_model* m) { if (m) { free(m->blob); free(m); } }
const char* SAPI(last_error)(void) { return s_err; }
int SAPI(get_hparams)(const stts_model* m, stts_hparams* out) {
  if (!m || !out) return sfail(VITS_ERR_ARG, "null argument");
  *out = m->hp;
  return VITS_OK;
}

/* ================================================================== word-embedding BERT encoder
 * transformers.BertModel as exported by onnx/bert-export.py:5-13 (output hidden_states[-3]).  Restated from the
 * published architecture (Devlin et al. 2019; transformers modeling_bert: BertEmbeddings, BertSelfAttention,
 * BertSelfOutput, BertIntermediate, BertOutput) and pinned to transformers' own output in tests/golden/bert_*.npz. */
struct bert_model {
  bert_hparams hp;
  unsigned char* blob;
  uint32_t n_entries;
  const vits_blob_entry* entries;
  int missing;
};
static const float* bget(bert_model* m, size_t nelem, const char* fmt, ...) {
  char name[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(name, sizeof name, fmt, ap);
  va_end(ap);
  for (uint32_t i = 0; i < m->n_entries; ++i) {
    const vits_blob_entry* e = &m->entries[i];
    if (strncmp(e->name, name, sizeof e->name) == 0) {
      if (e->nelem != nelem) { m->missing = 1; sfail(VITS_ERR_BLOB, "tensor %s: %llu elements, expected %zu", name, (unsigned long long)e->nelem, nelem); return NULL; }
      return (const float*)(m->blob + e->offset);
    }
  }
  m->missing = 1;
  sfail(VITS_ERR_BLOB, "tensor %s missing from blob", name);
  return NULL;
}
/* y[t] = W x[t] + b over rows of a [T, C] matrix */
static void dense_rows(const float* x, int T, int Cin, const float* w, const float* b, int Cout, float* y) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int t = 0; t < T; ++t)
    for (int co = 0; co < Cout; ++co) {
      float a = b[co];
      for (int ci = 0; ci < Cin; ++ci) a += w[(size_t)co * Cin + ci] * x[(size_t)t * Cin + ci];
      y[(size_t)t * Cout + co] = a;
    }
}
static void ln_rows(float* x, int T, int C, const float* g, const float* b, float eps) {
  for (int t = 0; t < T; ++t) {
    float mean = 0.f, var = 0.f;
    for (int c = 0; c < C; ++c) mean += x[(size_t)t * C + c];
    mean /= (float)C;
    for (int c = 0; c < C; ++c) { const float d = x[(size_t)t * C + c] - mean; var += d * d; }
    const float rstd = 1.0f / sqrtf(var / (float)C + eps);
    for (int c = 0; c < C; ++c) x[(size_t)t * C + c] = (x[(size_t)t * C + c] - mean) * rstd * g[c] + b[c];
  }
}
int SAPI(bert_encode)(bert_model* m, const int64_t* ids, const int64_t* types, int32_t T, float* out) {
  if (!m || !ids || !out || T <= 0) return sfail(VITS_ERR_ARG, "bad argument");
  const bert_hparams* hp = &m->hp;
  const int H = hp->hidden, F = hp->intermediate, nh = hp->n_heads, dk = H / nh;
  if (T > hp->max_position) return sfail(VITS_ERR_ARG, "%d tokens exceed max_position %d", T, hp->max_position);
  const float* we = bget(m, (size_t)hp->vocab_size * H, "embeddings.word_embeddings.weight");
  const float* pe = bget(m, (size_t)hp->max_position * H, "embeddings.position_embeddings.weight");
  const float* te = bget(m, (size_t)hp->type_vocab * H, "embeddings.token_type_embeddings.weight");
  const float* eg = bget(m, H, "embeddings.LayerNorm.weight");
  const float* eb = bget(m, H, "embeddings.LayerNorm.bias");
  if (m->missing) return VITS_ERR_BLOB;
  float* x = fal((size_t)T * H);
  for (int t = 0; t < T; ++t) {
    const int64_t id = ids[t], ty = types ? types[t] : 0;
    if (id < 0 || id >= hp->vocab_size || ty < 0 || ty >= hp->type_vocab) { free(x); return sfail(VITS_ERR_ARG, "token id out of range"); }
    for (int c = 0; c < H; ++c) x[(size_t)t * H + c] = we[(size_t)id * H + c] + te[(size_t)ty * H + c] + pe[(size_t)t * H + c];
  }
  ln_rows(x, T, H, eg, eb, hp->ln_eps);
  float* q = fal((size_t)T * H); float* k = fal((size_t)T * H); float* v = fal((size_t)T * H); float* c = fal((size_t)T * H);
  float* f = fal((size_t)T * F); float* y = fal((size_t)T * H);
  const float scale = 1.0f / sqrtf((float)dk);
  for (int l = 0; l < hp->out_layers && !m->missing; ++l) {
    dense_rows(x, T, H, bget(m, (size_t)H * H, "encoder.layer.%d.attention.self.query.weight", l), bget(m, H, "encoder.layer.%d.attention.self.query.bias", l), H, q);
    dense_rows(x, T, H, bget(m, (size_t)H * H, "encoder.layer.%d.attention.self.key.weight", l), bget(m, H, "encoder.layer.%d.attention.self.key.bias", l), H, k);
    dense_rows(x, T, H, bget(m, (size_t)H * H, "encoder.layer.%d.attention.self.value.weight", l), bget(m, H, "encoder.layer.%d.attention.self.value.bias", l), H, v);
    if (m->missing) break;
    for (int h = 0; h < nh; ++h)
      for (int i = 0; i < T; ++i) {
        float sc[2048];
        float mx = -3.0e38f;
        for (int j = 0; j < T; ++j) {
          float s = 0.f;
          for (int d = 0; d < dk; ++d) s += q[(size_t)i * H + h * dk + d] * k[(size_t)j * H + h * dk + d];
          sc[j] = s * scale;
          if (sc[j] > mx) mx = sc[j];
        }
        float den = 0.f;
        for (int j = 0; j < T; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
        for (int d = 0; d < dk; ++d) {
          float a = 0.f;
          for (int j = 0; j < T; ++j) a += sc[j] * v[(size_t)j * H + h * dk + d];
          c[(size_t)i * H + h * dk + d] = a / den;
        }
      }
    dense_rows(c, T, H, bget(m, (size_t)H * H, "encoder.layer.%d.attention.output.dense.weight", l), bget(m, H, "encoder.layer.%d.attention.output.dense.bias", l), H, y);
    for (size_t i = 0; i < (size_t)T * H; ++i) x[i] += y[i];
    ln_rows(x, T, H, bget(m, H, "encoder.layer.%d.attention.output.LayerNorm.weight", l), bget(m, H, "encoder.layer.%d.attention.output.LayerNorm.bias", l), hp->ln_eps);
    dense_rows(x, T, H, bget(m, (size_t)F * H, "encoder.layer.%d.intermediate.dense.weight", l), bget(m, F, "encoder.layer.%d.intermediate.dense.bias", l), F, f);
    for (size_t i = 0; i < (size_t)T * F; ++i) f[i] = 0.5f * f[i] * (1.0f + erff(f[i] * 0.70710678118654752440f));
    dense_rows(f, T, F, bget(m, (size_t)H * F, "encoder.layer.%d.output.dense.weight", l), bget(m, H, "encoder.layer.%d.output.dense.bias", l), H, y);
    for (size_t i = 0; i < (size_t)T * H; ++i) x[i] += y[i];
    ln_rows(x, T, H, bget(m, H, "encoder.layer.%d.output.LayerNorm.weight", l), bget(m, H, "encoder.layer.%d.output.LayerNorm.bias", l), hp->ln_eps);
  }
  if (!m->missing) memcpy(out, x, sizeof(float) * (size_t)T * H);
  free(x); free(q); free(k); free(v); free(c); free(f); free(y);
  return m->missing ? VITS_ERR_BLOB : VITS_OK;
}
int SAPI(bert_create)(const void* blob, size_t bytes, int device, bert_model** out) {
  (void)device;
  if (!blob || !out || bytes < 16 + sizeof(bert_hparams)) return sfail(VITS_ERR_ARG, "bad blob argument");
  const unsigned char* p = (const unsigned char*)blob;
  if (memcmp(p, "BERTW001", 8) != 0) return sfail(VITS_ERR_BLOB, "bad magic");
  uint32_t hb; memcpy(&hb, p + 8, 4);
  if (hb != sizeof(bert_hparams)) return sfail(VITS_ERR_BLOB, "hparams size %u != %zu", hb, sizeof(bert_hparams));
  bert_model* m = (bert_model*)calloc(1, sizeof *m);
  memcpy(&m->hp, p + 12, sizeof(bert_hparams));
  if (m->hp.abi_version != BERT_ABI_VERSION || m->hp.max_position > 2048) { free(m); return sfail(VITS_ERR_BLOB, "abi version / size mismatch"); }
  m->blob = (unsigned char*)malloc(bytes);
  memcpy(m->blob, blob, bytes);
  memcpy(&m->n_entries, m->blob + 12 + hb, 4);
  m->entries = (const vits_blob_entry*)(m->blob + 16 + hb);
  *out = m;
  return VITS_OK;
}
void SAPI(bert_destroy)(bert_model* m) { if (m) { free(m->blob); free(m); } }
int SAPI(bert_get_hparams)(const bert_model* m, bert_hparams* out) {
  if (!m || !out) return sfail(VITS_ERR_ARG, "null argument");
  *out = m->hp;
  return VITS_OK;
}
