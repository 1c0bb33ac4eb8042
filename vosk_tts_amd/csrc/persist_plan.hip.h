// persist_plan.hip.h — host side of the persistent step programs (persist.hip.h): eligibility, the size of the exchange buffers,
// the program builders for the three single-utterance stages (text encoder, stochastic duration predictor, flow) and the launch.
// Included by engine.hip after the model / session types.  Programs are rebuilt at every re-plan of a session (outside any capture).
#pragma once

#define PERSIST_SDP 1
#define PERSIST_ENC 2
#define PERSIST_FLOW 4

// contraction channels of one K-slice of a conv with C_in channels and K taps: the largest divisor of C_in (multiple of 16) that
// keeps a worker's operand window <= PS_MAXC channels and its weights within PS_MAXU tap units per wave; 0 = does not fit
static int persist_slice(int Cin, int K) {
  if (Cin % 16 || K < 1 || K > 5) return 0;
  for (int ks = 1; ks <= 4; ++ks) {
    if (Cin % ks) continue;
    const int cs = Cin / ks;
    if (cs % 16 == 0 && cs <= PS_MAXC && (cs / 16) * K <= PS_MAXU * PS_WAVES) return cs;
  }
  return 0;
}
static bool persist_conv_ok(const ConvW& W) { return W.w16 && persist_slice(W.Cin, W.K) > 0; }

static bool persist_encoder_ok(const vits_model* m, const EncoderW& E) {
  const vits_hparams& hp = m->hp;
  if (E.layers.empty() || E.H % 16 || E.H > PS_MAXC || hp.n_heads < 1 || hp.n_heads > 4 || E.H % hp.n_heads) return false;
  const int dk = E.H / hp.n_heads;
  if (dk > PS_DKP || dk % 2 || hp.window_size < 0 || hp.window_size > 4 || (2 * hp.window_size + 1) * dk > 1024) return false;
  for (const EncLayerW& L : E.layers)
    if (!persist_conv_ok(L.qkv) || !persist_conv_ok(L.o) || !persist_conv_ok(L.f1) || !persist_conv_ok(L.f2) || L.qkv.K != 1 || L.o.K != 1) return false;
  return true;
}

static bool persist_common_ok(const vits_model* m, int B, int T) {
  return m->acoustic && B == 1 && T >= 1 && T <= 256 && m->n_cu >= 16 && m->zeros;
}

// ---- stochastic duration predictor
static bool persist_sdp_eligible(const vits_model* m, int B, int Tx) {
  const vits_hparams& hp = m->hp;
  if (!persist_common_ok(m, B, Tx)) return false;
  const int D = hp.dp_filter_channels, H = hp.hidden_channels;
  if (D % 32 || D > PS_MAXC || H % 16 || H > PS_MAXC || hp.dp_kernel_size != 3) return false;
  const int nl = (int)m->dp_dds.pw.size();
  if (nl < 1 || nl > 3 || hp.dp_n_flows < 2 || hp.dp_num_bins > 16 || 3 * hp.dp_num_bins - 1 > 32) return false;
  if (1 + 2 * (nl + 1) * hp.dp_n_flows > PS_MAX_STEPS) return false;
  if (!persist_conv_ok(m->dp_pre) || !persist_conv_ok(m->dp_proj) || (int)m->dp_dds.swk[0].size() != nl) return false;
  for (const ConvW& c : m->dp_dds.pw) if (!persist_conv_ok(c)) return false;
  for (int k = 1; k < hp.dp_n_flows; ++k) {
    if ((int)m->cf[k].dds.pw.size() != nl || (int)m->cf[k].dds.swk[0].size() != nl || !persist_conv_ok(m->cf[k].proj)) return false;
    for (const ConvW& c : m->cf[k].dds.pw) if (!persist_conv_ok(c)) return false;
  }
  return true;
}
static size_t persist_sdp_cells(const vits_model* m, int B, int Tx) {
  if (!persist_sdp_eligible(m, B, Tx)) return 0;
  const vits_hparams& hp = m->hp;
  const size_t Tp = (size_t)cdiv(Tx, 16) * 16, D = hp.dp_filter_channels, nl = m->dp_dds.pw.size(), nf = hp.dp_n_flows;
  // x0 (dp.pre); per DDSConv stack: per layer the finished input, the 1x1 operand and the 1x1 output, the last finish, and the
  // proj output (dc; the ConvFlow projections stay in LDS); z after init and after every flow
  return Tp * (D * (1 + nf * (3 * nl + 2)) + 2 * (nf + 1));
}

// ---- one attentions.Encoder layer: qkv, attention partials, merged attention, y1, x1, FFN hidden, FFN partials, output
static size_t persist_enc_layer_cells(const vits_model* m, const EncoderW& E, size_t Tp) {
  const size_t H = E.H, F = E.F, nh = m->hp.n_heads, dk = H / nh, ntn = Tp / 16;
  const size_t ks2 = E.layers[0].f2.Cin / persist_slice(E.layers[0].f2.Cin, E.layers[0].f2.K);
  return Tp * (3 * H + ntn * nh * (dk + 2) + H + H + H + F + ks2 * H + H);
}
static bool persist_enc_eligible(const vits_model* m, int B, int Tx) {
  const vits_hparams& hp = m->hp;
  if (!persist_common_ok(m, B, Tx) || hp.bert_dim > 0 || !persist_encoder_ok(m, m->enc_p) || !persist_conv_ok(m->enc_proj) || m->enc_proj.K != 1) return false;
  return 2 + 8 * (int)m->enc_p.layers.size() <= PS_MAX_STEPS;
}
static size_t persist_enc_cells(const vits_model* m, int B, int Tx) {
  if (!persist_enc_eligible(m, B, Tx)) return 0;
  const size_t Tp = (size_t)cdiv(Tx, 16) * 16;
  return Tp * m->hp.hidden_channels + m->enc_p.layers.size() * persist_enc_layer_cells(m, m->enc_p, Tp);
}

// ---- flow (ResidualCouplingTransformersBlock reverse, folded WaveNet tail)
static bool persist_flow_eligible(const vits_model* m, int B, int Ty) {
  const vits_hparams& hp = m->hp;
  if (!persist_common_ok(m, B, Ty) || hp.flow_n_flows < 1) return false;
  const int I = hp.inter_channels, H = hp.hidden_channels, L = hp.flow_wn_layers;
  if (I % 32 || I > PS_MAXC || L < 1 || L > 4 || H % 16 || H > PS_MAXC || (I / 2) % 16) return false;
  for (const CouplingW& C : m->flow) {
    if (!persist_conv_ok(C.pre) || C.pre.K != 1 || !persist_encoder_ok(m, C.enc) || C.enc.layers.size() != 1) return false;
    if ((int)C.in_layers.size() != L || (int)C.rsx.size() != L - 1 || !C.skip_post.w16 || C.skip_post.Cin != L * H) return false;
    for (const ConvW& c : C.in_layers) if (!persist_conv_ok(c) || persist_slice(c.Cin, c.K) != c.Cin) return false;
    for (const ConvW& c : C.rsx) if (!persist_conv_ok(c) || c.K != 1) return false;
  }
  return hp.flow_n_flows * (11 + 2 * L) <= PS_MAX_STEPS;
}
static size_t persist_flow_cells(const vits_model* m, int B, int Ty) {
  if (!persist_flow_eligible(m, B, Ty)) return 0;
  const vits_hparams& hp = m->hp;
  const size_t Tp = (size_t)cdiv(Ty, 16) * 16, H = hp.hidden_channels, I = hp.inter_channels, L = hp.flow_wn_layers;
  // per coupling layer: pre output, the encoder layer, stacked gate outputs, L - 1 residual streams, L post partials, the new z
  const size_t per = Tp * (H + L * H + (L - 1) * H + L * (I / 2) + I) + persist_enc_layer_cells(m, m->flow[0].enc, Tp);
  return per * hp.flow_n_flows;
}

// ------------------------------------------------------------------------------------------------ program builders
struct PBuild {
  vits_model* m;
  PProgram* P;
  ll_t* cur;
  ll_t* end;
  int T, Tp, ntn;
  double flops = 0;
  bool overflow = false;
  ll_t* take(size_t cells) {
    ll_t* p = cur;
    cur += cells;
    if (cur > end) overflow = true;
    return p;
  }
  ll_t* take_rows(size_t ch) { return take(ch * (size_t)Tp); }
  PStep blank(int kind) const {  // every prefetched pointer valid (zeros when unused)
    PStep st;
    memset(&st, 0, sizeof st);
    st.kind = kind;
    st.Cin = 16; st.cin_pitch = 16; st.c_sign = 1; st.Cout = 16; st.n_mb = 1; st.G = 1; st.mbg = 1; st.ks = 1; st.K = 1; st.blen = 16;
    st.C = 16; st.pmask = 255; st.plen = 256; st.plain_T = T; st.np = 0; st.nh = 1; st.dk = 16;
    st.w16 = st.bias = st.cond = m->zeros;
    for (int k = 0; k < 8; ++k) st.par[k] = m->zeros;
    return st;
  }
  PStep& push(const PStep& st) {
    if (P->n_steps >= PS_MAX_STEPS) { overflow = true; return P->steps[PS_MAX_STEPS - 1]; }
    P->steps[P->n_steps] = st;
    return P->steps[P->n_steps++];
  }
  // matrix step skeleton for conv W over cells `bin` (channel pitch cin_pitch, first channel c_off, direction c_sign)
  PStep mm(const ConvW& W, const ll_t* bin, int cin_pitch, int c_off = 0, int c_sign = 1) {
    PStep st = blank(PK_MM);
    const int cs = persist_slice(W.Cin, W.K);
    st.Cin = cs; st.ks = W.Cin / cs; st.cin_pitch = cin_pitch; st.c_off = c_off; st.c_sign = c_sign;
    st.K = W.K; st.pad = (W.K - 1) / 2;
    st.Cout = W.M; st.n_mb = cdiv(W.M, 16); st.blen = W.M;
    st.w16 = W.w16;
    st.bias = (st.ks == 1 && W.bias) ? W.bias : m->zeros;
    st.bin = bin;
    st.ypitch = cdiv(W.M, 16) * 16;
    st.mbg = cdiv(st.n_mb * ntn * st.ks, m->n_cu);
    st.G = cdiv(st.n_mb, st.mbg);
    flops += 2.0 * T * (double)W.M * W.Cin * W.K;
    return st;
  }
  void col_par(PStep& st, int C, const float* p0, const float* p1, const float* p2, const float* p3) {
    st.C = C; st.pmask = 255; st.plen = C;
    st.par[0] = p0 ? p0 : m->zeros; st.par[1] = p1 ? p1 : m->zeros; st.par[2] = p2 ? p2 : m->zeros; st.par[3] = p3 ? p3 : m->zeros;
  }
};

// One attentions.Encoder layer (attentions.py:48-65) on x cells [Tp][H] -> new x cells.
//   vec_next: per-item vector added to the OUTPUT (the speaker embedding that the reference adds before the next layer), or null
//   base: cells added after the last LayerNorm (flow: h + pre_transformer(h)), or null;  xplain: also write the output as plain floats
static const ll_t* persist_encoder_layer(PBuild& b, const EncLayerW& L, const EncoderW& E, const ll_t* x, const float* vec_next, const ll_t* base,
                                         float* xplain) {
  vits_model* m = b.m;
  const int H = E.H, F = E.F, nh = m->hp.n_heads, dk = H / nh, W = m->hp.window_size, Tp = b.Tp, ntn = b.ntn;
  // q | k | v
  PStep st = b.mm(L.qkv, x, H);
  st.yout = b.take_rows(3 * H);
  const ll_t* qkv = b.push(st).yout;
  // attention blocks -> partial (O, m, l) per (key tile, column, head)
  st = b.blank(PK_ATT);
  st.nh = nh; st.dk = dk; st.W = W; st.qkv = qkv;
  st.ap = b.take((size_t)ntn * Tp * nh * (dk + 2));
  if (W > 0 && L.ek && L.ev) {
    st.pmask = 511; st.plen = (2 * W + 1) * dk;
    st.par[0] = st.par[1] = L.ek; st.par[2] = st.par[3] = L.ev;
    st.padd[1] = 512; st.padd[3] = 512;
  } else st.W = 0;
  b.flops += 4.0 * (double)H * b.T * b.T;
  const ll_t* ap = b.push(st).ap;
  st = b.blank(PK_MERGE);
  st.C = H; st.nh = nh; st.dk = dk; st.ap = const_cast<ll_t*>(ap);
  st.out = b.take_rows(H);
  const ll_t* att = b.push(st).out;
  // y1 = x + conv_o(att)
  st = b.mm(L.o, att, H);
  st.res = x; st.rpitch = H;
  st.yout = b.take_rows(H);
  const ll_t* y1 = b.push(st).yout;
  // x1 = norm_layers_1(y1)
  st = b.blank(PK_LN);
  b.col_par(st, H, L.g1, L.b1, nullptr, nullptr);
  st.np = 1; st.ln = 1; st.part = y1; st.part_stride = 0;
  st.out = b.take_rows(H);
  const ll_t* x1 = b.push(st).out;
  // FFN (attentions.py:308-317): conv_1(pad(x * mask)) -> relu -> * mask -> conv_2(pad(.)) -> * mask
  st = b.mm(L.f1, x1, H);
  st.in_mask = 1; st.relu = 1; st.out_mask = 1;
  st.yout = b.take_rows(F);
  const ll_t* hcell = b.push(st).yout;
  st = b.mm(L.f2, hcell, cdiv(F, 16) * 16);
  st.in_mask = 1;
  const int ks = st.ks;
  st.yout = b.take_rows((size_t)ks * H);
  st.ypitch = H;
  const ll_t* part = b.push(st).yout;
  // x = norm_layers_2(x1 + ffn) (+ vec_next) (+ base), masked
  st = b.blank(PK_LN);
  b.col_par(st, H, L.g2, L.b2, ks > 1 ? L.f2.bias : nullptr, vec_next);
  st.np = ks; st.ln = 1; st.part = part; st.part_stride = (long long)Tp * H; st.res = x1;
  st.base = base;
  st.out = b.take_rows(H);
  st.oplain = xplain;
  return b.push(st).out;
}

static void persist_upload(vits_session* s, vits_session::PersistProg& pp, PBuild& b) {
  pp.flops = b.flops;
  pp.ok = false;
  if (b.overflow) return;
  if (!pp.d && hipMalloc((void**)&pp.d, sizeof(PProgram)) != hipSuccess) { pp.d = nullptr; return; }
  if (hipMemcpyAsync(pp.d, &pp.h, sizeof(PProgram), hipMemcpyHostToDevice, s->stream) != hipSuccess) return;
  pp.ok = true;
}
static void persist_begin(vits_session* s, vits_session::PersistProg& pp, PBuild& b, int T, const int* len) {
  vits_model* m = s->m;
  memset(&pp.h, 0, sizeof(PProgram));
  b.m = m; b.P = &pp.h; b.cur = pp.ll; b.end = pp.ll + pp.cells;
  b.T = T; b.Tp = cdiv(T, 16) * 16; b.ntn = b.Tp / 16;
  PProgram& P = pp.h;
  P.T = T; P.Tp = b.Tp; P.ntn = b.ntn;
  P.nb = m->hp.dp_num_bins; P.bound = m->hp.dp_tail_bound; P.inv_sqrt_d = 1.0f / sqrtf((float)m->hp.dp_filter_channels);
  P.len = len; P.ea_m = m->ea_m ? m->ea_m : m->zeros; P.ea_logs = m->ea_logs ? m->ea_logs : m->zeros; P.logw = s->logw; P.err = s->d_err;
}

// text encoder (models.py:317-326): embedding, n_layers encoder layers, proj -> s->x (plain, masked) and s->stats (plain)
static void persist_build_enc(vits_session* s) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  vits_session::PersistProg& pp = s->ps_enc;
  PBuild b;
  persist_begin(s, pp, b, s->Tx, s->len_x);
  const int H = hp.hidden_channels, n = (int)m->enc_p.layers.size();
  const int cond_layer = (m->use_g && m->cond_enc_off >= 0) ? hp.enc_cond_layer : -1;
  const float* vec = cond_layer >= 0 ? s->condv + m->cond_enc_off : nullptr;
  PStep st = b.blank(PK_EMB);
  b.col_par(st, H, nullptr, nullptr, nullptr, cond_layer == 0 ? vec : nullptr);
  st.emb = m->emb; st.scale = sqrtf((float)H); st.n_vocab = hp.n_vocab;
  st.out = b.take_rows(H);
  const ll_t* x = b.push(st).out;
  for (int i = 0; i < n; ++i)
    x = persist_encoder_layer(b, m->enc_p.layers[i], m->enc_p, x, (i + 1 == cond_layer) ? vec : nullptr, nullptr, i == n - 1 ? s->x : nullptr);
  // stats = proj(x * mask) * mask   (models.py:324-325)
  st = b.mm(m->enc_proj, x, H);
  st.in_mask = 1; st.out_mask = 1;
  st.yplain = s->stats;
  b.push(st);
  persist_upload(s, pp, b);
}

// stochastic duration predictor, reverse (models.py:56-63,93-101)
static void persist_build_sdp(vits_session* s) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  vits_session::PersistProg& pp = s->ps_sdp;
  PBuild b;
  persist_begin(s, pp, b, s->Tx, s->len_x);
  const int D = hp.dp_filter_channels, nl = (int)m->dp_dds.pw.size(), K = hp.dp_kernel_size;
  // dp.pre (+ cond(g)) -> x0 ; z = noise * noise_scale_w          (models.py:58-60,96)
  PStep st = b.mm(m->dp_pre, nullptr, 0);
  st.bin_plain = s->x;
  if (m->use_g && m->cond_dp_off >= 0) st.cond = s->condv + m->cond_dp_off;
  st.zinit = 1;
  st.yout = b.take_rows(D); st.zout = b.take_rows(2);
  const ll_t* x = b.push(st).yout;
  const ll_t* z = pp.h.steps[0].zout;
  // one DDSConv stack + the 1x1 conv that consumes it (modules.py:96-108): per layer a column step (finish the previous layer,
  // depthwise conv, LN1, GELU) and a matrix step (the layer's 1x1 conv); then the last finish and the projection
  auto stack = [&](const DDSW& Wd, const ConvW& proj, bool spline, const ll_t* xin, const ll_t* zc, int z_row, const float* pw, const float* pb) -> PStep& {
    const ll_t* y2 = nullptr;
    int dil = 1;
    for (int i = 0; i <= nl; ++i) {
      const bool fin = i == nl;
      st = b.blank(PK_DDS);
      st.C = D; st.pmask = 255; st.plen = D; st.dw = fin ? 0 : 1; st.dil = fin ? 0 : dil;
      st.xin = xin; st.y2 = y2;
      if (i > 0) { st.fin = 1; st.par[0] = Wd.g2[i - 1]; st.par[1] = Wd.b2[i - 1]; }
      else if (zc) { st.fin = 2; st.z = zc; st.z_row = z_row; st.par[0] = pw; st.par[1] = pb; }
      if (!fin) {
        st.par[2] = Wd.sb[i]; st.par[3] = Wd.swk[0][i]; st.par[4] = Wd.swk[1][i]; st.par[5] = Wd.swk[2][i];
        st.par[6] = Wd.g1[i]; st.par[7] = Wd.b1[i];
        st.bout = b.take_rows(D);
      }
      st.xout = b.take_rows(D);
      b.flops += 2.0 * b.T * (double)D * (fin ? 0 : K);
      const PStep& col = b.push(st);
      xin = col.xout;
      st = b.mm(fin ? proj : Wd.pw[i], fin ? col.xout : col.bout, D);
      if (fin) {
        if (spline) { st.epi = PS_EPI_SPLINE; st.mbg = st.n_mb; st.G = 1; }
        else st.out_mask = 1;  // proj(x) * x_mask (models.py:63)
        return b.push(st);
      }
      st.yout = b.take_rows(D);
      y2 = b.push(st).yout;
      dil *= K;
    }
    return pp.h.steps[0];  // not reached
  };
  {
    PStep& pj = stack(m->dp_dds, m->dp_proj, false, x, nullptr, 0, nullptr, nullptr);
    pj.yout = b.take_rows(D);
  }
  const ll_t* dc = pp.h.steps[pp.h.n_steps - 1].yout;
  int swap = 0;
  for (int k = hp.dp_n_flows - 1; k >= 1; --k) {
    swap ^= 1;  // Flip (modules.py:270-277) is a row relabel on the 2-channel z
    const ConvFlowW& c = m->cf[k];
    PStep& pj = stack(c.dds, c.proj, true, dc, z, swap, c.pre_w, c.pre_b);
    pj.z = z; pj.z_row = swap;
    if (k > 1) { pj.zout = b.take_rows(2); z = pj.zout; }
    else { pj.last = 1; pj.ea_row = swap ^ 1; }
  }
  persist_upload(s, pp, b);
}

// flow, reverse (models.py:750-757, 374-393): z_p (plain, s->zA) -> z (plain, s->zB)
static void persist_build_flow(vits_session* s) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  vits_session::PersistProg& pp = s->ps_flow;
  PBuild b;
  persist_begin(s, pp, b, s->Ty, s->len_y);
  const int H = hp.hidden_channels, I = hp.inter_channels, half = I / 2, L = hp.flow_wn_layers;
  const ll_t* u = nullptr;  // previous z as cells (null: the plain z_p of the first layer)
  for (int f = hp.flow_n_flows - 1; f >= 0; --f) {
    const CouplingW& C = m->flow[f];
    // h = pre(x0) * mask, x0[c] = u[I-1-c]  (models.py:375-376 after Flip)
    PStep st = b.mm(C.pre, u, I, I - 1, -1);
    if (!u) st.bin_plain = s->zA;
    st.out_mask = 1;
    st.yout = b.take_rows(H);
    const ll_t* fh = b.push(st).yout;
    // h = h + pre_transformer(h * mask)  (models.py:377)
    const ll_t* fx = persist_encoder_layer(b, C.enc.layers[0], C.enc, fh, nullptr, fh, nullptr);
    // WN (modules.py:148-176), folded tail: gate outputs of all layers stacked [Tp][L*H]; res_skip layer i < L-1 only updates x
    ll_t* acts = b.take_rows((size_t)L * H);
    for (int i = 0; i < L; ++i) {
      st = b.mm(C.in_layers[i], fx, H);
      st.in_mask = 1; st.epi = PS_EPI_GATE; st.gate_H = H; st.blen = 2 * H;
      st.n_mb = cdiv(2 * H, 16); st.Cout = 2 * H;
      st.mbg = cdiv(st.n_mb * b.ntn, m->n_cu); st.G = cdiv(st.n_mb, st.mbg);
      if (m->use_g) st.cond = s->condv + C.cond_off + i * 2 * H;
      st.yout = acts; st.ypitch = L * H; st.y_off = i * H;
      b.push(st);
      if (i == L - 1) break;
      st = b.mm(C.rsx[i], acts, L * H, i * H, 1);  // x = (x + res_acts) * mask
      st.res = fx; st.rpitch = H; st.out_mask = 1;
      st.yout = b.take_rows(H);
      fx = b.push(st).yout;
      // (the residual is added AFTER the mask in the shared epilogue; x is masked on entry, so (x + r) * mask == x + r * mask)
    }
    // m = post(sum of skips) in K-slices over the stacked gate outputs ; new z = cat(x0, (x1 - m) * mask), Flip folded
    st = b.mm(C.skip_post, acts, L * H);
    const int ks = st.ks;
    st.yout = b.take_rows((size_t)ks * half); st.ypitch = half;
    const ll_t* part = b.push(st).yout;
    st = b.blank(PK_COUPLE);
    b.col_par(st, I, nullptr, nullptr, ks > 1 ? C.skip_post.bias : nullptr, nullptr);
    st.plen = half; st.padd[2] = -half;  // thread c >= half reads bias[c - half]
    st.H = half; st.np = ks; st.part = part; st.part_stride = (long long)b.Tp * half;
    st.u = u; st.u_plain = u ? nullptr : s->zA;
    st.out = b.take_rows(I);
    if (f == 0) st.oplain = s->zB;
    u = b.push(st).out;
  }
  persist_upload(s, pp, b);
}

// Called at every re-plan, outside any capture.  The exchange cells are zeroed (epoch 0 = "never written": whatever the arena held
// before must not look like a cell of a later forward); the epoch block survives re-plans, so epochs only ever grow.
static int persist_plan(vits_session* s) {
  s->ps_enc.ok = s->ps_sdp.ok = s->ps_flow.ok = false;
  if (!s->ps_enc.cells && !s->ps_sdp.cells && !s->ps_flow.cells) return VITS_OK;
  if (!s->ps_ctl) {
    HIP_TRY(hipMalloc((void**)&s->ps_ctl, sizeof(PersistCtl)));
    HIP_TRY(hipMemsetAsync(s->ps_ctl, 0, sizeof(PersistCtl), s->stream));
  }
  for (vits_session::PersistProg* pp : {&s->ps_enc, &s->ps_sdp, &s->ps_flow})
    if (pp->cells && pp->ll) HIP_TRY(hipMemsetAsync(pp->ll, 0, pp->cells * sizeof(ll_t), s->stream));
  if (s->ps_enc.cells && s->ps_enc.ll) persist_build_enc(s);
  if (s->ps_sdp.cells && s->ps_sdp.ll) persist_build_sdp(s);
  if (s->ps_flow.cells && s->ps_flow.ll) persist_build_flow(s);
  HIP_TRY(hipStreamSynchronize(s->stream));  // the programs live in pageable memory of the session: the copies must not outlive this call's view
  return VITS_OK;
}
