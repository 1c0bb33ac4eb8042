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
  if (Cin % 16 || (K != 1 && K != 3 && K != 5)) return 0;  // (1, 3 or 5 taps: the kernel's tap-unit table, persist.hip.h)
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
  if (dk > PS_DKP || dk % 16 || hp.window_size < 0 || hp.window_size > 4 || (2 * hp.window_size + 1) * dk > 1024) return false;  // (dk % 16: the PV tiles of the MFMA attention blocks)
  for (const EncLayerW& L : E.layers)
    if (!persist_conv_ok(L.qkv) || !persist_conv_ok(L.o) || !persist_conv_ok(L.f1) || !persist_conv_ok(L.f2) || L.qkv.K != 1 || L.o.K != 1) return false;
  return true;
}

// Longest sequence a program is built for.  Up to P columns every column kind is one step (worker t = column t); beyond, the resolver
// cuts a column step into rounds of P columns and an attention step into rounds of P blocks (the kernel does not know: a step is
// whatever records it finds).  VITS_PS_MAX_T overrides (A/B of the launch path against the program at a given length).
static int persist_max_t() {
  static const int v = getenv("VITS_PS_MAX_T") ? atoi(getenv("VITS_PS_MAX_T")) : PS_MAX_T;
  return v < 16 ? 16 : (v > PS_MAX_T ? PS_MAX_T : v);
}
static bool persist_common_ok(const vits_model* m, int B, int T) {
  return m->acoustic && B == 1 && T >= 1 && T <= persist_max_t() && m->n_cu >= 16 && m->zeros;
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
  if (!persist_common_ok(m, B, Tx) || !persist_encoder_ok(m, m->enc_p) || !persist_conv_ok(m->enc_proj) || m->enc_proj.K != 1) return false;
  // BERT-conditioned flavour (round 5): x = (emb * sqrt(H) + bert_proj(bert)) * mask as two more steps (a K-sliced matrix step over the
  // plain `bert` tensor + a summing column step); the program is only BUILT for sessions that own a fixed bert buffer (ps_bert)
  if (hp.bert_dim > 0 && (!persist_conv_ok(m->bert_proj) || m->bert_proj.K != 1)) return false;
  return 4 + 8 * (int)m->enc_p.layers.size() <= PS_MAX_STEPS;
}
static size_t persist_enc_cells(const vits_model* m, int B, int Tx) {
  if (!persist_enc_eligible(m, B, Tx)) return 0;
  const size_t Tp = (size_t)cdiv(Tx, 16) * 16, H = m->hp.hidden_channels;
  size_t bert = 0;
  if (m->hp.bert_dim > 0) bert = Tp * H * (m->bert_proj.Cin / persist_slice(m->bert_proj.Cin, 1) + 1);  // K-slice partials + the sum
  return Tp * H + bert + m->enc_p.layers.size() * persist_enc_layer_cells(m, m->enc_p, Tp);
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

// ------------------------------------------------------------------------------------------------ host-side step descriptors
// What a builder says about one step; persist_resolve() turns it into one 128-byte record per worker (persist.hip.h).
enum { PS_EPI_STORE = 0, PS_EPI_SPLINE = 1, PS_EPI_GATE = 2 };
struct PStep {
  int kind;
  int Tp;                  // padded column count of the segment the step belongs to (set by PBuild::push)
  // ---- PK_MM: y[Cout x 16-column tile] (+)= W[Cout x ks*Cin*K] * window(B)
  int Cin;                 // contraction channels of ONE K-slice (multiple of 16, <= PS_MAXC)
  int cin_pitch;           // channel pitch of the operand cells
  int c_off, c_sign;       // operand channel of slice-local channel c: c_off + c_sign * (slice * Cin + c)   (Flip folded into the read)
  int Cout, n_mb;          // rows stored, 16-row blocks of the packed matrix
  int G, mbg, ks;          // row-block groups per column tile, 16-row blocks per worker, K-slices
  int K, pad;              // taps, left padding: operand column of (output t, tap kk) = t + kk - pad
  int epi, relu;           // PS_EPI_*; 1 = ReLU on acc + bias
  int in_mask, out_mask;   // operand columns >= len read as 0 ; output columns >= len written as 0
  int ypitch, y_off;       // output channel pitch, first output channel
  int gate_H;              // PS_EPI_GATE: hidden channels (packed rows = [8 tanh | 8 sigmoid] per 16-row block)
  int plain_T;             // row length of bin_plain / yplain / oplain / u_plain
  int zinit;               // also draw z = noise * noise_scale_w into zout (duration predictor, first step)
  int blen;                // valid entries of bias / cond
  const float* w16;        // [n_mb][ks][Cin/16*K][64][4] 16x16x4 A-fragment order (pack_conv_weights16: slices are consecutive units)
  const float* bias;       // [Cout] (zeros for K-sliced steps: the consumer adds it once)
  const float* cond;       // per-item bias rows (cond(g)) or zeros
  const ll_t* bin;         // operand cells [Tp][cin_pitch] ...
  const float* bin_plain;  // ... or plain floats [channels][plain_T] written by an earlier kernel (null: cells)
  const ll_t* res;         // residual cells [Tp][rpitch] added after the mask (null: none)
  int rpitch;
  ll_t* yout;              // cells [ks][Tp][ypitch] (null: no cell output)
  float* yplain;           // plain floats [Cout][plain_T] for later kernels (null: none)
  const ll_t* z;           // z cells [2][Tp]  (PS_EPI_SPLINE; flow layer 0 of PK_DDS)
  ll_t* zout;              // zinit / PS_EPI_SPLINE: z cells out (null for the last flow)
  int z_row;               // row of z that conditions (x0); the spline acts on 1 - z_row
  int last, ea_row;        // last flow: write logw = ElementwiseAffine^-1(z[ea_row]) (modules.py:293-295)
  // ---- column steps: C channels (threads); par[k]: per-channel parameter vectors, packed 8 per thread (value k of thread row is
  //      par[k][row + padd[k]], 0 outside [0, plen))
  int C, plen;
  int padd[8];
  const float* par[8];
  const float* vec;        // per-item vector (speaker embedding), runtime data: not packed
  // PK_DDS: x_in = (xin + gelu(LN(y2; par0, par1))) * mask   [fin 0: xin * mask; fin 2: par0 * z + par1 + xin]
  //         b    = gelu(LN(depthwise3(x_in; par3..5, par2, dil); par6, par7))          (dw != 0)
  int dil, dw, fin;
  const ll_t* xin; const ll_t* y2;   // [Tp][C]
  ll_t* xout;              // x_in column
  ll_t* bout;              // b column (dw != 0)
  // PK_LN: v = par2 (bias) + sum_{k < np} part[k][t][c] + res[t][c] ; out = (LN(v; par0, par1) + vec + base[t][c]) * mask
  int np, ln;
  const ll_t* part; long long part_stride;   // cells [np][Tp][C]
  const ll_t* base;        // residual base added AFTER the norm (flow: h + pre_transformer(h)) or null
  // PK_EMB: out = emb[ids[t]][c] * scale * mask (+ vec)        (models.py:318-322)
  const float* emb; float scale; int n_vocab;
  // PK_ATT / PK_MERGE
  int nh, dk, W;           // heads, head dimension, relative-position window (0: no relative terms)
  const ll_t* qkv;         // cells [Tp][3*nh*dk]: q | k | v
  ll_t* ap;                // partial cells [key tile][Tp][nh][dk + 2]
  const float* ek; const float* ev;
  // PK_COUPLE: new z (Flip folded): out[r] = u[2H-1-r] (r < H) ; out[H + r] = (u[H-1-r] - (par2[r] + sum_k part[k][t][r])) * mask
  const ll_t* u; const float* u_plain;  // previous z: cells [Tp][2H] or plain floats [2H][plain_T]
  int H;
  // common outputs of column steps
  ll_t* out;               // cells [Tp][C] (null: none)
  float* oplain;           // plain floats [C][plain_T] (null: none)
};

// packed per-thread parameters (dst[row][k] = src[k][row + add[k]]), built once per distinct source tuple and kept with the model
static const float* persist_pack(vits_session* s, const float* const (&src)[8], const int (&add)[8], int len, int rows) {
  vits_model* m = s->m;
  std::vector<long long> key;
  for (int k = 0; k < 8; ++k) { key.push_back((long long)(uintptr_t)src[k]); key.push_back(add[k]); }
  key.push_back(len); key.push_back(rows);
  for (auto& pk : s->ps_pending)  // built earlier in this plan, on this session's stream: usable by this plan's programs, not yet published
    if (pk.first == key) return pk.second;
  float* dst = nullptr;
  {
    std::lock_guard<std::mutex> g(m->pack_mu);
    auto it = m->packs.find(key);
    if (it != m->packs.end()) return it->second;
    if (hipMalloc((void**)&dst, sizeof(float) * 8 * (size_t)rows) != hipSuccess) return nullptr;
    m->allocs.push_back(dst);
  }
  PsPackArgs a;
  for (int k = 0; k < 8; ++k) { a.src[k] = src[k] == m->zeros ? nullptr : src[k]; a.add[k] = add[k]; a.len[k] = len; }
  hipLaunchKernelGGL(ps_pack_kernel, dim3(cdiv(rows * 8, 256)), dim3(256), 0, s->stream, dst, a, rows);
  // Published (persist_publish_packs) only once it is written: another thread planning a session on ITS stream would otherwise find the
  // entry and launch a program that reads the pack before this stream has run the kernel.  Round 5: ONE synchronisation per plan (the
  // one persist_plan ends with) instead of one per pack under the model's mutex; two threads that plan the same model at the same time may
  // each build a pack -- the first to publish wins, the other copy stays with the model's allocations (a few KB, once).
  s->ps_pending.emplace_back(std::move(key), dst);
  return dst;
}
// after the stream that ran the pack kernels has been synchronised
static void persist_publish_packs(vits_session* s, bool ok) {
  if (ok && !s->ps_pending.empty()) {
    std::lock_guard<std::mutex> g(s->m->pack_mu);
    for (auto& pk : s->ps_pending) s->m->packs.emplace(pk.first, pk.second);  // (emplace keeps an entry another thread published first)
  }
  s->ps_pending.clear();
}

// ------------------------------------------------------------------------------------------------ program builders
struct PBuild {
  vits_model* m;
  std::vector<PStep>* steps;  // (reserved to PS_MAX_STEPS: references returned by push() stay valid)
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
    st.C = 16; st.plen = 256; st.plain_T = T; st.np = 0; st.nh = 1; st.dk = 16;
    st.w16 = st.bias = st.cond = m->zeros;
    for (int k = 0; k < 8; ++k) st.par[k] = m->zeros;
    return st;
  }
  PStep& push(const PStep& st) {
    if ((int)steps->size() >= PS_MAX_STEPS) { overflow = true; return steps->back(); }
    steps->push_back(st);
    steps->back().Tp = Tp;
    return steps->back();
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
    set_groups(st);
    flops += 2.0 * T * (double)W.M * W.Cin * W.K;
    return st;
  }
  // row-block groups so that a step has at most one work item per worker
  void set_groups(PStep& st) const {
    st.mbg = cdiv(st.n_mb * ntn * st.ks, m->n_cu);
    st.G = cdiv(st.n_mb, st.mbg);
    while (ntn * st.G * st.ks > m->n_cu && st.mbg < st.n_mb) { ++st.mbg; st.G = cdiv(st.n_mb, st.mbg); }
  }
  void col_par(PStep& st, int C, const float* p0, const float* p1, const float* p2, const float* vec) {
    st.C = C; st.plen = C;
    st.par[0] = p0 ? p0 : m->zeros; st.par[1] = p1 ? p1 : m->zeros; st.par[2] = p2 ? p2 : m->zeros;
    st.vec = vec;
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
  if (W > 0 && L.ek && L.ev) { st.ek = L.ek; st.ev = L.ev; }
  else st.W = 0;
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

// ---- steps -> one record per (step, worker)
static void persist_resolve(vits_session* s, const std::vector<PStep>& steps, int P, int seg_pstep, int& seg_step, std::vector<PRec>& recs, bool& bad) {
  vits_model* m = s->m;
  int Tp = 16, ntn = 1;  // (per step: a two-halves program changes geometry at seg_pstep)
  seg_step = -1;
  auto U = [](const void* p) { return (unsigned long long)(uintptr_t)p; };
  PRec idle;
  memset(&idle, 0, sizeof idle);
  idle.kf = PK_IDLE; idle.a1 = 16;
  for (int k = 6; k < 10; ++k) idle.p[k] = U(m->zeros);
  recs.clear();
  auto new_step = [&]() -> PRec* { recs.insert(recs.end(), (size_t)P, idle); return recs.data() + recs.size() - P; };
  // Which worker runs which matrix item.  The ntn column tiles of one (row-block group, K-slice) stream the SAME weight fragments;
  // in dispatch order (worker = item) they sit on ntn different XCDs -- block b runs on XCD b % 8 (MI355X_MICROARCH.md; a
  // placement for speed only, nothing depends on it) -- and every XCD's L2 fetches every fragment of the step from the fabric
  // (round 3 PMC: 200 MB per launch against 26 MB of weights).  Placement by weight group: group q -> XCD q % 8, its tiles on
  // consecutive workers of that XCD; groups beyond 8 * floor(P / 8 / ntn) fill the workers left over.  VITS_PS_XCD=0: dispatch order.
  static const bool xcd_place = !(getenv("VITS_PS_XCD") && atoi(getenv("VITS_PS_XCD")) == 0);
  std::vector<int> rank_of, used;
  auto place = [&](int Q) {
    const int items = Q * ntn;
    rank_of.assign(items, -1);
    if (!xcd_place || P % 8 || P / 8 < ntn) { for (int i = 0; i < items; ++i) rank_of[i] = i; return; }
    used.assign(P, 0);
    const int per = P / 8, cap = per / ntn;
    for (int q = 0; q < Q && q < 8 * cap; ++q)
      for (int j = 0; j < ntn; ++j) {
        const int rk = (q % 8) + 8 * ((q / 8) * ntn + j);
        rank_of[q * ntn + j] = rk;
        used[rk] = 1;
      }
    int nx = 0;
    for (int i = 8 * cap * ntn; i < items; ++i) {
      while (used[nx]) ++nx;
      rank_of[i] = nx;
      used[nx] = 1;
    }
  };
  for (size_t si = 0; si < steps.size(); ++si) {
    const PStep& st = steps[si];
    Tp = st.Tp; ntn = Tp / 16;
    if ((int)si == seg_pstep) seg_step = (int)(recs.size() / P);
    if (st.kind == PK_DUR) {
      PRec* R = new_step();
      PRec& r = R[0];
      r.kf = PK_DUR | (st.dw ? 0 : PF_PLAIN_IN);
      r.a1 = st.C; r.a2 = st.dil;
      r.p[0] = U(s->ps_x_logw); r.p[1] = U(s->dur); r.p[2] = U(s->cum); r.p[3] = U(s->ps_x_cum);
      r.p[4] = U(s->len_y); r.p[5] = U(s->ylen64); r.p[6] = U(s->ps_x_leny);
    } else if (st.kind == PK_EXPAND) {
      PRec* R = nullptr;
      for (int t = 0; t < Tp; ++t) {
        if (t % P == 0) R = new_step();  // (a round of P columns per step)
        PRec& r = R[t % P];
        r.kf = PK_EXPAND | (st.dw ? 0 : PF_PLAIN_IN);
        r.a1 = st.C; r.a2 = t; r.a3 = s->Tx;
        r.p[0] = st.dw ? U(s->ps_x_stats) : U(s->stats);
        r.p[1] = st.dw ? U(s->ps_x_cum) : U(s->cum);
        r.p[3] = U(st.out); r.p[4] = U(st.oplain);
        r.b[4] = st.plain_T;
      }
    } else if (st.kind == PK_MM) {
      const int items = ntn * st.G * st.ks;
      if (items > P) { bad = true; return; }
      PRec* R = new_step();
      const int n_u = st.Cin / 16 * st.K;
      const bool gate = st.epi == PS_EPI_GATE;
      if (n_u > 127 || st.mbg > 15) { bad = true; return; }  // (field widths of record dword b0)
      place(st.G * st.ks);
      for (int item = 0; item < items; ++item) {
        const int j = item % ntn, q = item / ntn, g = q % st.G, slice = q / st.G;
        const int n0 = j * 16, mb0 = g * st.mbg, nblk = std::min(st.mbg, st.n_mb - mb0);
        PRec& r = R[rank_of[q * ntn + j]];
        r.kf = PK_MM | (st.relu ? PF_RELU : 0) | (st.in_mask ? PF_INMASK : 0) | (st.out_mask ? PF_OUTMASK : 0) | (gate ? PF_GATE : 0) |
               (st.epi == PS_EPI_SPLINE ? PF_SPLINE : 0) | ((st.zinit && g == 0 && slice == 0) ? PF_ZINIT : 0) | (st.last ? PF_LAST : 0) |
               (st.bin_plain ? PF_PLAIN_IN : 0);
        r.a1 = st.Cin | (st.K << 16);
        r.a2 = st.bin_plain ? st.c_sign : st.cin_pitch * st.c_sign;
        r.a3 = n0 - st.pad;
        const int ch0 = st.c_off + st.c_sign * slice * st.Cin;
        if (st.bin_plain) r.p[0] = U(st.bin_plain + (size_t)ch0 * st.plain_T);
        else r.p[0] = U(st.bin + (st.c_sign > 0 ? ch0 : ch0 - (st.Cin - 1)));
        const int row0 = gate ? mb0 * 8 : mb0 * 16;
        r.p[1] = st.yout ? U(st.yout + ((size_t)slice * Tp + n0) * st.ypitch + st.y_off + row0) : 0;
        r.p[2] = st.yplain ? U(st.yplain + (size_t)row0 * st.plain_T + n0) : 0;
        r.p[3] = st.res ? U(st.res + (size_t)n0 * st.rpitch + row0) : 0;
        r.p[4] = U(st.z); r.p[5] = U(st.zout);
        r.p[6] = U(st.w16 + ((size_t)mb0 * st.ks + slice) * n_u * 256);
        r.p[7] = U(st.bias + (st.bias == m->zeros ? 0 : row0));
        r.p[8] = U(st.cond + (st.cond == m->zeros ? 0 : row0));
        r.b[0] = n_u | (nblk << 7);  // (+ the next matrix item of this worker: chained below)
        r.b[1] = st.ks * n_u * 256;
        r.b[2] = st.ypitch;
        r.b[3] = gate ? 2 * st.gate_H : st.Cout - row0;
        r.b[4] = st.plain_T; r.b[5] = st.rpitch;
        r.b[6] = n0 | (st.z_row << 16) | (st.ea_row << 20);
        r.b[7] = st.gate_H;
      }
    } else if (st.kind == PK_ATT) {
      const int items = st.nh * ntn * ntn;
      const int add[8] = {0, 512, 0, 512, 0, 0, 0, 0};
      const float* const src[8] = {st.ek, st.ek, st.ev, st.ev, nullptr, nullptr, nullptr, nullptr};
      const float* pack = m->zeros;
      if (st.W > 0) {
        // rows = thread ids: {E_k[tid], E_k[tid + 512], E_v[tid], E_v[tid + 512]}; the two tables are packed by two calls (one length each)
        const float* const sk[8] = {st.ek, st.ek, st.ev, st.ev, m->zeros, m->zeros, m->zeros, m->zeros};
        (void)src;
        pack = persist_pack(s, sk, add, (2 * st.W + 1) * st.dk, 512);
        if (!pack) { bad = true; return; }
      }
      for (int i0 = 0; i0 < items; i0 += P) {
        PRec* R = new_step();
        for (int item = i0; item < std::min(items, i0 + P); ++item) {
          const int kt = item % ntn, qt = (item / ntn) % ntn, hd = item / (ntn * ntn);
          PRec& r = R[item - i0];
          r.kf = PK_ATT;
          r.a1 = st.dk | (st.nh << 8) | (st.W << 16);
          r.a2 = (qt * 16) | ((kt * 16) << 16);
          r.a3 = hd | (kt << 8);
          r.p[0] = U(st.qkv); r.p[1] = U(st.ap); r.p[9] = U(pack);
        }
      }
    } else {
      PRec* R = nullptr;
      const float* pack = m->zeros;
      if (st.kind != PK_MERGE && st.kind != PK_EMB) {
        pack = persist_pack(s, st.par, st.padd, st.plen, 256);
        if (!pack) { bad = true; return; }
      }
      for (int t = 0; t < Tp; ++t) {
        if (t % P == 0) R = new_step();  // (a round of P columns per step)
        PRec& r = R[t % P];
        r.a1 = st.C; r.a2 = t; r.p[9] = U(pack);
        r.p[3] = U(st.out); r.p[4] = U(st.oplain);
        r.b[4] = st.plain_T;
        if (st.kind == PK_DDS) {
          r.kf = PK_DDS | (st.dw ? PF_DW : 0) | (st.fin == 1 ? PF_FIN_LN : 0) | (st.fin == 2 ? PF_FIN_PRE : 0);
          r.a1 = st.C | (st.dil << 16);
          r.p[0] = U(st.xin); r.p[1] = U(st.y2); r.p[2] = st.z ? U(st.z + (size_t)st.z_row * Tp) : 0; r.p[3] = U(st.xout); r.p[4] = U(st.bout);
        } else if (st.kind == PK_LN) {
          r.kf = PK_LN | (st.ln ? PF_LN : 0);
          r.a3 = st.np;
          r.p[0] = U(st.part); r.p[1] = U(st.res); r.p[2] = U(st.base);
          r.p[8] = U(st.vec ? st.vec : m->zeros);
          r.b[0] = (int)(unsigned)(st.part_stride & 0xffffffffll); r.b[1] = (int)(st.part_stride >> 32);
        } else if (st.kind == PK_EMB) {
          r.kf = PK_EMB;
          r.p[0] = U(st.emb);
          r.p[8] = U(st.vec ? st.vec : m->zeros);
          memcpy(&r.b[0], &st.scale, 4);
          r.b[1] = st.n_vocab;
        } else if (st.kind == PK_MERGE) {
          r.kf = PK_MERGE;
          r.a3 = st.dk | (st.nh << 8);
          r.p[0] = U(st.ap);
          r.b[0] = Tp * st.nh * (st.dk + 2);
        } else {  // PK_COUPLE
          r.kf = PK_COUPLE | (st.u_plain ? PF_PLAIN_IN : 0);
          r.a3 = st.np | (st.H << 16);
          r.p[0] = U(st.part); r.p[1] = U(st.u); r.p[2] = U(st.u_plain);
          r.b[0] = (int)(unsigned)(st.part_stride & 0xffffffffll); r.b[1] = (int)(st.part_stride >> 32);
        }
      }
    }
    if ((int)(recs.size() / P) > PS_MAX_STEPS) { bad = true; return; }
  }
  // chain: every matrix record says where the same worker's NEXT matrix item streams its weights from (the kernel pulls them into L2
  // while it computes this one)
  const int nst = (int)(recs.size() / P);
  for (int w = 0; w < P; ++w) {
    unsigned long long nw = 0;
    int nx = 0;
    for (int si = nst - 1; si >= 0; --si) {
      PRec& r = recs[(size_t)si * P + w];
      if ((r.kf & 0xff) != PK_MM) continue;
      const int n_u = r.b[0] & 0x7f, nblk = (r.b[0] >> 7) & 0xf, kb = r.b[1] / 256;  // b1 = floats between row blocks = ks * n_u KiB
      r.p[9] = nw;
      r.b[0] |= nx << 11;
      nw = r.p[6];
      nx = (kb < 512 && n_u < 128) ? (n_u | (nblk << 7) | (kb << 11)) : 0;
    }
  }
}

static void persist_upload(vits_session* s, vits_session::PersistProg& pp, PBuild& b) {
  pp.flops = b.flops;
  pp.ok = false;
  if (b.overflow) return;
  vits_model* m = s->m;
  const int P = m->n_cu;
  std::vector<PRec> recs;
  bool bad = false;
  int seg_step = -1;
  persist_resolve(s, *b.steps, P, pp.h.seg_step, seg_step, recs, bad);
  if (bad || recs.empty() || (pp.h.seg_step >= 0 && seg_step < 0)) return;
  pp.h.seg_step = seg_step;
  const size_t bytes = recs.size() * sizeof(PRec);
  if (bytes > pp.recs_bytes) {
    if (pp.recs_d) { hipStreamSynchronize(s->stream); hipFree(pp.recs_d); pp.recs_d = nullptr; pp.recs_bytes = 0; }
    if (hipMalloc((void**)&pp.recs_d, bytes) != hipSuccess) { pp.recs_d = nullptr; return; }
    pp.recs_bytes = bytes;
  }
  if (!pp.d && hipMalloc((void**)&pp.d, sizeof(PProgram)) != hipSuccess) { pp.d = nullptr; return; }
  pp.h.n_steps = (int)(recs.size() / P);
  pp.h.P = P;
  pp.h.recs = pp.recs_d;
  pp.kinds.clear();
  for (int i = 0; i < pp.h.n_steps; ++i) {  // (for tools/ps_trace.py: the kind of a step = the kind of its first busy worker)
    int k = 0;
    for (int w = 0; w < P && !k; ++w) k = recs[(size_t)i * P + w].kf & 0xff;
    pp.kinds.push_back(k);
  }
  // (pageable sources: both copies complete before persist_plan returns -- it synchronises the stream)
  pp.recs_h.swap(recs);
  if (hipMemcpyAsync(pp.recs_d, pp.recs_h.data(), bytes, hipMemcpyHostToDevice, s->stream) != hipSuccess) return;
  if (hipMemcpyAsync(pp.d, &pp.h, sizeof(PProgram), hipMemcpyHostToDevice, s->stream) != hipSuccess) return;
  pp.ok = true;
}
// geometry / exchange region of the segment that is being built
static void seg_geom(PBuild& b, int T) { b.T = T; b.Tp = cdiv(T, 16) * 16; b.ntn = b.Tp / 16; }
static void seg_cells(PBuild& b, vits_session::PersistProg& region) { b.cur = region.ll; b.end = region.ll + region.cells; }

static void persist_begin(vits_session* s, vits_session::PersistProg& pp, PBuild& b, std::vector<PStep>& steps, int T, const int* len) {
  vits_model* m = s->m;
  memset(&pp.h, 0, sizeof(PProgram));
  steps.clear();
  steps.reserve(PS_MAX_STEPS + 1);
  b.m = m; b.steps = &steps; b.cur = pp.ll; b.end = pp.ll + pp.cells;
  seg_geom(b, T);
  PProgram& P = pp.h;
  P.T = T; P.Tp = b.Tp;
  P.nb = m->hp.dp_num_bins; P.bound = m->hp.dp_tail_bound; P.inv_sqrt_d = 1.0f / sqrtf((float)m->hp.dp_filter_channels);
  P.len = len; P.ea_m = m->ea_m ? m->ea_m : m->zeros; P.ea_logs = m->ea_logs ? m->ea_logs : m->zeros; P.logw = s->logw; P.err = s->d_err;
  P.seg_step = -1;
}

// ---- segments (a program is one segment, or several laid one after the other: persist_build_front / _back / _full)
// text encoder (models.py:317-326): embedding, n_layers encoder layers, proj -> s->x (plain, masked) and s->stats (plain); with
// stats_cells also as cells [Tp][2I] for a PK_EXPAND step of the same launch.  Returns the encoder output as cells [Tp][H].
static const ll_t* seg_enc(vits_session* s, PBuild& b, ll_t* stats_cells) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  const int H = hp.hidden_channels, n = (int)m->enc_p.layers.size();
  const int cond_layer = (m->use_g && m->cond_enc_off >= 0) ? hp.enc_cond_layer : -1;
  const float* vec = cond_layer >= 0 ? s->condv + m->cond_enc_off : nullptr;
  const bool bert = hp.bert_dim > 0;  // (persist_plan builds this program for such a voice only when the session owns a bert buffer)
  PStep st = b.blank(PK_EMB);
  b.col_par(st, H, nullptr, nullptr, nullptr, (cond_layer == 0 && !bert) ? vec : nullptr);
  st.emb = m->emb; st.scale = sqrtf((float)H); st.n_vocab = hp.n_vocab;
  st.out = b.take_rows(H);
  const ll_t* x = b.push(st).out;
  if (bert) {
    // x = (emb * sqrt(H) + bert_proj(bert)) * mask  (vosk_tts/synth.py:88-99; run_text_encoder): bert is a PLAIN tensor [bert_dim][T_x
    // bucket] the graph's memcpy node fills (front sessions of the host path); 768 channels = 3 K-slices, summed by a column step
    st = b.mm(m->bert_proj, nullptr, hp.bert_dim);
    st.bin_plain = s->ps_bert;
    const int ks = st.ks;
    st.yout = b.take_rows((size_t)ks * H);
    st.ypitch = H;
    const ll_t* part = b.push(st).yout;
    st = b.blank(PK_LN);
    b.col_par(st, H, nullptr, nullptr, ks > 1 ? m->bert_proj.bias : nullptr, cond_layer == 0 ? vec : nullptr);
    st.np = ks; st.ln = 0; st.part = part; st.part_stride = (long long)b.Tp * H; st.res = x;
    st.out = b.take_rows(H);
    x = b.push(st).out;
  }
  for (int i = 0; i < n; ++i)
    x = persist_encoder_layer(b, m->enc_p.layers[i], m->enc_p, x, (i + 1 == cond_layer) ? vec : nullptr, nullptr, i == n - 1 ? s->x : nullptr);
  // stats = proj(x * mask) * mask   (models.py:324-325)
  st = b.mm(m->enc_proj, x, H);
  st.in_mask = 1; st.out_mask = 1;
  st.yplain = s->stats;
  st.yout = stats_cells;
  b.push(st);
  return x;
}

// stochastic duration predictor, reverse (models.py:56-63,93-101); x_cells: the text encoder output of the SAME launch (null: the plain
// s->x an earlier launch left)
static void seg_sdp(vits_session* s, PBuild& b, const ll_t* x_cells) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  std::vector<PStep>& steps = *b.steps;
  const int D = hp.dp_filter_channels, nl = (int)m->dp_dds.pw.size(), K = hp.dp_kernel_size, H = hp.hidden_channels;
  // dp.pre (+ cond(g)) -> x0 ; z = noise * noise_scale_w          (models.py:58-60,96)
  PStep st = b.mm(m->dp_pre, x_cells, H);
  if (!x_cells) st.bin_plain = s->x;
  if (m->use_g && m->cond_dp_off >= 0) st.cond = s->condv + m->cond_dp_off;
  st.zinit = 1;
  st.yout = b.take_rows(D); st.zout = b.take_rows(2);
  const size_t first = steps.size();
  const ll_t* x = b.push(st).yout;
  const ll_t* z = steps[first].zout;
  // one DDSConv stack + the 1x1 conv that consumes it (modules.py:96-108): per layer a column step (finish the previous layer,
  // depthwise conv, LN1, GELU) and a matrix step (the layer's 1x1 conv); then the last finish and the projection
  auto stack = [&](const DDSW& Wd, const ConvW& proj, bool spline, const ll_t* xin, const ll_t* zc, int z_row, const float* pw, const float* pb) -> PStep& {
    const ll_t* y2 = nullptr;
    int dil = 1;
    for (int i = 0; i <= nl; ++i) {
      const bool fin = i == nl;
      st = b.blank(PK_DDS);
      st.C = D; st.plen = D; st.dw = fin ? 0 : 1; st.dil = fin ? 0 : dil;
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
        if (spline) { st.epi = PS_EPI_SPLINE; st.mbg = st.n_mb; st.G = 1; }  // all rows of a column tile in one worker: it runs the spline
        else st.out_mask = 1;  // proj(x) * x_mask (models.py:63)
        return b.push(st);
      }
      st.yout = b.take_rows(D);
      y2 = b.push(st).yout;
      dil *= K;
    }
    return steps[first];  // not reached
  };
  {
    PStep& pj = stack(m->dp_dds, m->dp_proj, false, x, nullptr, 0, nullptr, nullptr);
    pj.yout = b.take_rows(D);
  }
  const ll_t* dc = steps.back().yout;
  int swap = 0;
  for (int k = hp.dp_n_flows - 1; k >= 1; --k) {
    swap ^= 1;  // Flip (modules.py:270-277) is a row relabel on the 2-channel z
    const ConvFlowW& c = m->cf[k];
    PStep& pj = stack(c.dds, c.proj, true, dc, z, swap, c.pre_w, c.pre_b);
    pj.z = z; pj.z_row = swap;
    if (k > 1) { pj.zout = b.take_rows(2); z = pj.zout; }
    else { pj.last = 1; pj.ea_row = swap ^ 1; }
  }
}

// flow, reverse (models.py:750-757, 374-393): z_p -> z (plain, s->zB); zp_cells: the prior sample of the SAME launch (null: plain s->zA)
static void seg_flow(vits_session* s, PBuild& b, const ll_t* zp_cells) {
  vits_model* m = s->m;
  const vits_hparams& hp = m->hp;
  const int H = hp.hidden_channels, I = hp.inter_channels, half = I / 2, L = hp.flow_wn_layers;
  const ll_t* u = zp_cells;  // previous z as cells (null: the plain z_p of the first layer)
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
      b.set_groups(st);
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
}

// PK_DUR: durations / cumsum / frame count of the utterance (one worker).  with_logw: wait for the duration predictor of this launch
static void seg_dur(vits_session* s, PBuild& b, bool with_logw, int Tcap) {
  PStep st = b.blank(PK_DUR);
  st.C = s->Tx; st.dil = Tcap; st.dw = with_logw ? 1 : 0;
  b.push(st);
}
// PK_EXPAND: z_p of every frame.  cells: stats / cum / frame count come from THIS launch (else: the plain arrays an earlier launch left)
static const ll_t* seg_expand(vits_session* s, PBuild& b, bool cells) {
  PStep st = b.blank(PK_EXPAND);
  st.C = s->m->hp.inter_channels; st.dw = cells ? 1 : 0;
  st.out = s->ps_x_zp;
  st.oplain = s->zA;
  return b.push(st).out;
}

static void persist_build_enc(vits_session* s) {
  vits_session::PersistProg& pp = s->ps_enc;
  PBuild b;
  std::vector<PStep> steps;
  persist_begin(s, pp, b, steps, s->Tx, s->len_x);
  seg_enc(s, b, nullptr);
  persist_upload(s, pp, b);
}
static void persist_build_sdp(vits_session* s) {
  vits_session::PersistProg& pp = s->ps_sdp;
  PBuild b;
  std::vector<PStep> steps;
  persist_begin(s, pp, b, steps, s->Tx, s->len_x);
  seg_sdp(s, b, nullptr);
  persist_upload(s, pp, b);
}
static void persist_build_flow(vits_session* s) {
  vits_session::PersistProg& pp = s->ps_flow;
  PBuild b;
  std::vector<PStep> steps;
  persist_begin(s, pp, b, steps, s->Ty, s->len_y);
  seg_flow(s, b, nullptr);
  persist_upload(s, pp, b);
}

// ---- programs of the graph-replayed paths: the stages of one forward in ONE launch each side of the host's T_y round trip, or in one
// launch altogether where the caller brings the frame capacity (device sessions).  They use the exchange regions of the single-stage
// programs (a session never runs two programs at once) plus the small region ps_x (stats / logw / cum / frame count / z_p cells).
//   front = text encoder [+ duration predictor] + PK_DUR            (phase 1 of vits_synthesize*)
//   back  = PK_EXPAND + flow                                          (phase 2)
//   full  = front + back, frame capacity = the session's T_y          (vits_session_synthesize_device)
static void persist_build_front(vits_session* s, vits_session::PersistProg& pp, bool with_sdp, int Tcap, bool and_back) {
  PBuild b;
  std::vector<PStep> steps;
  persist_begin(s, pp, b, steps, s->Tx, s->len_x);
  pp.h.logw_cells = s->ps_x_logw;
  seg_cells(b, s->ps_enc);
  const ll_t* x = seg_enc(s, b, and_back ? s->ps_x_stats : nullptr);
  if (with_sdp) { seg_cells(b, s->ps_sdp); seg_sdp(s, b, x); }
  seg_dur(s, b, with_sdp, Tcap);
  if (and_back) {
    pp.h.seg_step = (int)steps.size();  // (= its index among the resolved steps as long as no earlier step splits: checked in persist_resolve)
    pp.h.T2 = s->Ty; pp.h.Tp2 = cdiv(s->Ty, 16) * 16; pp.h.len2 = s->ps_x_leny;
    seg_geom(b, s->Ty);
    const ll_t* zp = seg_expand(s, b, true);
    seg_cells(b, s->ps_flow);
    seg_flow(s, b, zp);
  }
  persist_upload(s, pp, b);
}
static void persist_build_back(vits_session* s, vits_session::PersistProg& pp) {
  PBuild b;
  std::vector<PStep> steps;
  persist_begin(s, pp, b, steps, s->Ty, s->len_y);
  const ll_t* zp = seg_expand(s, b, false);
  seg_cells(b, s->ps_flow);
  seg_flow(s, b, zp);
  persist_upload(s, pp, b);
}

// Called at every re-plan, outside any capture.  The exchange cells are zeroed (epoch 0 = "never written": whatever the arena held
// before must not look like a cell of a later forward); the epoch block survives re-plans, so epochs only ever grow.
// A failure here (allocation, upload) is not an error of the call: the affected programs stay !ok and their stages run on launches.
static void persist_plan(vits_session* s) {
  // EVERY program of the previous layout dies here, the composite ones included: their records point into the arena as it was laid
  // out before (a session that served an eligible shape and is then reserved for one beyond PS_MAX_T must not find a stale `ok`)
  auto clear_all = [&]() {
    s->ps_enc.ok = s->ps_sdp.ok = s->ps_flow.ok = s->ps_back.ok = false;
    for (int k = 0; k < 2; ++k) s->ps_front[k].ok = s->ps_full[k].ok = false;
  };
  clear_all();
  if (!s->ps_enc.cells && !s->ps_sdp.cells && !s->ps_flow.cells) return;
  auto give_up = [&]() {
    clear_all();
    s->ps_pending.clear();
    const hipError_t e = hipGetLastError();
    static std::atomic<bool> said{false};  // (once per process: the reserve does not fail, so say why single utterances are slower)
    if (!said.exchange(true) && !getenv("VITS_QUIET"))
      fprintf(stderr, "[vits_mi355] persistent programs could not be built (%s): their stages run the launch path\n", hipGetErrorString(e));
  };
  if (!s->ps_ctl) {
    if (hipMalloc((void**)&s->ps_ctl, sizeof(PersistCtl)) != hipSuccess) { s->ps_ctl = nullptr; return give_up(); }
    if (hipMemsetAsync(s->ps_ctl, 0, sizeof(PersistCtl), s->stream) != hipSuccess) return give_up();
  }
  for (vits_session::PersistProg* pp : {&s->ps_enc, &s->ps_sdp, &s->ps_flow})
    if (pp->cells && pp->ll && hipMemsetAsync(pp->ll, 0, pp->cells * sizeof(ll_t), s->stream) != hipSuccess) return give_up();
  if (s->ps_x.cells && s->ps_x.ll && hipMemsetAsync(s->ps_x.ll, 0, s->ps_x.cells * sizeof(ll_t), s->stream) != hipSuccess) return give_up();
  const bool enc_buildable = s->m->hp.bert_dim == 0 || s->ps_bert;  // a BERT-conditioned voice: only sessions with a fixed bert buffer
  if (s->ps_enc.cells && s->ps_enc.ll && enc_buildable) persist_build_enc(s);
  if (s->ps_sdp.cells && s->ps_sdp.ll) persist_build_sdp(s);
  if (s->ps_flow.cells && s->ps_flow.ll) persist_build_flow(s);
  if (s->ps_x.cells && s->ps_x.ll) {
    const bool enc = s->ps_enc.ok, sdp = s->ps_sdp.ok, flow = s->ps_flow.ok;
    if (enc) {
      persist_build_front(s, s->ps_front[0], false, 0, false);
      if (sdp) persist_build_front(s, s->ps_front[1], true, 0, false);
    }
    if (flow) persist_build_back(s, s->ps_back);
    if (enc && flow && s->Ty > 1) {
      persist_build_front(s, s->ps_full[0], false, s->Ty, true);
      if (sdp) persist_build_front(s, s->ps_full[1], true, s->Ty, true);
    }
  }
  // the programs live in pageable memory of the session: the copies must not outlive this call's view
  const bool synced = hipStreamSynchronize(s->stream) == hipSuccess;
  persist_publish_packs(s, synced);
  if (!synced) give_up();
}
