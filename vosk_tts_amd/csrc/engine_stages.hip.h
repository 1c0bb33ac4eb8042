// engine_stages.hip.h -- the stages of SynthesizerTrn.infer as launch sequences: text encoder, duration predictor, length regulator, flow, decoder.
// Part of the ONE translation unit engine.hip (included there, in order; not a standalone header): split out in round 6 so that the
// planner / launch selection / stages / host paths can be read on their own.
#pragma once
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
  const long long base = g_ps_rearm_base_ns.load();
  if (!iv || !at || now - at > 10 * iv) iv = base;                        // first timeout, or the last re-arm held: start over
  else if (iv < 64 * base) iv *= 2;                        // timed out again soon after a re-arm: back off
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


