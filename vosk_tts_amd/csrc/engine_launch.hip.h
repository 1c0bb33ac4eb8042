// engine_launch.hip.h -- launch helpers and KERNEL SELECTION: which conv / attention / LayerNorm kernel a launch gets (launch_conv and friends).
// Part of the ONE translation unit engine.hip (included there, in order; not a standalone header): split out in round 6 so that the
// planner / launch selection / stages / host paths can be read on their own.
#pragma once
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
static thread_local int g_ks_waves = 0;
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
static thread_local int g_wp_mode = 0;  // 0 = heuristic, 1 = never, 2 = whenever eligible (tests)
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
static thread_local int g_no_bf3 = 0;   // test hook: 1 = a conv_precision == 1 model runs its fp32 kernels (A/B of the split-bf16 variant)
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
static thread_local int g_sp_mode = -1;
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
// stream-K schedule of the same tile (conv_sk.hip.h; prototype, VITS_SK): G persistent workgroups, equal-cost contiguous ranges
static bool sk_takes(vits_session* s, const ConvParams& P, int epi, int halo) {
  return sk_mode() && epi == EPI_STORE && s && s->sk_ws && s->sk_ctl && conv_sp_ok(P, epi, halo);
}
static void launch_sk(vits_session* s, ConvParams& P, int halo) {
  attach_tile_table(s, P, 64);
  P.ntiles_m = cdiv(P.M, 64);
  P.ntiles_n = cdiv(P.Tout, 64);
  P.row_len = 64 + halo;
  const long nblk = (long)P.ntiles_m * P.ntiles_n * P.B * P.n_groups;
  static const int gmax = getenv("VITS_SK_G") ? atoi(getenv("VITS_SK_G")) : 0;
  long G = gmax > 0 ? gmax : 2L * s->m->n_cu;  // two workgroups per CU are co-resident at every halo (73 KB of LDS at the widest)
  if (G > SK_SLOTS) G = SK_SLOTS;
  if (G > nblk) G = nblk;
  const size_t lds = (size_t)2 * 4 * P.row_len * SP_PITCH * sizeof(float);
  if (lds > 64 * 1024) {
    static std::atomic<unsigned long long> done1{0}, done2{0};
    if (P.row_len <= 64) { if (big_lds_needed(done1)) hipFuncSetAttribute((const void*)conv_sk_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    else if (big_lds_needed(done2)) hipFuncSetAttribute((const void*)conv_sk_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const SkArgs A{s->sk_ctl, s->sk_ws, 1 << 22};
  if (P.row_len <= 64) hipLaunchKernelGGL((conv_sk_kernel<1>), dim3((unsigned)G), dim3(256), lds, s->stream, P, A);
  else hipLaunchKernelGGL((conv_sk_kernel<2>), dim3((unsigned)G), dim3(256), lds, s->stream, P, A);
}
// the 64 x 64 tile of a launch that was routed to conv_mfma_kernel<2,2,1,1,EPI>: the pipelined kernel when the grid is small
static bool sp_takes(const ConvParams& P, int epi, int halo) {
  static const long max_blk = getenv("VITS_SP_MAXBLK") ? atol(getenv("VITS_SP_MAXBLK")) : 2048;
  const long nblk = (long)cdiv(P.M, 64) * cdiv(P.Tout, 64) * P.B * P.n_groups;
  return conv_sp_ok(P, epi, halo) && (sp_mode() == 2 || nblk <= max_blk);
}

// ---- independent-wave 64 x 64 tiles (conv_w1.hip.h): stands in for conv_mfma_kernel<2,2,2,2,STORE> (the ResBlock convs of a batch).
// MEASURED (profiles/r6_w1_ab.txt): parity green, 1.3 % SLOWER than the four-wave kernel on c3 / c4 (20.40 -> 20.66 ms, 127.3 -> 129.1):
// the chunk barrier is not what the 128 x 128 kernel loses -- an independent-wave form with no barrier at all lands on the same plateau
// (and moves 2.6 x the activation bytes through L2).  Kept as an A/B (like the producer / consumer split-bf16 kernel): VITS_W1=1 takes
// the launches the 128 x 128 kernel would; default off.
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
      // (only where the four-wave grid is 2 - 6 workgroups per CU: a 6000-frame single utterance -- 282 four-wave workgroups, about one
      //  per CU -- is 3 % SLOWER on two-wave tiles, profiles/r6_gate2w_ab.txt)
      if (g_force_tile == 0 && g2w && 32 + halo <= 64 && (g2w == 2 || (nblk64 >= 512 && nblk64 <= 1536))) {
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
  if (g_force_tile == 0 && sk_takes(s, P, epi, halo) && (sk_mode() == 2 || sp_takes(P, epi, halo))) { ps.set_kernel("conv_sk_kernel<STORE>"); launch_sk(s, P, halo); return; }
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

