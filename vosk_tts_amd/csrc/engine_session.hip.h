// engine_session.hip.h -- sessions: workspace arena, per-shape planning, the persistent-program owner token / lock lease, profiling records.
// Part of the ONE translation unit engine.hip (included there, in order; not a standalone header): split out in round 6 so that the
// planner / launch selection / stages / host paths can be read on their own.
#pragma once
// ------------------------------------------------------------------------------------ sessions
// stream-K prototype switch (conv_sk.hip.h): VITS_SK=1 the launches conv_sp_kernel<STORE> would take by size, 2 wherever it is eligible
#define SK_SLOTS 1024
static thread_local int g_sk_mode = -1;  // vits_debug_conv_sk: -1 = environment
static int sk_mode() {
  static const int v = getenv("VITS_SK") ? atoi(getenv("VITS_SK")) : 0;
  return g_sk_mode >= 0 ? g_sk_mode : v;
}
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
  // stream-K prototype (conv_sk.hip.h, VITS_SK=1): partial-accumulator cells of SK_SLOTS workgroups and the launch epoch block
  ll_t* sk_ws = nullptr;
  SkCtl* sk_ctl = nullptr;
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
  if (dev < 0 || dev >= 64 || g_tok_busy[dev] >= persist_owners()) return false;
  if (g_tok_busy[dev] == 0 && !persist_process_lock(dev)) return false;  // the process lease is taken by the first owner, dropped by the last
  ++g_tok_busy[dev];
  return true;
}
static void persist_token_release(int dev) {
  std::lock_guard<std::mutex> g(g_tok_mu);
  if (dev < 0 || dev >= 64) return;
  if (g_tok_busy[dev] > 0 && --g_tok_busy[dev] == 0) persist_process_unlock(dev);
}
// Host calls in flight per device (graph-replayed entry points, every batch size).  Round 6 (profiles/r6_owners.txt): a persistent program
// needs every CU's whole register file for one of its workgroups, so next to other calls' kernels it is placed late, polls long and --
// once resident -- keeps 256 CUs' issue slots busy with its spinning workers: with 4 closed-loop clients the launch path alone serves
// 2040 requests/s at p50 1.9 ms, "one call on the programs, three on launches" 1910 at 2.4 ms; with 2 clients it is the other way round
// (1460 against 1240 requests/s).  So a call takes the programs when at most `g_persist_when` OTHER host calls are in flight on its
// device as it starts (default 1; VITS_PERSIST_WHEN / vits_debug_persist_when; -1: whenever the token is free, the rounds 3-5 rule).
static std::atomic<int> g_calls_inflight[64];
struct InFlight {
  int dev; int before;
  explicit InFlight(int dev_) : dev(dev_ >= 0 && dev_ < 64 ? dev_ : -1), before(0) { if (dev >= 0) before = g_calls_inflight[dev].fetch_add(1); }
  ~InFlight() { if (dev >= 0) g_calls_inflight[dev].fetch_sub(1); }
};
static std::atomic<int> g_persist_when{getenv("VITS_PERSIST_WHEN") ? atoi(getenv("VITS_PERSIST_WHEN")) : 1};
static bool persist_quiet_enough(int others_in_flight) {
  const int t = g_persist_when.load(std::memory_order_relaxed);
  return t < 0 || others_in_flight <= t;
}
// a host call that launches AND waits for its kernels: owns the token (when it is free) from here to its end
struct PersistScope {
  int dev; bool own;
  explicit PersistScope(int dev_, int others_in_flight = 0) : dev(dev_) {
    const int cfg = persist_cfg();
    own = cfg != 0 && persist_quiet_enough(others_in_flight) && persist_token_try(dev_);
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
  if (sk_mode() && !s->sk_ws) {  // (outside any capture: the stream-K kernel's exchange buffers, once per session)
    void* p = nullptr; void* c = nullptr;
    if (hipMalloc(&p, sizeof(ll_t) * (size_t)SK_SLOTS * SK_CELLS) == hipSuccess && hipMalloc(&c, sizeof(SkCtl)) == hipSuccess) {
      hipMemsetAsync(p, 0, sizeof(ll_t) * (size_t)SK_SLOTS * SK_CELLS, s->stream);
      hipMemsetAsync(c, 0, sizeof(SkCtl), s->stream);
      s->sk_ws = static_cast<ll_t*>(p); s->sk_ctl = static_cast<SkCtl*>(c);
    } else { if (p) hipFree(p); if (c) hipFree(c); (void)hipGetLastError(); }
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
  if (s->sk_ws) hipFree(s->sk_ws);
  if (s->sk_ctl) hipFree(s->sk_ctl);
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

