// persist.hip.h — the single-utterance stochastic duration predictor as ONE persistent kernel (round 3).
//
// What it replaces: StochasticDurationPredictor.forward(reverse=True) (training/vits2/models.py:56-63,93-101; DDSConv
// modules.py:96-108, ConvFlow modules.py:363-390, spline transforms.py:55-177) at B = 1 is a chain of ~21 dependent launches on a
// [256 x T_x] tensor (16 x 11.96 us DDSConv layers + pre + init + 3 splines = 0.27 ms of the 1.33 ms forward for 0.4 % of its FLOPs).
// Every launch pays the dispatch, a cold L2 and its own chain of dependent cold misses.
//
// How: the whole predictor is a step program run by ONE kernel of P workgroups (one per CU) that never leave the machine.  Between
// steps there is NO barrier and NO flag: every exchanged tensor is an array of 8-byte "LL cells" {float value, u32 epoch}
// written with one agent-scope 8-byte store and polled by the consumers with agent-scope (L1-bypassing, sc1) 8-byte loads until the
// epoch matches this forward -- the data is its own arrival signal (tools/llprobe.hip: 0.32 us one-way inside an XCD, 0.73 us across
// XCDs, coherent chip-wide with sc1 stores; a produce / exchange / consume round of a [256 x 64] tensor costs 3.3-4.8 us with this
// protocol whether the workers sit on one XCD or on all eight, against 5.5-12 us per launch today).  A worker that is idle in a step
// simply moves on; a consumer waits only for the cells it reads.  Epochs make stale data harmless: every forward uses epoch =
// (last completed forward) + 1, cells are never reset, every step of a forward writes its OWN buffers (no reuse inside a forward,
// so there is no write-after-read hazard either), and a cell whose epoch does not match is simply not there yet.  Every poll loop is
// bounded (PS_SPIN_LIMIT): a lost worker turns into an error word, never into a hung GPU.
//
// Decomposition of a step (1x1 conv of the [C_in x T] layer input with a [M x C_in] matrix): work item = (16-column tile j,
// group of `mbg` 16-row blocks); worker r takes item r (items <= P by construction).  Every worker of a column tile gathers the full
// channel window it needs (all C_in channels x 16 + 2 dil columns: the depthwise taps' halo is recomputed, not exchanged), runs the
// layer's prologue on it (finish the previous layer: x + gelu(LN2(y2)); depthwise conv; LN1; GELU -- the same arithmetic as
// conv16_kernel's PRO == 1, conv_small.hip.h), then its 16 x 16 x C_in MFMA tile (v_mfma_f32_16x16x4_f32, exact fp32) with the NW
// waves splitting the contraction, weights prefetched into registers during the PREVIOUS step's MFMA phase.  ConvFlow.proj workers
// own all 29 rows of their columns and run the spline inverse in their epilogue; the last one folds the final ElementwiseAffine and
// writes logw.
#pragma once
#include "conv_small.hip.h"
#include "kernels_misc.hip.h"

typedef unsigned long long ll_t;  // {float value (bits 0..31), u32 epoch (bits 32..63)}

#define PS_THREADS 512
#define PS_WAVES 8
#define PS_MAX_STEPS 24
#define PS_XP 36         // LDS pitch of a window row (<= 16 + 2 * 9 columns)
#define PS_MAXC 256      // channels of an exchanged tensor / contraction length
#define PS_MAXI 16       // channels per thread in the window phase (PS_MAXC / 16)
#define PS_MAXU 2        // tap units (16 channels) per wave: PS_MAXC / 16 / PS_WAVES
#define PS_SPIN_LIMIT (1 << 18)
#define PS_ERR_TIMEOUT 8  // bit in the session error word

struct PersistCtl {
  unsigned epoch;     // epoch of the last completed forward
  unsigned done;      // workers that finished the current forward (the last one publishes the epoch and resets this)
  unsigned abort;     // a worker timed out: everybody stops polling
  unsigned timeouts;  // diagnostics
};

enum { PS_PRE = 0, PS_DDS = 1, PS_PROJ = 2, PS_CFPROJ = 3 };

struct SdpStep {
  int kind;
  int Cin, Cout, n_mb;     // contraction channels, rows stored, 16-row blocks of the packed matrix
  int G, mbg;              // workers per column tile, 16-row blocks per worker
  int dil;                 // PS_DDS: dilation of the 3-tap depthwise conv
  int z_row;               // flow layer 0: row of z that conditions (x0); PS_CFPROJ: the same flow's x0 row (the spline acts on 1 - z_row)
  int last, ea_row;        // PS_CFPROJ of the last flow: write logw = ElementwiseAffine^-1(z[ea_row]) (modules.py:293-295)
  const float* w16;        // [n_mb][Cin/16][64][4] 16x16x4 A-fragment order (pack_conv_weights16)
  const float* bias;
  const float* cond;       // PS_PRE: per-item bias rows (cond(g), models.py:60) or null
  const ll_t* xin;         // residual stream x [Cin][Tp]; flow layer 0: the conditioning tensor dc
  const ll_t* y2;          // previous layer's 1x1 output [Cin][Tp] (finish: x + gelu(LN(y2; g2, b2))) or null
  const float* g2; const float* b2;
  const float* sw; const float* sb; const float* g1; const float* b1;  // PS_DDS: depthwise + LN1
  const ll_t* z;           // flow layer 0 / PS_CFPROJ: z [2][Tp]
  const float* pw; const float* pb;  // flow layer 0: ConvFlow.pre (Conv1d(1, D, 1)), modules.py:365
  ll_t* yout;              // [Cout][Tp]
  ll_t* xout;              // PS_DDS: the finished layer input [Cin][Tp] (next layer's residual stream)
  ll_t* zout;              // PS_PRE: z = noise * noise_scale_w; PS_CFPROJ: transformed z
};

struct SdpProgram {
  int n_steps, T, Tp, ntn;
  int nb; float bound, inv_sqrt_d;      // spline
  const int* len;
  const float* ea_m; const float* ea_logs;
  float* logw;                          // [T] plain floats (the kernel's result)
  int* err;
  SdpStep steps[PS_MAX_STEPS];
};

struct SdpCall {                        // per-call values (by value: a captured graph re-reads `dv`, not these, when dv != null)
  PersistCtl* ctl;
  const float* x;                       // text-encoder output [H][T], plain floats
  const float* noise;                   // [2][T] injected noise or null (Philox)
  float nsw;
  unsigned long long seed;
  int solo;
  const SynthDev* dv;
  const unsigned long long* item_seeds;
};

#define PS_G __attribute__((address_space(1)))
__device__ __forceinline__ ll_t ll_pack(float v, unsigned e) { return ((ll_t)e << 32) | (ll_t)__float_as_uint(v); }
__device__ __forceinline__ void ll_store(PS_G ll_t* p, float v, unsigned e) {
  __hip_atomic_store(p, ll_pack(v, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 ... sc1
}
__device__ __forceinline__ ll_t ll_load(const PS_G ll_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 ... sc1
}
__device__ __forceinline__ int ps_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Pointers of the step program come out of LDS as generic ("flat") per-lane values: make them what they are -- wave-uniform
// GLOBAL pointers -- so that loads become global_load v, v_off, s[base] instead of flat_load on per-lane 64-bit addresses.
template <typename T>
__device__ __forceinline__ PS_G T* ps_unip(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (PS_G T*)(((unsigned long long)hi << 32) | lo);
}

// time-only experiment switches for tools/ab_build.sh (results are garbage): -DPS_EXP_NOGELU, -DPS_EXP_NOPOLL
#ifdef PS_EXP_NOGELU
#define PS_GELU(v) (v)
#else
#define PS_GELU(v) c16_gelu(v)
#endif
struct PsCtx {
  unsigned epoch;
  int aborted;        // this wave gave up (or saw ctl->abort): polls return at once
  PersistCtl* ctl;
};

// One poll round trip for N cells of tensor `a` (+ N of `b` when b != null, + one of `c` when c != null): all loads are issued before the
// first epoch is looked at; the wave repeats the batch until every lane that `need`s its cells has seen this forward's epoch.
// Cell i sits at off0 + i * stride, i < n_live (wave-uniform).  Must be called from wave-uniform control flow.
template <int N>
__device__ __forceinline__ void ps_gather(const PS_G ll_t* a, const PS_G ll_t* b, const PS_G ll_t* c, int off0, int stride, int n_live, int coff, bool need,
                                          PsCtx& cx, float (&va)[N], float (&vb)[N], float& vc) {
  int spins = 0;
  // addresses = (uniform base + uniform i * stride) + ONE per-lane 32-bit byte offset: global_load_dwordx2 v, v_off, s[base] sc1
  unsigned vo = (unsigned)off0 * 8u, vc_off = (unsigned)coff * 8u;
  for (;;) {
    // (opaque per iteration: otherwise the 64-bit address of every load is hoisted out of the loop into a VGPR pair -- 66 pairs --
    //  and the kernel spills; inside the loop body the backend folds base + offset into the load's saddr / voffset operands)
    asm volatile("" : "+v"(vo), "+v"(vc_off));
    ll_t qa[N], qb[N], qc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
      qa[i] = ll_load((const PS_G ll_t*)((const PS_G char*)(a + (size_t)(i < n_live ? i : n_live - 1) * stride) + vo));
    if (b) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        qb[i] = ll_load((const PS_G ll_t*)((const PS_G char*)(b + (size_t)(i < n_live ? i : n_live - 1) * stride) + vo));
    }
    if (c) qc = ll_load((const PS_G ll_t*)((const PS_G char*)c + vc_off));
    unsigned bad = 0;  // (bitwise, not &&: one straight-line block instead of a branch per cell)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      bad |= (unsigned)(qa[i] >> 32) ^ cx.epoch;
      va[i] = __uint_as_float((unsigned)qa[i]);
    }
    if (b) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        bad |= (unsigned)(qb[i] >> 32) ^ cx.epoch;
        vb[i] = __uint_as_float((unsigned)qb[i]);
      }
    }
    if (c) {
      bad |= (unsigned)(qc >> 32) ^ cx.epoch;
      vc = __uint_as_float((unsigned)qc);
    }
#ifdef PS_EXP_NOPOLL
    break;
#endif
    if (__builtin_amdgcn_ballot_w64(need && bad != 0) == 0 || cx.aborted) break;
    ++spins;
    if ((spins & 1023) == 0 && __hip_atomic_load(&cx.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { cx.aborted = 1; break; }
    if (spins >= PS_SPIN_LIMIT) {
      cx.aborted = 1;
      if ((threadIdx.x & 63) == 0) {
        __hip_atomic_store(&cx.ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicAdd(&cx.ctl->timeouts, 1u);
      }
      break;
    }
  }
}

// this wave's weight fragments of 16-row block mb: tap units u = wave + PS_WAVES * i (K = 1: unit = 16-channel chunk)
__device__ __forceinline__ void ps_load_weights(const PS_G float* w16, int mb, int n_u, int wave, int lane, f32x4 (&a)[PS_MAXU]) {
  const PS_G f32x4* wp = (const PS_G f32x4*)w16 + (size_t)mb * n_u * 64 + lane;
#pragma unroll
  for (int i = 0; i < PS_MAXU; ++i) {
    const int u = wave + PS_WAVES * i;
    a[i] = wp[(size_t)(u < n_u ? u : n_u - 1) * 64];
  }
}

// per-thread channel parameters of the window prologue (thread = channel, tid < C): requested with the weights, one step ahead
struct PsPar { float v[8]; };
__device__ __forceinline__ void ps_load_par(const SdpStep& st, int tid, PsPar& p) {
#pragma unroll
  for (int k = 0; k < 8; ++k) p.v[k] = 0.f;
  const int kind = ps_uni(st.kind), C = ps_uni(st.Cin);
  if (kind == PS_PRE || tid >= C) return;
  if (st.y2) { p.v[0] = ps_unip(st.g2)[tid]; p.v[1] = ps_unip(st.b2)[tid]; }
  else if (st.pw) { p.v[0] = ps_unip(st.pw)[tid]; p.v[1] = ps_unip(st.pb)[tid]; }
  if (kind == PS_DDS) {
    const PS_G float* sw = ps_unip(st.sw);
    p.v[2] = ps_unip(st.sb)[tid]; p.v[3] = sw[tid * 3]; p.v[4] = sw[tid * 3 + 1]; p.v[5] = sw[tid * 3 + 2];
    p.v[6] = ps_unip(st.g1)[tid]; p.v[7] = ps_unip(st.b1)[tid];
  }
}

// epilogue operand of thread tid < 256 (row tid >> 4 of 16-row block mb): bias (+ the per-item conditioning row of dp.pre)
__device__ __forceinline__ float ps_load_bias(const SdpStep& st, int mb, int tid) {
  const int Cout = ps_uni(st.Cout);
  const int r = mb * 16 + ((tid >> 4) & 15), rc = r < Cout ? r : Cout - 1;
  float v = ps_unip(st.bias)[rc];
  if (ps_uni(st.kind) == PS_PRE && st.cond) v += ps_unip(st.cond)[rc];
  return v;
}

__global__ void __launch_bounds__(PS_THREADS) sdp_persist_kernel(const SdpProgram* __restrict__ prog, const SdpCall call) {
  extern __shared__ float lds[];
  __shared__ SdpProgram sp;
  __shared__ unsigned s_epoch;
  const int tid0 = threadIdx.x;
  const int wave = ps_uni(tid0 >> 6);
  const int rank = blockIdx.x;
  {
    const int tid = tid0;
    const int* src = reinterpret_cast<const int*>(prog);
    int* dst = reinterpret_cast<int*>(&sp);
    for (int i = tid; i < (int)(sizeof(SdpProgram) / 4); i += PS_THREADS) dst[i] = src[i];
    if (tid == 0) {
      unsigned e = __hip_atomic_load(&call.ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      s_epoch = e ? e : 1u;  // 0 marks "never written"
    }
  }
  __syncthreads();
  PsCtx cx;
  cx.epoch = s_epoch; cx.aborted = 0; cx.ctl = call.ctl;
  const unsigned epoch = cx.epoch;
  const int n_steps = ps_uni(sp.n_steps), T = ps_uni(sp.T), Tp = ps_uni(sp.Tp), ntn = ps_uni(sp.ntn);
  float* tile = lds;                       // [PS_MAXC][16]   B operand of the MFMA tile
  float* xs = tile + PS_MAXC * 16;         // [PS_MAXC][PS_XP] finished layer input over the window; later the cross-wave reduction buffer
  float* red = xs + PS_MAXC * PS_XP;       // 512 floats: block reductions
  float* par = red + 512;                  // [8][PS_MAXC]
  int len_raw;
  {
    const int tid = tid0;
    (void)tid;
    int zero = 0;
    asm volatile("" : "+v"(zero));
    len_raw = ps_unip(sp.len)[zero];  // vector load (stays off the scalar counter), first used in step 0's prologue
  }
  f32x4 a[PS_MAXU];
  PsPar pp;
  float eb = 0.f;
  bool prefetched = false;
  const float ea_m = ps_unip(sp.ea_m)[0], ea_is = expf(-ps_unip(sp.ea_logs)[0]);  // requested now, used by the very last epilogue

  for (int s = 0; s < n_steps; ++s) {
    // (opaque per step: every per-thread index below derives from this copy, so that the compiler does not hoist the address
    //  arithmetic of ALL phases out of the step loop -- it did, and spilled 1400 registers)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const SdpStep& st = sp.steps[s];
    const int kind = ps_uni(st.kind), Cin = ps_uni(st.Cin), Cout = ps_uni(st.Cout), n_mb = ps_uni(st.n_mb);
    const int G = ps_uni(st.G), mbg = ps_uni(st.mbg);
    if (rank >= ntn * G) { prefetched = false; continue; }  // idle in this step: nothing to wait for
    const int j = rank % ntn, g = rank / ntn;
    const int n0 = j * 16;
    const int n_u = Cin >> 4;
    const int mb0 = g * mbg;
    if (!prefetched) {
      ps_load_par(st, tid, pp);
      eb = ps_load_bias(st, mb0, tid);
      ps_load_weights(ps_unip(st.w16), mb0, n_u, wave, lane, a);
    }
    __syncthreads();  // the previous step's readers of the LDS buffers are done
    const int L = len_raw < T ? len_raw : T;

    // ------------------------------------------------------------------ 1. B operand of the tile: [Cin][16]
    if (kind == PS_PRE) {
      // x = text-encoder output (already masked), plain floats written by the previous kernel
      const float* xg = call.x;
      const int col = tid & 15, r0 = tid >> 4;
      const int t = n0 + col, tc = t < T ? t : T - 1;
      float xv[PS_MAXC / 32];
#pragma unroll
      for (int i = 0; i < PS_MAXC / 32; ++i) {
        const int c = r0 + 32 * i;
        xv[i] = xg[(long long)(c < Cin ? c : Cin - 1) * T + tc];
      }
#pragma unroll
      for (int i = 0; i < PS_MAXC / 32; ++i) {
        const int c = r0 + 32 * i;
        if (c < Cin) tile[c * 16 + col] = t < T ? xv[i] : 0.f;
      }
    } else {
      if (tid < Cin) {
#pragma unroll
        for (int k = 0; k < 8; ++k) par[k * PS_MAXC + tid] = pp.v[k];
      }
      __syncthreads();
      const int D = Cin, nci = D >> 4;
      const float invD = 1.0f / (float)D;
      const bool dw = kind == PS_DDS;
      const int dil = dw ? ps_uni(st.dil) : 0;
      const int Wc = 16 + 2 * dil;
      const PS_G ll_t* xin = ps_unip(st.xin);
      const PS_G ll_t* y2 = ps_unip(st.y2);
      const PS_G ll_t* zc = (!y2 && st.pw) ? ps_unip(st.z) + (long long)ps_uni(st.z_row) * Tp : nullptr;
      PS_G ll_t* xout = ps_unip(st.xout);
      // ---- phase A: x_in = (x + gelu(LN2(y2))) * mask over the window columns t = n0 - dil + j'
      {
        for (int jb = 0; jb < Wc; jb += 32) {
          int tq = tid;
          asm volatile("" : "+v"(tq));  // (per pass: keeps the 2 x 16 per-channel LDS parameters of a pass out of loop-invariant registers)
          const int jl = tq & 31, cg = tq >> 5;
          const int jj = jb + jl;
          const bool jok = jj < Wc;
          const int t = n0 - dil + jj;
          const bool tin = jok && t >= 0 && t < L;
          const int tc = t < 0 ? 0 : (t >= Tp ? Tp - 1 : t);
          float xv[PS_MAXI], yv[PS_MAXI], zv = 0.f;
          ps_gather<PS_MAXI>(xin, y2, zc, cg * Tp + tc, 16 * Tp, nci, tc, tin, cx, xv, yv, zv);
          if (zc) {
#pragma unroll
            for (int i = 0; i < PS_MAXI; ++i) {
              const int c = cg + 16 * i, cc = c < D ? c : D - 1;
              xv[i] = par[cc] * zv + par[PS_MAXC + cc] + xv[i];
            }
          }
          if (y2) {
            float m = 0.f;
#pragma unroll
            for (int i = 0; i < PS_MAXI; ++i) m += i < nci ? yv[i] : 0.f;
            m = c16_groupsum<16, 32>(m, red, cg, jl) * invD;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < PS_MAXI; ++i) { const float d = yv[i] - m; q += i < nci ? d * d : 0.f; }
            q = c16_groupsum<16, 32>(q, red, cg, jl);
            const float rstd = 1.0f / sqrtf(q * invD + 1e-5f);
#pragma unroll
            for (int i = 0; i < PS_MAXI; ++i) {
              const int c = cg + 16 * i, cc = c < D ? c : D - 1;
              xv[i] += PS_GELU((yv[i] - m) * rstd * par[cc] + par[PS_MAXC + cc]);
            }
          }
          const bool own = xout && jok && t >= n0 && t < n0 + 16;
          const bool mine = (cg % G) == g;  // channel groups of the tile's own columns are spread over its G workers
#pragma unroll
          for (int i = 0; i < PS_MAXI; ++i) {
            const int c = cg + 16 * i;
            const float v = tin ? xv[i] : 0.f;  // x = (x + y) * mask; select: columns that were not polled hold garbage
            if (i < nci && jok) xs[c * PS_XP + jj] = v;
            if (i < nci && own && mine) ll_store(xout + (long long)c * Tp + t, v, epoch);  // the tile's workers share the write
          }
        }
      }
      __syncthreads();
      // ---- phase B: depthwise conv, LN1, GELU -> B tile (PS_DDS), or the finished input itself (proj layers)
      const int jc = tid & 15, cg = tid >> 4, ncj = D >> 5;
      if (!dw) {
#pragma unroll
        for (int i = 0; i < PS_MAXI / 2; ++i) {
          const int c = cg + 32 * i;
          if (i < ncj) tile[c * 16 + jc] = xs[c * PS_XP + jc];
        }
      } else {
        float y1[PS_MAXI / 2];
        float m1 = 0.f;
#pragma unroll
        for (int i = 0; i < PS_MAXI / 2; ++i) {
          const int c = cg + 32 * i, cc = c < D ? c : D - 1;
          float acc = par[2 * PS_MAXC + cc];
#pragma unroll
          for (int k = 0; k < 3; ++k) acc += par[(3 + k) * PS_MAXC + cc] * xs[cc * PS_XP + jc + k * dil];
          y1[i] = acc;
          m1 += i < ncj ? acc : 0.f;
        }
        m1 = c16_groupsum<32, 16>(m1, red, cg, jc) * invD;
        float v1 = 0.f;
#pragma unroll
        for (int i = 0; i < PS_MAXI / 2; ++i) { const float d = y1[i] - m1; v1 += i < ncj ? d * d : 0.f; }
        v1 = c16_groupsum<32, 16>(v1, red, cg, jc);
        const float rstd1 = 1.0f / sqrtf(v1 * invD + 1e-5f);
#pragma unroll
        for (int i = 0; i < PS_MAXI / 2; ++i) {
          const int c = cg + 32 * i, cc = c < D ? c : D - 1;
          if (i < ncj) tile[c * 16 + jc] = PS_GELU((y1[i] - m1) * rstd1 * par[6 * PS_MAXC + cc] + par[7 * PS_MAXC + cc]);
        }
      }
    }
    __syncthreads();

    // ------------------------------------------------------------------ 2. MFMA tiles of this worker's 16-row blocks + epilogues
    float* mred = xs;               // [PS_WAVES][4][64] partial tiles (the window is dead)
    float* hb = xs + PS_WAVES * 256;  // PS_CFPROJ: h [32][16]
    const float* bl = tile + (lane >> 4) * 16 + (lane & 15);
    for (int mi = 0; mi < mbg; ++mi) {
      const int mb = mb0 + mi;
      if (mb >= n_mb) break;
      if (mi > 0) {
        __syncthreads();  // mred of the previous block has been read
        eb = ps_load_bias(st, mb, tid);
        ps_load_weights(ps_unip(st.w16), mb, n_u, wave, lane, a);
      }
      const float ebias = eb;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < PS_MAXU; ++i) {
        const int u = wave + PS_WAVES * i;
        if (u < n_u) {  // wave-uniform
          const float* bp = bl + u * (16 * 16);
          const float b0 = bp[0], b1 = bp[4 * 16], b2 = bp[8 * 16], b3 = bp[12 * 16];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0], b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][1], b1, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][2], b2, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][3], b3, acc1, 0, 0, 0);
        }
      }
      // the weight registers are free: request the NEXT step's parameters and first weight block now, so that they fly under this
      // step's reduction, epilogue and the exchange (in-order vmcnt: older than every poll of the next step)
      if (mi == mbg - 1 || mb == n_mb - 1) {
        prefetched = false;
        if (s + 1 < n_steps) {
          const SdpStep& nx = sp.steps[s + 1];
          const int nG = ps_uni(nx.G);
          if (rank < ntn * nG) {
            const int nmb0 = (rank / ntn) * ps_uni(nx.mbg), nn_mb = ps_uni(nx.n_mb);
            ps_load_par(nx, tid, pp);
            eb = ps_load_bias(nx, nmb0 < nn_mb ? nmb0 : nn_mb - 1, tid);
            ps_load_weights(ps_unip(nx.w16), nmb0 < nn_mb ? nmb0 : nn_mb - 1, ps_uni(nx.Cin) >> 4, wave, lane, a);
            prefetched = true;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) mred[(wave * 4 + r) * 64 + lane] = acc0[r] + acc1[r];
      __syncthreads();
      if (tid < 256) {
        const int row = tid >> 4, col = tid & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PS_WAVES; ++w) v += mred[(w * 4 + (row & 3)) * 64 + (row >> 2) * 16 + col];
        const int r = mb * 16 + row;
        const int t = n0 + col;
        if (r < Cout) {
          v += ebias;
          if (kind == PS_PROJ && t >= L) v = 0.f;  // proj(x) * x_mask (models.py:63)
          if (kind == PS_CFPROJ) hb[r * 16 + col] = v;
          else ll_store(ps_unip(st.yout) + (long long)r * Tp + t, v, epoch);
        }
      }
    }
    if (kind == PS_PRE && g == 0 && tid < 32) {
      // z = randn * noise_scale_w (models.py:96): injected noise or the Philox stream of dp_init_z_kernel
      const int c = tid >> 4, t = n0 + (tid & 15);
      float nsw = call.nsw;
      unsigned long long seed = call.seed;
      if (call.dv) { nsw = call.dv->scales[2]; seed = call.dv->seed; }
      float e = 0.f;
      if (t < T) e = call.noise ? call.noise[(long long)c * T + t]
                                : (call.solo ? philox_normal(call.item_seeds ? call.item_seeds[0] : seed, 1, (uint32_t)c, (uint32_t)t)
                                             : philox_normal(seed, 1, (uint32_t)c, (uint32_t)t));
      ll_store(ps_unip(st.zout) + (long long)c * Tp + t, e * nsw, epoch);
    }
    if (kind == PS_CFPROJ) {
      __syncthreads();  // h complete
      if (wave == 0) {  // one column per lane (lanes >= 16 idle but inside the wave-uniform poll)
        const int col = lane & 15, t = n0 + col;
        const int x0r = ps_uni(st.z_row), x1r = 1 - x0r;
        const PS_G ll_t* zin = ps_unip(st.z);
        float z0[1], z1[1], dummy = 0.f;
        const bool need = lane < 16 && t < L;
        ps_gather<1>(zin + (long long)x0r * Tp, zin + (long long)x1r * Tp, nullptr, t, 0, 1, 0, need, cx, z0, z1, dummy);
        if (lane < 16) {
          float v0 = 0.f, v1 = 0.f;
          if (t < L) {
            v0 = z0[0];
            v1 = spline_inverse_elem(z1[0], [&](int i) { return hb[i * 16 + col]; }, ps_uni(sp.nb), sp.bound, sp.inv_sqrt_d);
          }
          if (st.zout) {
            ll_store(ps_unip(st.zout) + (long long)x0r * Tp + t, v0, epoch);
            ll_store(ps_unip(st.zout) + (long long)x1r * Tp + t, v1, epoch);
          }
          if (ps_uni(st.last) && t < T) {
            const float zz = ps_uni(st.ea_row) == x0r ? v0 : v1;
            ps_unip(sp.logw)[t] = t < L ? (zz - ea_m) * ea_is : 0.f;
          }
        }
      }
    }
  }
  // ---- the last worker to finish publishes the epoch (every worker read it before doing anything else)
  __syncthreads();
  if (tid0 == 0) {
    if (cx.aborted || __hip_atomic_load(&call.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr((int*)sp.err, PS_ERR_TIMEOUT);
    const unsigned old = atomicAdd(&call.ctl->done, 1u);
    if (old == gridDim.x - 1) {
      __hip_atomic_store(&call.ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&call.ctl->abort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&call.ctl->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#define PS_LDS_BYTES ((PS_MAXC * (16 + PS_XP + 8) + 512) * sizeof(float))
