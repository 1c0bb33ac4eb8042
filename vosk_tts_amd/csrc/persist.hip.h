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
// XCDs, coherent chip-wide with sc1 stores).  A worker that is idle in a step simply moves on; a consumer waits only for the cells
// it reads.  Epochs make stale data harmless: every forward uses epoch = (last completed forward) + 1, cells are never reset, every
// step of a forward writes its OWN buffers (no reuse inside a forward, so there is no write-after-read hazard either), and a cell
// whose epoch does not match is simply not there yet.  Every poll loop is bounded (PS_SPIN_LIMIT): a lost worker turns into an
// error word, never into a hung GPU.
//
// Decomposition (v2; v1 ran a whole DDSConv layer per step like conv16_kernel's PRO == 1 and was bound by what that costs per
// workgroup: every one of the 16 workgroups of a column tile pulled the full 256-channel x 34-column window of two tensors through
// L1-bypassing loads -- 131-262 KB per step at the ~30 GB/s one CU sustains on such loads -- and redid its LayerNorms and 8 k erf
// evaluations: 12 us per step, measured 208 us for the predictor).  A layer is now TWO kinds of steps over column-major cells
// [T][C] (a column's channels are contiguous: 2 KB):
//   * column step (PS_COL), one worker per COLUMN t: finish the previous layer for the three columns the depthwise taps touch
//     (x + gelu(LN2(y2)) at t - d, t, t + d: 6 column vectors = 12 KB gathered), depthwise conv, LN1, GELU -> the layer's 1x1
//     input column and the finished residual column.  Thread = channel; a LayerNorm is a DPP wave reduction + one LDS exchange.
//     Every element of the layer is computed once on the chip (3x for the finish), not once per row block.
//   * matrix step (PS_MM), one worker per (16-column tile, 16-row block): gather the [C_in x 16] operand (32 KB, 4 KB per wave
//     load instruction batch), 16x16x4 fp32 MFMAs with the 8 waves splitting the contraction, weights prefetched into registers
//     during the PREVIOUS step, bias / mask epilogue, cells out.  ConvFlow.proj workers own all 29 rows of their columns and run
//     the spline inverse in their epilogue; the last one folds the final ElementwiseAffine and writes logw.
#pragma once
#include "conv_small.hip.h"
#include "kernels_misc.hip.h"

typedef unsigned long long ll_t;  // {float value (bits 0..31), u32 epoch (bits 32..63)}

#define PS_THREADS 512
#define PS_WAVES 8
#define PS_MAX_STEPS 40
#define PS_MAXC 256      // channels of an exchanged tensor / contraction length
#define PS_MAXU 2        // tap units (16 channels) per wave: PS_MAXC / 16 / PS_WAVES
#define PS_TP 17         // LDS pitch of the MFMA operand tile [C_in][16] (odd: the transposing writes are conflict-free)
#define PS_SPIN_LIMIT (1 << 18)
#define PS_ERR_TIMEOUT 8  // bit in the session error word

struct PersistCtl {
  unsigned epoch;     // epoch of the last completed forward
  unsigned done;      // workers that finished the current forward (the last one publishes the epoch and resets this)
  unsigned abort;     // a worker timed out: everybody stops polling
  unsigned timeouts;  // diagnostics
};

enum { PS_PRE = 0, PS_COL = 1, PS_MM = 2 };
enum { PS_EPI_RAW = 0, PS_EPI_MASK = 1, PS_EPI_SPLINE = 2 };

struct SdpStep {
  int kind;
  // ---- PS_PRE / PS_MM: y[Cout x 16-column tile] = W[Cout x Cin] * B + bias
  int Cin, Cout, n_mb;     // contraction channels, rows stored, 16-row blocks of the packed matrix
  int G, mbg;              // workers per column tile, 16-row blocks per worker
  int epi;                 // PS_EPI_*
  int ypitch;              // channel pitch of yout
  const float* w16;        // [n_mb][Cin/16][64][4] 16x16x4 A-fragment order (pack_conv_weights16)
  const float* bias;
  const float* cond;       // PS_PRE: per-item bias rows (cond(g), models.py:60) or null
  const ll_t* bin;         // PS_MM: operand cells [Tp][Cin]
  ll_t* yout;              // [Tp][ypitch]
  // PS_EPI_SPLINE (ConvFlow.proj + spline inverse) and flow layer 0 of PS_COL
  const ll_t* z;           // z cells [2][Tp]
  ll_t* zout;              // PS_PRE: z = noise * noise_scale_w; PS_EPI_SPLINE: transformed z (null for the last flow)
  int z_row;               // row of z that conditions (x0); the spline acts on 1 - z_row
  int last, ea_row;        // last flow: write logw = ElementwiseAffine^-1(z[ea_row]) (modules.py:293-295)
  // ---- PS_COL: column t of   x_in = (xin + gelu(LN(y2; g2, b2))) * mask      [y2 == null: xin * mask; pw != null: pw * z + pb + xin]
  //                            b    = gelu(LN(depthwise3(x_in; sw, sb, dil); g1, b1))          (dw != 0)
  // (every parameter pointer of EVERY step is valid -- unused ones point at a block of zeros -- so that the one-step-ahead
  //  prefetch is straight-line code: a load behind a branch makes hipcc wait for it at the join, i.e. puts a cold round trip
  //  in the middle of a step; measured 4-5 k cycles per matrix step)
  int D, dil, dw;
  int fin;                 // 0: x_in = xin; 1: xin + gelu(LN(y2)); 2: pw * z + pb + xin
  const ll_t* xin; const ll_t* y2;   // [Tp][D]
  const float* g2; const float* b2;
  const float* sw; const float* sb; const float* g1; const float* b1;
  const float* pw; const float* pb;  // flow layer 0: ConvFlow.pre (Conv1d(1, D, 1)), modules.py:365
  ll_t* xout;              // x_in column [Tp][D] (next layer's residual stream; the proj layers' MFMA operand)
  ll_t* bout;              // b column [Tp][D] (dw != 0)
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
  long long* trace;                     // tools only (VITS_PS_TRACE): [P][PS_MAX_STEPS][4] cycle stamps, null in production
};

// Pointers of the step program come out of LDS as generic ("flat") per-lane values: make them what they are -- wave-uniform
// GLOBAL pointers -- so that loads become global_load instead of flat_load on per-lane 64-bit addresses.
#define PS_G __attribute__((address_space(1)))
__device__ __forceinline__ ll_t ll_pack(float v, unsigned e) { return ((ll_t)e << 32) | (ll_t)__float_as_uint(v); }
__device__ __forceinline__ void ll_store(PS_G ll_t* p, float v, unsigned e) {
  __hip_atomic_store(p, ll_pack(v, e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 ... sc1
}
__device__ __forceinline__ ll_t ll_load(const PS_G ll_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_load_dwordx2 ... sc1
}
__device__ __forceinline__ float ll_val(ll_t q) { return __uint_as_float((unsigned)q); }
__device__ __forceinline__ int ps_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
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

#define PS_STAMP(k) do { if (call.trace && tid0 == 0) call.trace[((long long)rank * PS_MAX_STEPS + s) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
struct PsCtx {
  unsigned epoch;
  int aborted;        // this wave gave up (or saw ctl->abort): polls return at once
  int spins;
  PersistCtl* ctl;
};
// one more round of a poll loop: true = keep polling.  `pending` is wave-uniform (a ballot).
__device__ __forceinline__ bool ps_again(PsCtx& cx, bool pending) {
#ifdef PS_EXP_NOPOLL
  return false;
#endif
  if (!pending || cx.aborted) { cx.spins = 0; return false; }
  ++cx.spins;
  if ((cx.spins & 1023) == 0 && __hip_atomic_load(&cx.ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { cx.aborted = 1; return false; }
  if (cx.spins >= PS_SPIN_LIMIT) {
    cx.aborted = 1;
    if ((threadIdx.x & 63) == 0) {
      __hip_atomic_store(&cx.ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicAdd(&cx.ctl->timeouts, 1u);
    }
    return false;
  }
  return true;
}

// sum over the 64 lanes of a wave, returned in every lane (DPP row operations + two row broadcasts: no LDS, ~10 instructions).
// Lanes disabled by ROW_MASK contribute old = 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float ps_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float ps_wave_sum(float v) {
#ifdef PS_SHFL_REDUCE
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  v = ps_dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = ps_dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = ps_dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = ps_dpp_add<0x140, 0xF>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
  v = ps_dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v = ps_dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
// sums of two values over the 256 threads of each HALF of the workgroup (waves 0-3 / 4-7): one barrier; `red` is a private
// 16-float scratch of this call site (no second barrier: the next call site uses another one)
__device__ __forceinline__ void ps_half_sum2(float& a, float& b, float* red, int wave, int lane) {
  a = ps_wave_sum(a);
  b = ps_wave_sum(b);
  if (lane == 0) { red[wave] = a; red[8 + wave] = b; }
  __syncthreads();
  const int w0 = wave & 4;
  a = (red[w0] + red[w0 + 1]) + (red[w0 + 2] + red[w0 + 3]);
  b = (red[8 + w0] + red[8 + w0 + 1]) + (red[8 + w0 + 2] + red[8 + w0 + 3]);
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

// what a worker requests one step ahead (registers): the per-channel parameters of a column step (thread = channel c = tid & 255),
// or the weights + epilogue operand of a matrix step
struct PsPar { float v[8]; };  // g2 | pw, b2 | pb, sb, sw0, sw1, sw2, g1, b1
__device__ __forceinline__ void ps_load_par(const SdpStep& st, int tid, PsPar& p) {
  const int D = ps_uni(st.D);
  int c = tid & 255;
  c = c < D ? c : D - 1;
  const PS_G float* sw = ps_unip(st.sw);
  p.v[0] = ps_unip(st.g2)[c]; p.v[1] = ps_unip(st.b2)[c];
  p.v[2] = ps_unip(st.sb)[c]; p.v[3] = sw[c * 3]; p.v[4] = sw[c * 3 + 1]; p.v[5] = sw[c * 3 + 2];
  p.v[6] = ps_unip(st.g1)[c]; p.v[7] = ps_unip(st.b1)[c];
}
// epilogue operands of thread tid < 256 (row tid & 15 of 16-row block mb): bias and the per-item conditioning row (dp.pre; zeros elsewhere)
__device__ __forceinline__ void ps_load_bias(const SdpStep& st, int mb, int tid, float& eb, float& ec) {
  const int Cout = ps_uni(st.Cout);
  const int r = mb * 16 + (tid & 15), rc = r < Cout ? r : Cout - 1;
  eb = ps_unip(st.bias)[rc];
  ec = ps_unip(st.cond)[rc];
}

__global__ void __launch_bounds__(PS_THREADS) sdp_persist_kernel(const SdpProgram* __restrict__ prog, const SdpCall call) {
  __shared__ SdpProgram sp;
  __shared__ unsigned s_epoch;
  __shared__ float tile[PS_MAXC * PS_TP];   // MFMA operand [C_in][16] (pitch 17)
  __shared__ float mred[PS_WAVES * 256];    // partial tiles of the 8 waves
  __shared__ float hb[32 * 16];             // ConvFlow.proj output of the tile (spline parameters)
  __shared__ float xs[3 * PS_MAXC];         // column step: x_in at t - d, t, t + d
  __shared__ float red[4 * 16];             // block reductions (one 16-float scratch per call site)
  const int tid0 = threadIdx.x;
  const int wave = ps_uni(tid0 >> 6);
  const int rank = blockIdx.x, P = gridDim.x;
  {
    const int* src = reinterpret_cast<const int*>(prog);
    int* dst = reinterpret_cast<int*>(&sp);
    for (int i = tid0; i < (int)(sizeof(SdpProgram) / 4); i += PS_THREADS) dst[i] = src[i];
    if (tid0 == 0) {
      unsigned e = __hip_atomic_load(&call.ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      s_epoch = e ? e : 1u;  // 0 marks "never written"
    }
  }
  __syncthreads();
  PsCtx cx;
  cx.epoch = s_epoch; cx.aborted = 0; cx.spins = 0; cx.ctl = call.ctl;
  const unsigned epoch = cx.epoch;
  const int n_steps = ps_uni(sp.n_steps), T = ps_uni(sp.T), Tp = ps_uni(sp.Tp), ntn = ps_uni(sp.ntn);
  int len_raw;
  {
    int zero = 0;
    asm volatile("" : "+v"(zero));
    len_raw = ps_unip(sp.len)[zero];  // vector load (stays off the scalar counter), first used in step 0
  }
  const float ea_m = ps_unip(sp.ea_m)[0], ea_is = expf(-ps_unip(sp.ea_logs)[0]);  // requested now, used by the very last epilogue
  f32x4 a[PS_MAXU];
  PsPar pp;
  float eb = 0.f, ec = 0.f;
  bool prefetched = false;
  const int wj = rank % ntn, wg = rank / ntn;  // matrix steps: this worker's column tile and row-block group

  // is this worker busy in step `st`?  column steps: column `rank` (and rank + P, ...); matrix steps: item `rank`
  auto busy = [&](const SdpStep& st) -> bool {
    return ps_uni(st.kind) == PS_COL ? rank < Tp : rank < ntn * ps_uni(st.G);
  };
  // everything a step needs from read-only memory, requested one step ahead (straight-line: see SdpStep)
  auto prefetch = [&](const SdpStep& st, int tid, int lane) {
    ps_load_par(st, tid, pp);
    const int n_mb = ps_uni(st.n_mb);
    int mb0 = wg * ps_uni(st.mbg);
    mb0 = mb0 < n_mb ? mb0 : n_mb - 1;
    ps_load_bias(st, mb0, tid, eb, ec);
    ps_load_weights(ps_unip(st.w16), mb0, ps_uni(st.Cin) >> 4, wave, lane, a);
  };

  for (int s = 0; s < n_steps; ++s) {
    // (opaque per step: every per-thread index below derives from this copy, so that the compiler does not hoist the address
    //  arithmetic of ALL phases out of the step loop and spill it)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const SdpStep& st = sp.steps[s];
    const int kind = ps_uni(st.kind);
    if (!busy(st)) { prefetched = false; continue; }  // idle in this step: nothing to wait for
    PS_STAMP(0);
    if (!prefetched) prefetch(st, tid, lane);
    prefetched = false;
    __syncthreads();  // the previous step's readers of the LDS buffers are done
    const int L = len_raw < T ? len_raw : T;

    if (kind == PS_COL) {
      // ================================================================== column step
      const int D = ps_uni(st.D), dw = ps_uni(st.dw), d = ps_uni(st.dil);
      const float invD = 1.0f / (float)D;
      const PS_G ll_t* xin = ps_unip(st.xin);
      const int fin = ps_uni(st.fin);
      const PS_G ll_t* y2 = fin == 1 ? ps_unip(st.y2) : nullptr;
      const PS_G ll_t* zc = fin == 2 ? ps_unip(st.z) + (long long)ps_uni(st.z_row) * Tp : nullptr;
      PS_G ll_t* xout = ps_unip(st.xout);
      PS_G ll_t* bout = ps_unip(st.bout);
      const PsPar par = pp;  // this step's parameters; pp is re-requested for the next step below
      // the next step's operands fly under this step (in-order vmcnt: they are older than every poll of the next step)
      prefetch(sp.steps[s + 1 < n_steps ? s + 1 : s], tid, lane);
      prefetched = true;
      const int c = tid & 255, h = tid >> 8;
      const bool cok = c < D;
      for (int t = rank; t < Tp; t += P) {
        if (t != rank) __syncthreads();
        if (t >= L) {  // padding column (worker-uniform): zeros, nothing to wait for
          if (h == 1 && cok) {
            ll_store(xout + (long long)t * D + c, 0.f, epoch);
            if (bout) ll_store(bout + (long long)t * D + c, 0.f, epoch);
          }
          continue;
        }
        // slots of this thread: half 1 -> column t; half 0 -> columns t - d and t + d (depthwise steps only)
        const int t0 = h ? t : t - d, t1 = t + d;
        const bool n0 = cok && (h ? true : (dw && t0 >= 0)), n1 = cok && !h && dw && t1 < L;
        const int tc0 = t0 < 0 ? 0 : t0, tc1 = t1 < Tp ? t1 : Tp - 1;
        float x0 = 0.f, y0 = 0.f, z0 = 0.f, x1 = 0.f, y1 = 0.f, z1 = 0.f;
        {
          unsigned o0 = (unsigned)(tc0 * D + c) * 8u, o1 = (unsigned)(tc1 * D + c) * 8u, oz0 = (unsigned)tc0 * 8u, oz1 = (unsigned)tc1 * 8u;
          bool pending;
          do {
            asm volatile("" : "+v"(o0), "+v"(o1), "+v"(oz0), "+v"(oz1));  // (addresses stay inside the loop body: no hoisted 64-bit pairs)
            ll_t qx0 = 0, qy0 = 0, qz0 = 0, qx1 = 0, qy1 = 0, qz1 = 0;
            if (n0) {
              qx0 = ll_load((const PS_G ll_t*)((const PS_G char*)xin + o0));
              if (y2) qy0 = ll_load((const PS_G ll_t*)((const PS_G char*)y2 + o0));
              if (zc) qz0 = ll_load((const PS_G ll_t*)((const PS_G char*)zc + oz0));
            }
            if (n1) {
              qx1 = ll_load((const PS_G ll_t*)((const PS_G char*)xin + o1));
              if (y2) qy1 = ll_load((const PS_G ll_t*)((const PS_G char*)y2 + o1));
              if (zc) qz1 = ll_load((const PS_G ll_t*)((const PS_G char*)zc + oz1));
            }
            unsigned bad = 0;
            if (n0) bad |= ((unsigned)(qx0 >> 32) ^ epoch) | (y2 ? (unsigned)(qy0 >> 32) ^ epoch : 0u) | (zc ? (unsigned)(qz0 >> 32) ^ epoch : 0u);
            if (n1) bad |= ((unsigned)(qx1 >> 32) ^ epoch) | (y2 ? (unsigned)(qy1 >> 32) ^ epoch : 0u) | (zc ? (unsigned)(qz1 >> 32) ^ epoch : 0u);
            x0 = ll_val(qx0); y0 = ll_val(qy0); z0 = ll_val(qz0); x1 = ll_val(qx1); y1 = ll_val(qy1); z1 = ll_val(qz1);
            pending = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
          } while (ps_again(cx, pending));
        }
        PS_STAMP(1);
        if (zc) { x0 = par.v[0] * z0 + par.v[1] + x0; x1 = par.v[0] * z1 + par.v[1] + x1; }  // ConvFlow.pre(x0) + g  (modules.py:365-366)
        if (y2) {  // x + gelu(LN2(y2)), two-pass statistics like F.layer_norm; both slots of a half in the same reductions
          float m0 = n0 ? y0 : 0.f, m1 = n1 ? y1 : 0.f;
          ps_half_sum2(m0, m1, red, wave, lane);
          m0 *= invD; m1 *= invD;
          const float e0 = y0 - m0, e1 = y1 - m1;
          float q0 = n0 ? e0 * e0 : 0.f, q1 = n1 ? e1 * e1 : 0.f;
          ps_half_sum2(q0, q1, red + 16, wave, lane);
          x0 += PS_GELU(e0 * (1.0f / sqrtf(q0 * invD + 1e-5f)) * par.v[0] + par.v[1]);
          x1 += PS_GELU(e1 * (1.0f / sqrtf(q1 * invD + 1e-5f)) * par.v[0] + par.v[1]);
        }
        // x = (x + y) * mask: columns outside [0, L) are zero (never polled)
        const float xi0 = (n0 && t0 < L) ? x0 : 0.f, xi1 = n1 ? x1 : 0.f;
        if (cok) {
          if (h) { xs[PS_MAXC + c] = xi0; ll_store(xout + (long long)t * D + c, xi0, epoch); }
          else { xs[c] = xi0; xs[2 * PS_MAXC + c] = xi1; }
        }
        if (dw) {
          __syncthreads();
          // depthwise conv (modules.py:100), LN1, GELU: the 256 threads of half 1 (one erf so far; half 0 had two and runs
          // along for the barriers)
          float v = 0.f;
          if (cok) v = par.v[2] + par.v[3] * xs[c] + par.v[4] * xs[PS_MAXC + c] + par.v[5] * xs[2 * PS_MAXC + c];
          float m = (cok && h) ? v : 0.f, dummy = 0.f;
          ps_half_sum2(m, dummy, red + 32, wave, lane);
          m *= invD;
          const float e = v - m;
          float q = (cok && h) ? e * e : 0.f;
          ps_half_sum2(q, dummy, red + 48, wave, lane);
          if (cok && h) ll_store(bout + (long long)t * D + c, PS_GELU(e * (1.0f / sqrtf(q * invD + 1e-5f)) * par.v[6] + par.v[7]), epoch);
        }
      }
      PS_STAMP(3);
      continue;
    }

    // ==================================================================== matrix step (PS_PRE / PS_MM)
    const int Cin = ps_uni(st.Cin), Cout = ps_uni(st.Cout), n_mb = ps_uni(st.n_mb), mbg = ps_uni(st.mbg), epi = ps_uni(st.epi);
    const int j = wj, g = wg;
    const int n0 = j * 16, n_u = Cin >> 4, mb0 = g * mbg;
    // ---- operand tile [Cin][16] -> LDS (transposed: cells are column-major)
    if (kind == PS_PRE) {
      // x = text-encoder output [H][T] (already masked), plain floats written by the previous kernel
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
        if (c < Cin) tile[c * PS_TP + col] = t < T ? xv[i] : 0.f;
      }
    } else {
      // thread = (channel c = tid & 255, column pair): cells (column 2 k + (tid >> 8), channel c), k < 8 -- no index arithmetic
      // beyond one multiply per cell; a wave-load covers 64 consecutive channels of one column (512 contiguous bytes)
      const PS_G ll_t* bin = ps_unip(st.bin) + (long long)n0 * Cin;  // the tile's 16 columns are one contiguous 16 * Cin block
      const int c = tid & 255, jh = tid >> 8;
      const bool cok = c < Cin;
      const int cc = cok ? c : Cin - 1;
      constexpr int NG = 8;
      unsigned off[NG];
#pragma unroll
      for (int k = 0; k < NG; ++k) off[k] = (unsigned)((2 * k + jh) * Cin + cc) * 8u;
      float v[NG];
      bool pending;
      do {
#pragma unroll
        for (int k = 0; k < NG; ++k) asm volatile("" : "+v"(off[k]));
        ll_t q[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) q[k] = ll_load((const PS_G ll_t*)((const PS_G char*)bin + off[k]));
        unsigned bad = 0;
#pragma unroll
        for (int k = 0; k < NG; ++k) { bad |= (unsigned)(q[k] >> 32) ^ epoch; v[k] = ll_val(q[k]); }
        pending = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
      } while (ps_again(cx, pending));
      PS_STAMP(1);
      if (cok) {
#pragma unroll
        for (int k = 0; k < NG; ++k) tile[c * PS_TP + 2 * k + jh] = v[k];
      }
    }
    __syncthreads();

    PS_STAMP(2);
    // ---- MFMA tiles of this worker's 16-row blocks + epilogues
    const float* bl = tile + (lane >> 4) * PS_TP + (lane & 15);
    for (int mi = 0; mi < mbg; ++mi) {
      const int mb = mb0 + mi;
      if (mb >= n_mb) break;
      if (mi > 0) {
        __syncthreads();  // mred of the previous block has been read
        ps_load_bias(st, mb, tid, eb, ec);
        ps_load_weights(ps_unip(st.w16), mb, n_u, wave, lane, a);
      }
      const float ebias = eb, econd = ec;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < PS_MAXU; ++i) {
        const int u = wave + PS_WAVES * i;
        if (u < n_u) {  // wave-uniform
          const float* bp = bl + u * (16 * PS_TP);
          const float b0 = bp[0], b1 = bp[4 * PS_TP], b2 = bp[8 * PS_TP], b3 = bp[12 * PS_TP];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0], b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][1], b1, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][2], b2, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][3], b3, acc1, 0, 0, 0);
        }
      }
      // the weight registers are free: request the NEXT step's operands now, so that they fly under this step's reduction,
      // epilogue and the exchange
      if (mi == mbg - 1 || mb == n_mb - 1) { prefetch(sp.steps[s + 1 < n_steps ? s + 1 : s], tid, lane); prefetched = true; }
#pragma unroll
      for (int r = 0; r < 4; ++r) mred[(wave * 4 + r) * 64 + lane] = acc0[r] + acc1[r];
      __syncthreads();
      if (tid < 256) {
        const int row = tid & 15, col = tid >> 4;  // rows fastest: a column's 16 cells are one 128-byte segment
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < PS_WAVES; ++w) v += mred[(w * 4 + (row & 3)) * 64 + (row >> 2) * 16 + col];
        const int r = mb * 16 + row;
        const int t = n0 + col;
        if (r < Cout) {
          v += ebias + econd;
          if (epi == PS_EPI_MASK && t >= L) v = 0.f;  // proj(x) * x_mask (models.py:63)
          if (epi == PS_EPI_SPLINE) hb[r * 16 + col] = v;
          else ll_store(ps_unip(st.yout) + (long long)t * ps_uni(st.ypitch) + r, v, epoch);
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
    if (epi == PS_EPI_SPLINE) {
      // Inverse rational-quadratic spline of the tile's 16 columns (transforms.py:55-177), the arithmetic of spline_inverse_elem
      // (kernels_misc.hip.h) spread over the workgroup: one thread per column ran ~23 k cycles (20 expf and 20 divisions in a
      // dependent chain); here the exponentials are one per thread, the two short serial scans (sum, cumulative widths / heights:
      // same order as the serial form) run on 32 threads, and 16 threads finish (bin search, quadratic).
      __syncthreads();  // h complete
      const int nb = ps_uni(sp.nb);
      const float bound = sp.bound, isd = sp.inv_sqrt_d;
      float* se = mred;            // [32][16] exp(w - max): slots 0..nb-1 widths, 16..16+nb-1 heights
      float* sc = mred + 32 * 16;  // [2][17][16] cumulative widths / heights (knots)
      {
        const int col = tid & 15, slot = tid >> 4, which = slot >> 4, i = slot & 15;
        if (i < nb) {
          const float* src = hb + (which * nb) * 16 + col;
          float mx = -3.0e38f;
          for (int k = 0; k < nb; ++k) mx = fmaxf(mx, src[k * 16] * isd);
          se[slot * 16 + col] = expf(src[i * 16] * isd - mx);
        }
      }
      __syncthreads();
      if (tid < 32) {
        const int col = tid & 15, which = tid >> 4;
        const float* e = se + which * 256 + col;
        float* cdst = sc + which * 17 * 16 + col;
        const float mn = 1e-3f;  // min_bin_width == min_bin_height
        float sum = 0.f;
        for (int k = 0; k < nb; ++k) sum += e[k * 16];
        float acc = 0.f;
        cdst[0] = -bound;
        for (int k = 0; k < nb; ++k) {
          acc += mn + (1.f - mn * nb) * (e[k * 16] / sum);
          cdst[(k + 1) * 16] = 2.f * bound * acc - bound;
        }
        cdst[nb * 16] = bound;
      }
      __syncthreads();
      if (wave == 0) {  // one column per lane (lanes >= 16 ride along in the wave-uniform poll)
        const int col = lane & 15, t = n0 + col;
        const int x0r = ps_uni(st.z_row), x1r = 1 - x0r;
        const PS_G ll_t* zin = ps_unip(st.z);
        const bool need = lane < 16 && t < L;
        unsigned o0 = (unsigned)(x0r * Tp + t) * 8u, o1 = (unsigned)(x1r * Tp + t) * 8u;
        float z0 = 0.f, z1 = 0.f;
        bool pending;
        do {
          asm volatile("" : "+v"(o0), "+v"(o1));
          ll_t q0 = 0, q1 = 0;
          if (need) {
            q0 = ll_load((const PS_G ll_t*)((const PS_G char*)zin + o0));
            q1 = ll_load((const PS_G ll_t*)((const PS_G char*)zin + o1));
          }
          const unsigned bad = need ? (((unsigned)(q0 >> 32) ^ epoch) | ((unsigned)(q1 >> 32) ^ epoch)) : 0u;
          z0 = ll_val(q0); z1 = ll_val(q1);
          pending = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
        } while (ps_again(cx, pending));
        if (lane < 16) {
          float v0 = 0.f, v1 = 0.f;
          if (t < L) {
            v0 = z0;
            v1 = z1;
            const float y = z1;
            if (y >= -bound && y <= bound) {  // identity outside the interval (transforms.py:65-77)
              const float* cw = sc + col;
              const float* ch = sc + 17 * 16 + col;
              int bin = -1;
              for (int k = 0; k <= nb; ++k) {
                const float loc = ch[k * 16] + (k == nb ? 1e-6f : 0.f);
                if (y >= loc) bin++;
              }
              bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
              const float in_cw = cw[bin * 16], in_w = cw[(bin + 1) * 16] - in_cw;
              const float in_ch = ch[bin * 16], in_h = ch[(bin + 1) * 16] - in_ch;
              const float min_d = 1e-3f;
              const float cst = logf(expf(1.f - min_d) - 1.f);
              const float ud0 = (bin == 0) ? cst : hb[(2 * nb + bin - 1) * 16 + col];
              const float ud1 = (bin == nb - 1) ? cst : hb[(2 * nb + bin) * 16 + col];
              const float d0 = min_d + softplus_f(ud0), d1 = min_d + softplus_f(ud1);
              const float delta = in_h / in_w;
              const float t1 = (y - in_ch) * (d0 + d1 - 2.f * delta);
              const float qa = t1 + in_h * (delta - d0);
              const float qb = in_h * d0 - t1;
              const float qc = -delta * (y - in_ch);
              const float disc = qb * qb - 4.f * qa * qc;
              const float root = (2.f * qc) / (-qb - sqrtf(disc));
              v1 = root * in_w + in_cw;
            }
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
    PS_STAMP(3);
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
