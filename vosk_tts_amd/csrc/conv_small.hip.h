// conv_small.hip.h — Conv1d for the FEW-COLUMN regime of a single utterance (text encoder over T_x tokens, duration
// predictor, flow over T_y frames: N = B*T <= ~1000 columns, C = 192..768 channels).
//
// What bounds that regime (measured): not the matrix cores and not HBM, but how fast ONE compute unit can pull
// its weight slab — a 32-row tile of a K = 960 conv is 123 KB per workgroup, a CU sustains only ~7-13 B/cycle of it, and with
// 32 x 32 tiles a [384 x 150] output has just 60 workgroups on 256 CUs.  So this kernel makes the workgroups SMALL and MANY:
//   * 16 x 16 output tile per workgroup on v_mfma_f32_16x16x4_f32 (exact fp32, same 64 FLOP/clk/SIMD rate as the 32x32x2
//     form): 4x the workgroups of the K-split kernel, each streaming half the weight bytes; column tiles of one M-tile sit on
//     the same XCD (block id -> XCD is id % 8), so all but the first read their slab from that XCD's L2;
//   * the workgroup stages its whole B operand ONCE in LDS — every input channel, 16 columns + halo — cooperatively and
//     coalesced along time (leaky-relu / mask / channel flip applied once per element instead of once per fragment per tap);
//     every tap of every wave then reads shifted columns of that tile with ds_read_b32 (row pitch == 16 mod 32: the two
//     k-rows a half-wave touches fall into disjoint banks);
//   * all weight fragments of a wave (<= C16_MAXU dwordx4 per lane, pre-packed in 16x16x4 A-fragment order) are requested
//     BEFORE the staging barrier: the weight stream, the longest latency of the kernel, flies under the staging phase and
//     nothing in the MFMA loop waits on global memory;
//   * the NW waves split the contraction by tap units (16 channels x 1 tap = 4 MFMAs), two independent accumulators per wave
//     hide the 16x16x4 dependent-issue latency, partial tiles meet in LDS and 256 threads run the shared epilogues
//     (conv_mfma.hip.h: bias / cond / ReLU / mask / residual, WN gate, res-skip, coupling tail) one element each.
// Reference ops served: attentions.py:133-136,292-293 (q/k/v/o, FFN), modules.py:126-141 (WN in / res-skip layers),
// models.py:374-393 (coupling pre / post), models.py:56-63 + modules.py:96-108,363-366 (duration predictor 1x1 convs).
#pragma once
#include "conv_mfma.hip.h"

#define C16_MAXU 20  // tap units a wave keeps in registers (one dwordx4 of weights per lane each)

// src(row, ci, kk) -> packed[((mb * n_u + u) * 64 + lane) * 4 + q], tap unit u = chunk*K + kk, k4-step q:
// ci = chunk*16 + 4q + (lane>>4), row = mb*16 + (lane&15).
// A (16x16x4 MFMA): lane l holds A[i = l&15][k = l>>4]; B: lane l holds B[k = l>>4][j = l&15];
// C/D: lane l, register r: row 4*(l>>4) + r, column l&15.
template <typename F>
static void pack_conv_weights16(float* dst, int Mpad16, int Cin, int K, F src) {
  const int n_u = Cin / CONV_CI_T * K;
  for (int mb = 0; mb < Mpad16 / 16; ++mb)
    for (int u = 0; u < n_u; ++u)
      for (int lane = 0; lane < 64; ++lane)
        for (int q = 0; q < 4; ++q) {
          const int chunk = u / K, kk = u % K;
          const int ci = chunk * CONV_CI_T + 4 * q + (lane >> 4);
          const int row = mb * 16 + (lane & 15);
          dst[(((size_t)mb * n_u + u) * 64 + lane) * 4 + q] = src(row, ci, kk);
        }
}

// LDS row pitch for a staged row of `row` floats: == 16 (mod 32) so that B-fragment reads are bank-conflict free
static inline int c16_row_pitch(int row) { return row <= 16 ? 16 : (row <= 48 ? 48 : 80); }

// ---- DDSConv prologue (PRO == 1; 512 threads, C_in <= 256, 3-tap depthwise conv) ---------------------------------------
// Phase A: the finished input x_in = (x + gelu(LN2(y2))) * mask of this layer over the CONTIGUOUS column range the tile's
//          depthwise taps touch, [n0 - dil, n0 + 16 + dil), once per column -> LDS xs[c][DDS_XP] (and -> dds_xout for the tile's
//          own columns).  thread = (column slot tid & 31, channel group tid >> 5), channels cg + 16 i.
// Phase B: y1 = conv_sep(x_in) from LDS, LN1, GELU -> the B tile [c][16].  thread = (column tid & 15, group tid >> 4),
//          channels cg + 32 i.
// Channel LayerNorms are two-pass (mean, then centred second moment) like modules.LayerNorm / F.layer_norm.
#define DDS_MAXI 16  // channels per thread in phase A (C_in <= 256)
#define DDS_XP 36    // LDS pitch of xs rows: 16 + 2 * 9 columns, padded
// GELU, erf form (F.gelu default).  (A branch-free Abramowitz-Stegun erf was measured here: no change -- the prologue is bound by
// its memory round trips and barriers, not by the 24 erf evaluations per thread -- so the library erff stays.)
__device__ __forceinline__ float c16_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// sum over the NG channel groups of one value per thread; red is [NG][NC] floats
template <int NG, int NC>
__device__ __forceinline__ float c16_groupsum(float v, float* red, int cg, int j) {
  red[cg * NC + j] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) s += red[g * NC + j];
  __syncthreads();
  return s;
}

// tile: [D][16] B operand; xs: [D][DDS_XP]; red: 16 * 32 floats; par: [8][D] per-channel parameters of both phases
// (g2, b2, sb, sw0, sw1, sw2, g1, b1), fetched ONCE per workgroup by the first D threads together with the tensor loads
// -- read per thread they would be a third cold round trip in front of phase B
__device__ __forceinline__ void c16_stage_dds(const ConvParams& P, const ConvGroup& G, int b, int mt, int n0, float* tile, float* xs, float* red,
                                              float* par) {
  const int tid = threadIdx.x;
  CONV_DBG_DO(const int lane = tid & 63; const int wave = tid >> 6;)
  const int D = P.Cin, T = P.Tin;
  int bi_ = b;
  asm volatile("" : "+v"(bi_));   // vector load: keeps the cold len[b] line off the scalar-load counter (see conv16_kernel)
  const int len_raw = P.len[bi_];  // requested here, first USED after the tensor loads below are in flight
  const float invD = 1.0f / (float)D;
  const bool dw = P.dds_sw != nullptr;
  const int dil = dw ? P.dds_dil : 0;
  const int Wc = 16 + 2 * dil;  // columns of x_in this tile needs: t = n0 - dil + j
  const long long bo = (long long)b * P.x_bstride;
  const float* xb = G.x + bo;
  const float* yb = P.dds_y2 ? P.dds_y2 + bo : nullptr;
  float pv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < D) {
    if (yb) { pv[0] = P.dds_g2[tid]; pv[1] = P.dds_b2[tid]; }
    else if (P.dds_pw) { pv[0] = P.dds_pw[tid]; pv[1] = P.dds_pb[tid]; }
    if (dw) {
      pv[2] = P.dds_sb[tid]; pv[3] = P.dds_sw[tid * 3]; pv[4] = P.dds_sw[tid * 3 + 1]; pv[5] = P.dds_sw[tid * 3 + 2];
      pv[6] = P.dds_g1[tid]; pv[7] = P.dds_b1[tid];
    }
  }
  bool par_stored = false;
  // ---------------- phase A
  {
    const int jl = tid & 31, cg = tid >> 5, nci = D >> 4;
    for (int jb = 0; jb < Wc; jb += 32) {
      const int j = jb + jl;
      const bool jok = j < Wc;
      const int t = n0 - dil + j;
      const int tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
      float xv[DDS_MAXI], yv[DDS_MAXI];
      const float zv = (!yb && P.dds_pw) ? P.dds_z[(long long)b * P.dds_z_bstride + tc] : 0.f;
#pragma unroll
      for (int i = 0; i < DDS_MAXI; ++i) {
        const int c = cg + 16 * i, cc = c < D ? c : D - 1;
        xv[i] = xb[(long long)cc * T + tc];
        yv[i] = yb ? yb[(long long)cc * T + tc] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);  // every tensor load above is issued before anything waits for len[b]
      if (!par_stored) {  // (block-uniform) the parameters land with the first batch of tensor loads
        if (tid < D) {
#pragma unroll
          for (int k = 0; k < 8; ++k) par[k * D + tid] = pv[k];
        }
        par_stored = true;
        __syncthreads();
      }
      const int len_u = __builtin_amdgcn_readfirstlane(len_raw);
      const int L = len_u < T ? len_u : T;
      const bool tin = jok && t >= 0 && t < L;
      if (!yb && P.dds_pw) {
#pragma unroll
        for (int i = 0; i < DDS_MAXI; ++i) {
          const int c = cg + 16 * i, cc = c < D ? c : D - 1;
          xv[i] = par[cc] * zv + par[D + cc] + xv[i];
        }
      }
      if (yb) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < DDS_MAXI; ++i) m += i < nci ? yv[i] : 0.f;
        m = c16_groupsum<16, 32>(m, red, cg, jl) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < DDS_MAXI; ++i) { const float d = yv[i] - m; q += i < nci ? d * d : 0.f; }
        q = c16_groupsum<16, 32>(q, red, cg, jl);
        const float rstd = 1.0f / sqrtf(q * invD + 1e-5f);
#pragma unroll
        for (int i = 0; i < DDS_MAXI; ++i) {
          const int c = cg + 16 * i, cc = c < D ? c : D - 1;
          xv[i] += c16_gelu((yv[i] - m) * rstd * par[cc] + par[D + cc]);
        }
      }
      float* xo = (P.dds_xout && mt == 0 && jok && t >= n0 && t < n0 + 16 && t < T) ? P.dds_xout + bo : nullptr;
#pragma unroll
      for (int i = 0; i < DDS_MAXI; ++i) {
        const int c = cg + 16 * i;
        const float v = tin ? xv[i] : 0.f;  // x = (x + y) * mask, masked every layer (every read of x is masked)
        if (i < nci && jok) xs[c * DDS_XP + j] = v;
        if (i < nci && xo) xo[(long long)c * T + t] = v;
      }
    }
  }
  __syncthreads();
  CONV_DBG(6);
  // ---------------- phase B
  const int j = tid & 15, cg = tid >> 4, ncj = D >> 5;  // 32 groups, channels cg + 32 i
  if (!dw) {
#pragma unroll
    for (int i = 0; i < DDS_MAXI / 2; ++i) {
      const int c = cg + 32 * i;
      if (i < ncj) tile[c * 16 + j] = xs[c * DDS_XP + j];
    }
    return;
  }
  float y1[DDS_MAXI / 2];
  float m1 = 0.f;
#pragma unroll
  for (int i = 0; i < DDS_MAXI / 2; ++i) {
    const int c = cg + 32 * i, cc = c < D ? c : D - 1;
    float a = par[2 * D + cc];
#pragma unroll
    for (int k = 0; k < 3; ++k) a += par[(3 + k) * D + cc] * xs[cc * DDS_XP + j + k * dil];
    y1[i] = a;
    m1 += i < ncj ? a : 0.f;
  }
  CONV_DBG(7);
  m1 = c16_groupsum<32, 16>(m1, red, cg, j) * invD;
  float v1 = 0.f;
#pragma unroll
  for (int i = 0; i < DDS_MAXI / 2; ++i) { const float d = y1[i] - m1; v1 += i < ncj ? d * d : 0.f; }
  v1 = c16_groupsum<32, 16>(v1, red, cg, j);
  const float rstd1 = 1.0f / sqrtf(v1 * invD + 1e-5f);
#pragma unroll
  for (int i = 0; i < DDS_MAXI / 2; ++i) {
    const int c = cg + 32 * i, cc = c < D ? c : D - 1;
    if (i < ncj) tile[c * 16 + j] = c16_gelu((y1[i] - m1) * rstd1 * par[6 * D + cc] + par[7 * D + cc]);
  }
}

// ---- LayerNorm prologue (PRO == 2): in-place over the staged tile [C_in][ROWP] (ROW <= 32 columns).
// thread = (column j = tid & 31, channel group cg = tid >> 5) keeps its <= LN_MAXC channel values in registers between the two
// statistics passes (two-pass like F.layer_norm) and the write-back; gamma / beta were requested before the staging barrier.
#define C16_LN_MAXC 24
template <int NW>
struct C16LnRegs { float g[C16_LN_MAXC], b[C16_LN_MAXC]; };
template <int NW>
__device__ __forceinline__ void c16_ln_prefetch(const ConvParams& P, C16LnRegs<NW>& R) {
  constexpr int NG = NW * 2;
  const int cg = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < C16_LN_MAXC; ++i) {
    const int c = cg + NG * i, cc = c < P.Cin ? c : P.Cin - 1;
    R.g[i] = P.ln_g[cc]; R.b[i] = P.ln_b[cc];
  }
}
template <int NW>
__device__ __forceinline__ void c16_ln_tile(const ConvParams& P, const ConvGroup& G, int b, int mt, int n0, int ROW, int ROWP, int t_lim,
                                             int lenb, float* tile, float* red, const C16LnRegs<NW>& R) {
  constexpr int NG = NW * 2;  // channel groups
  const int tid = threadIdx.x, j = tid & 31, cg = tid >> 5;
  const int Cin = P.Cin, T = P.Tin;
  const bool jok = j < ROW;
  const int jc = jok ? j : 0;
  const float invC = 1.0f / (float)Cin;
  float v[C16_LN_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < C16_LN_MAXC; ++i) {
    const int c = cg + NG * i;
    v[i] = c < Cin ? tile[c * ROWP + jc] : 0.f;
    s += v[i];
  }
  red[cg * 32 + j] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) mean += red[g * 32 + j];
  mean *= invC;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < C16_LN_MAXC; ++i) { const float d = v[i] - mean; q += (cg + NG * i) < Cin ? d * d : 0.f; }
  red[cg * 32 + j] = q;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) var += red[g * 32 + j];
  const float rstd = 1.0f / sqrtf(var * invC + 1e-5f);
  const int t = n0 - G.pad_l + j;
  // conv zero padding, the mask when the consumer masks its input, and -- masked stages of a ragged batch / padded bucket --
  // columns beyond the item's length, whose raw values were never written by the (tile-skipping) producer
  const bool valid = jok && t >= 0 && t < t_lim && (!P.skip_len || t < lenb);
  const bool center = jok && t >= n0 && t < n0 + 16 && t < T;
  const float* vec = P.ln_vec ? P.ln_vec + (long long)b * P.ln_vec_stride + P.ln_vec_off : nullptr;
  const float* base = P.ln_base ? P.ln_base + (long long)b * P.x_bstride : nullptr;
  float* out = (P.ln_out && mt == 0) ? P.ln_out + (long long)b * P.x_bstride : nullptr;
  const int tcl = t < 0 ? 0 : (t >= T ? T - 1 : t);
#pragma unroll
  for (int i = 0; i < C16_LN_MAXC; ++i) {
    const int c = cg + NG * i;
    if (c < Cin) {
      float o = (v[i] - mean) * rstd * R.g[i] + R.b[i];
      if (vec) o += vec[c];
      if (base) o += base[(long long)c * T + tcl];
      o = valid ? o : 0.f;
      if (jok) tile[c * ROWP + j] = o;
      if (out && center) out[(long long)c * T + t] = o;
    }
  }
}

template <int EPI, int NW, int MAXU, int PRO = 0>
__global__ void __launch_bounds__(NW * 64) conv16_kernel(const ConvParams P) {
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // block -> (M-tile, batch item, column tile): every column tile of an M-tile on the XCD mt % 8
  CONV_DBG(0);
  const int Lb = blockIdx.x, xcd = Lb & 7, slot = Lb >> 3;
  const int per = P.ntiles_n * P.B;
  const int mt = xcd + 8 * (slot / per);
  if (mt >= P.ntiles_m) return;
  const int rr = slot - (slot / per) * per;
  const int b = rr / P.ntiles_n, nt = rr - b * P.ntiles_n;
  const ConvGroup& G = P.g[0];
  const int n0 = nt * 16, m0 = mt * 16;
  const int K = G.K, dil = G.dil;
  // A kernel of this regime is a chain of dependent COLD memory accesses (~0.8 us each: kernel arguments -> len[b] ->
  // operands -> epilogue operands -> stores), not arithmetic.  So every load whose address is known from the arguments alone
  // is requested up front -- len[b], the weight fragments, the epilogue's bias / conditioning / residual values, then the
  // staging loads (addresses clamped by T only) -- and len[b] is first USED when the staged values are written to LDS.
  constexpr bool kNeedLen = EPI == EPI_RESSKIP || EPI == EPI_COUPLE || PRO == 1;
  // (a VECTOR load: a scalar load would share the lgkm counter with the kernel-argument loads, and the first
  //  s_waitcnt lgkmcnt(0) hipcc places before ANY later argument use would wait for this cold line too)
  int len_raw = 0x7fffffff;
  if (kNeedLen || P.in_mask || P.out_mask || P.skip_len) {
    int bi = b;
    asm volatile("" : "+v"(bi));
    len_raw = P.len[bi];
  }
  const int ROW = 16 + (K - 1) * dil, ROWP = P.row_len;
  const int total_u = P.Cin / CONV_CI_T * K;
  const int my_units = wave < total_u ? (total_u - wave + NW - 1) / NW : 0;

  // ---- 1. every weight fragment of this wave, requested up front (uniform base + lane*16 bytes)
  const f32x4* wp = reinterpret_cast<const f32x4*>(G.w16) + (size_t)mt * total_u * 64 + lane;
  // (unconditional with a clamped unit index: a guarded load makes hipcc wait for the weight stream before it issues
  // the staging loads; MAXU is instantiated at 8 and C16_MAXU so short contractions do not issue dead loads)
  f32x4 a[MAXU];
#pragma unroll
  for (int i = 0; i < MAXU; ++i) {
    const int u = wave + NW * i;
    a[i] = wp[(size_t)(u < total_u ? u : total_u - 1) * 64];
  }

  // ---- 1b. epilogue operands of this thread's output element (row = tid >> 4, column = tid & 15), clamped addresses
  float ep0 = 0.f, ep1 = 0.f, ep2 = 0.f, ep3 = 0.f;
  {
    const int erow = tid >> 4, ecol = n0 + (tid & 15);
    const int colc = ecol < P.Tout ? ecol : P.Tout - 1;
    if (EPI == EPI_STORE) {
      const int r = m0 + (erow & 15), rc = r < P.Cout ? r : P.Cout - 1;
      if (G.bias) ep0 = G.bias[rc];
      if (P.bias_b) ep1 = P.bias_b[(long long)b * P.bias_b_stride + P.bias_b_off + rc];
      ep2 = P.scale_b ? P.scale_b[(long long)b * P.scale_b_stride + P.scale_b_off + rc] : 1.f;
      if (G.res) ep3 = G.res[(long long)b * P.y_bstride + (long long)rc * P.Tout_stride + colc];
    } else if (EPI == EPI_GATE) {
      const int c = mt * 8 + (erow & 7), cc = c < P.H ? c : P.H - 1;
      ep0 = G.bias[cc]; ep1 = G.bias[P.H + cc];
      if (P.bias_b) {
        const float* bb = P.bias_b + (long long)b * P.bias_b_stride + P.bias_b_off;
        ep2 = bb[cc]; ep3 = bb[P.H + cc];
      }
    } else if (EPI == EPI_RESSKIP) {
      const int r = m0 + (erow & 15), rc = r < P.Cout ? r : P.Cout - 1;
      const bool to_skip = P.last || rc >= P.H;
      const int sr = (P.last || rc < P.H) ? rc : rc - P.H;
      ep0 = G.bias[rc];
      const float* src = to_skip ? P.skip : P.io;
      if (!(to_skip && P.first)) ep1 = src[(long long)b * P.y_bstride + (long long)sr * P.Tout_stride + colc];
    } else {  // EPI_COUPLE
      const int r = m0 + (erow & 15), rc = r < P.Cout ? r : P.Cout - 1;
      const long long bo = (long long)b * P.y_bstride + colc;
      ep0 = G.bias[rc];
      ep1 = P.u[bo + (long long)(P.H - 1 - rc) * P.Tout_stride];      // x1 (before the Flip that precedes this layer)
      ep2 = P.u[bo + (long long)(2 * P.H - 1 - rc) * P.Tout_stride];  // x0
    }
  }
  C16LnRegs<NW> lnr;
  if (PRO == 2) c16_ln_prefetch<NW>(P, lnr);
  CONV_DBG(1);
  // ---- 2. stage the B operand: all C_in channels x ROW columns, rpi rows per wave-instruction
  if (PRO == 1) {
    c16_stage_dds(P, G, b, mt, n0, lds, lds + P.Cin * 16, lds + P.Cin * (16 + DDS_XP), lds + P.Cin * (16 + DDS_XP) + 16 * 32);
  } else {
    const int rpi = ROW <= 16 ? 4 : (ROW <= 21 ? 3 : (ROW <= 32 ? 2 : 1));
    const int seg = 64 / rpi;
    const int rsub = lane / seg, j = lane - rsub * seg;
    const bool jok = j < ROW && rsub < rpi;
    const int t = n0 - G.pad_l + j;
    const int tc = t < 0 ? 0 : (t >= P.Tin ? P.Tin - 1 : t);
    const float* xb = G.x + (long long)b * P.x_bstride;
    const float* xb2 = G.x2 ? G.x2 + (long long)b * P.x_bstride : xb;
    const float slope = P.in_slope, scale = P.in_scale;
    const int Cin = P.Cin, split = P.x_split ? P.x_split : 0x7fffffff;
    const int step = NW * rpi;
    // batches of C16_SB rows per thread: every load of a batch is issued before the first LDS write, so a batch costs one
    // memory round trip (C_in = 192, 4 waves, 3 rows per instruction: the whole tile is ONE batch)
    constexpr int C16_SB = 16;
    const int chs = P.x_ch_sign * P.Tin_stride;           // element offset of one channel step (negative: Flip folded in)
    const int ch0 = P.x_ch_off * P.Tin_stride + tc;         // (x_ch_off + c*sign) >= 0 for every channel
    // PRO == 3: LayerNorm of the staged tensor from the producer's per-block statistics; a thread's elements share ONE column
    float ln_mean = 0.f, ln_rstd = 1.f;
    float stm[16], stq[16];
    if (PRO == 3) {
      const float* st = P.ln_stat_in + ((long long)b * P.ln_nmb * P.Tin + tc) * 2;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ic = i < P.ln_nmb ? i : P.ln_nmb - 1;
        stm[i] = st[(long long)ic * P.Tin * 2];
        stq[i] = st[(long long)ic * P.Tin * 2 + 1];
      }
    }
    for (int cb = wave * rpi; cb < Cin; cb += step * C16_SB) {
      float v[C16_SB];
      float lg[C16_SB], lb[C16_SB], lbase[C16_SB];
      if (PRO == 3) {
        const float* vec = P.ln_vec ? P.ln_vec + (long long)b * P.ln_vec_stride + P.ln_vec_off : nullptr;
        const float* base = P.ln_base ? P.ln_base + (long long)b * P.x_bstride : nullptr;
#pragma unroll
        for (int k = 0; k < C16_SB; ++k) {
          const int c = cb + k * step + rsub;
          const int cc = c < Cin ? c : Cin - 1;
          lg[k] = P.ln_g[cc];
          lb[k] = P.ln_b[cc] + (vec ? vec[cc] : 0.f);
          lbase[k] = base ? base[(long long)cc * P.Tin + tc] : 0.f;
        }
      }
      if (P.x_split == 0) {  // uniform base pointer + 32-bit lane offset: one address add per load
#pragma unroll
        for (int k = 0; k < C16_SB; ++k) {
          const int c = cb + k * step + rsub;
          const int cc = c < Cin ? c : Cin - 1;
          v[k] = ks_ld(xb, (unsigned)(ch0 + cc * chs) * 4u);
        }
      } else {               // channel-concatenated second input (cat((x, x2), dim=1) never materialised)
#pragma unroll
        for (int k = 0; k < C16_SB; ++k) {
          const int c = cb + k * step + rsub;
          const int cc = c < Cin ? c : Cin - 1;
          const bool second = cc >= split;
          v[k] = ks_ld(second ? xb2 : xb, (unsigned)(ch0 + (second ? cc - split : cc) * chs) * 4u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the batch's loads are issued before anything below waits for len[b]
      len_raw = __builtin_amdgcn_readfirstlane(len_raw);
      const int t_lim_ = (P.in_mask && len_raw < P.Tin) ? len_raw : P.Tin;
      const bool tok = jok && t >= 0 && t < t_lim_;
      if (PRO == 3) {
        if (cb == wave * rpi) {  // merge the blocks' (mean, M2) in fixed order: equal-sized blocks of 16 rows, the last one shorter
          const int nmb = P.ln_nmb, last_n = Cin - 16 * (nmb - 1);
          float ms = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) ms += i < nmb ? stm[i] * (float)(i == nmb - 1 ? last_n : 16) : 0.f;
          ln_mean = ms / (float)Cin;
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float d = stm[i] - ln_mean;
            q += i < nmb ? stq[i] + (float)(i == nmb - 1 ? last_n : 16) * d * d : 0.f;
          }
          ln_rstd = 1.0f / sqrtf(q / (float)Cin + 1e-5f);
        }
        // conv zero padding, the mask, and columns beyond the item's length of a tile-skipping producer (garbage statistics there)
        const bool valid = tok && (!P.skip_len || t < len_raw);
        const bool center = jok && t >= n0 && t < n0 + 16 && t < P.Tin && mt == 0 && P.ln_out;
        float* out = P.ln_out ? P.ln_out + (long long)b * P.x_bstride : nullptr;
#pragma unroll
        for (int k = 0; k < C16_SB; ++k) {
          const int c = cb + k * step + rsub;
          float o = (v[k] - ln_mean) * ln_rstd * lg[k] + lb[k] + lbase[k];
          o = valid ? o : 0.f;  // select: padding may hold NaN
          if (jok && c < Cin) lds[c * ROWP + j] = o;
          if (center && c < Cin) out[(long long)c * P.Tin + t] = o;
        }
      } else {
#pragma unroll
      for (int k = 0; k < C16_SB; ++k) {
        const int c = cb + k * step + rsub;
        const float o = tok ? conv_act_in(v[k], scale, slope) : 0.f;  // select, not multiply: stale padding may hold NaN
        if (jok && c < Cin) lds[c * ROWP + j] = o;
      }
      }
    }
  }
  const int lenb = __builtin_amdgcn_readfirstlane(len_raw);
  const int t_lim = (P.in_mask && lenb < P.Tin) ? lenb : P.Tin;
  // masked stage of a ragged batch / padded bucket: the tile is all padding (block-uniform; decided only now so that the
  // loads above did not wait for len[b] -- a skipped tile has merely prefetched for nothing)
  if (P.skip_len && n0 >= lenb) return;
  __syncthreads();
  CONV_DBG(2);
  if (PRO == 2) {
    c16_ln_tile<NW>(P, G, b, mt, n0, ROW, ROWP, t_lim, lenb, lds, lds + P.Cin * ROWP, lnr);
    __syncthreads();
  }

  // ---- 3. MFMAs: unit u = (chunk c, tap kk); k4-step q covers channels 16c + 4q + (lane>>4)
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  {
    const float* bl = lds + (lane >> 4) * ROWP + (lane & 15);
    int uc = wave / K, uk = wave - (wave / K) * K;
    const int step_c = NW / K, step_k = NW - step_c * K;
    // B fragments are read one unit ahead of the MFMAs that consume them (LDS latency ~ one unit's MFMA time)
    float bc[4], bn[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const float* bp = bl + uc * (CONV_CI_T * ROWP) + uk * dil;
      bc[0] = bp[0]; bc[1] = bp[4 * ROWP]; bc[2] = bp[8 * ROWP]; bc[3] = bp[12 * ROWP];
    }
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      if (i < my_units) {
        uk += step_k;
        uc += step_c + (uk >= K ? 1 : 0);
        uk -= uk >= K ? K : 0;
        if (i + 1 < my_units) {
          const float* bp = bl + uc * (CONV_CI_T * ROWP) + uk * dil;
          bn[0] = bp[0]; bn[1] = bp[4 * ROWP]; bn[2] = bp[8 * ROWP]; bn[3] = bp[12 * ROWP];
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][0], bc[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][1], bc[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][2], bc[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][3], bc[3], acc1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) bc[q] = bn[q];
      }
    }
  }

  // ---- 4. cross-wave reduction (the staged tile is dead: reuse its LDS) and one-element-per-thread epilogue
  CONV_DBG(3);
  __syncthreads();
  float* red = lds;  // [wave][r][lane]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc0[r] + acc1[r];
  __syncthreads();
  // (operands ep0..ep3 were requested at the top of the kernel; semantics identical to conv_epilogue_frag / conv_epilogue_gate)
  if (EPI == EPI_GATE) {
    // packed 16-row block = [8 tanh rows | 8 sigmoid rows] of channels 8*mt .. 8*mt + 7 (commons.py:100-107)
    if (tid >= 128) return;
    const int ch = tid >> 4, col = n0 + (tid & 15);
    float at = 0.f, as = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      at += red[(w * 4 + (ch & 3)) * 64 + (ch >> 2) * 16 + (tid & 15)];
      as += red[(w * 4 + (ch & 3)) * 64 + ((ch >> 2) + 2) * 16 + (tid & 15)];
    }
    const int c = mt * 8 + ch;
    const float tv = tanhf(at + ep0 + ep2);
    const float sv = 1.0f / (1.0f + __expf(-(as + ep1 + ep3)));
    if (col < P.Tout && c < P.H) G.y[(long long)b * P.y_bstride + (long long)c * P.Tout_stride + col] = tv * sv;
    return;
  }
  const bool want_stat = EPI == EPI_STORE && P.ln_stat_out != nullptr;  // kernel-uniform
  if (tid >= 256 && !want_stat) return;
  const bool act = tid < 256;
  const int row = (tid >> 4) & 15, col = n0 + (tid & 15);
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) v += red[(w * 4 + (row & 3)) * 64 + (row >> 2) * 16 + (tid & 15)];
  const int r = m0 + row;
  const bool ok = act && col < P.Tout && r < P.Cout;
  if (!ok && !want_stat) return;
  if (EPI == EPI_STORE) {
    v += ep0;
    v += ep1;
    if (P.relu == 1) v = v > 0.f ? v : 0.f;
    else if (P.relu == 2) v = v / (1.0f + __expf(-v));
    else if (P.relu == 3) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (P.out_mask && col >= lenb) v = 0.f;
    v *= ep2;
    v += ep3;
    const long long o = (long long)b * P.y_bstride + (long long)r * P.Tout_stride + col;
    if (ok) {
      G.y[o] = v;
      if (G.y2) G.y2[o] = v;
    }
    CONV_DBG(5);
    if (want_stat) {
      // per-column mean and centred second moment over this block's rows (two passes, like F.layer_norm), for the LayerNorm
      // the consumer applies while staging (PRO == 3): rows of a wave sit in lane bits 4..5, the 4 row-waves meet in LDS
      float* st = lds + NW * 4 * 64;  // behind the reduction buffer
      const int nrow = P.Cout - m0 < 16 ? P.Cout - m0 : 16;
      float sv = ok ? v : 0.f;
      sv += __shfl_xor(sv, 16, 64);
      sv += __shfl_xor(sv, 32, 64);
      if (act && lane < 16) st[wave * 16 + lane] = sv;
      __syncthreads();
      const int c15 = tid & 15;
      const float mean = (st[c15] + st[16 + c15] + st[32 + c15] + st[48 + c15]) / (float)nrow;
      const float d = ok ? v - mean : 0.f;
      float q = d * d;
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      __syncthreads();
      if (act && lane < 16) st[wave * 16 + lane] = q;
      __syncthreads();
      if (tid < 16 && col < P.Tout) {
        float* dst = P.ln_stat_out + (((long long)b * P.ln_nmb + mt) * P.Tout + col) * 2;
        dst[0] = mean;
        dst[1] = st[c15] + st[16 + c15] + st[32 + c15] + st[48 + c15];
      }
    }
  } else if (EPI == EPI_RESSKIP) {
    // rows < H update x in place (modules.py:171); rows >= H (or every row of the last layer) feed the skip accumulator
    const bool to_skip = P.last || r >= P.H;
    const int sr = (P.last || r < P.H) ? r : r - P.H;
    const bool valid = col < lenb;
    float o = ep1 + v + ep0;
    if (to_skip) { if (P.last && !valid) o = 0.f; }
    else if (!valid) o = 0.f;
    float* dst = to_skip ? P.skip : P.io;
    dst[(long long)b * P.y_bstride + (long long)sr * P.Tout_stride + col] = o;
  } else {  // EPI_COUPLE: new z = cat(x0, (x1 - m) * mask) with the following Flip folded in (models.py:390-392)
    const long long bo = (long long)b * P.y_bstride + col;
    P.io[bo + (long long)(P.H + r) * P.Tout_stride] = col < lenb ? (ep1 - (v + ep0)) : 0.f;
    P.io[bo + (long long)r * P.Tout_stride] = ep2;
  }
}

// ---------------------------------------------------------------------------------------------
// conv_wp_kernel — single-utterance decoder ResBlock convs (T = 600..2400 columns, C = 128..256, 3..11 taps, 3 grouped convs).
// What the older kernels showed on these launches (tools/convdbg.py phase stamps, PMC): the K-split kernel issues one global load
// per MFMA and its matrix pipe is ~43 % busy from start to end; a first LDS version (whole workgroup stages the full B tile, barrier,
// then MFMAs; removed) ran its MFMA phase at ~100 % -- but only after a staging phase nothing overlapped (one workgroup per CU).
// Here every wave is its own pipeline: the contraction is split over the NW waves by 16-CHANNEL CHUNK, a wave stages ITS chunk
// (16 rows x (32 + halo) columns, leaky-relu applied once) into a wave-private LDS slab with 4..8 unaligned dwordx4 loads, and then
// issues 8 K MFMAs on it with B fragments read from LDS at shifted columns one tap ahead and weight fragments requested two taps
// ahead.  No workgroup barrier before the final reduction, 48 KB of LDS and <= 128 registers per wave: 2 workgroups per CU overlap
// each other's start-up, reduction and epilogue.  Measured (K = 11, C = 128, T = 2400): the first wave of a SIMD runs at 76 cycles
// per MFMA (64 = peak); what is left is the start-up -- every launch begins with cold L2s, and the last waves of a workgroup get
// their slab ~5 k cycles after the first ones (outstanding-miss limited fetch from the Infinity Cache).
// ---------------------------------------------------------------------------------------------
#define WP_PITCH 96  // slab row pitch in floats: == 32 mod 64 (rows 2p / 2p + 1 of a B fragment land in different bank halves)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <int V> struct wp_int { static constexpr int value = V; };
// KT = compile-time tap count (3 / 7 / 11: the decoder's ResBlock kernels, fully unrolled so that every register of the weight
// and B-fragment rings is named statically -- a rolled loop makes hipcc rotate the rings with v_mov behind s_waitcnt vmcnt(0)),
// 0 = run-time tap count (any other conv; same arithmetic, rolled loop).
template <int NW, int KT>
__device__ __forceinline__ void conv_wp_body(const ConvParams& P, const ConvGroup& G, float* lds, int mt, int nt, int b, int wave, int lane) {
  constexpr int NE = 16 / NW;
  const int h = lane >> 5, l31 = lane & 31;
  const int n0 = nt * 32, m0 = mt * 32;
  const int K = KT ? KT : G.K, dil = G.dil;
  const int nchunks = P.Cin / CONV_CI_T;
  const int my_chunks = wave < nchunks ? (nchunks - wave + NW - 1) / NW : 0;
  const int last_c = wave + NW * (my_chunks > 0 ? my_chunks - 1 : 0);
  // ---- request order: weight fragments of the first two taps (no geometry needed), len[b] / rag[b] (vector loads: needed only
  //      after the slab loads have been issued), then the first slab
  int mb = m0 >> 5;
  if (mb >= (P.M >> 5)) mb = 0;
  const f32x4* wp = reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64 + lane;
  f32x4 a[2][2];        // weight fragments of two consecutive taps; a slot is re-requested right after its last MFMA, two taps ahead
  // (Tried: rotating the tap order per column tile so that the workgroups of an XCD that share an m-block do not pull the same
  // weight stream through the fabric in lockstep -- no measurable change, dropped.)
  int lci = 0, lk = 0;  // load cursor of the weight stream (this wave's chunk index, tap)
  auto load_a_half = [&](f32x4 (&d)[2], int half) {  // half 1 advances the cursor
    const int c = wave + NW * lci;
    const bool live = c <= last_c && my_chunks > 0;
    const size_t sg = 2 * ((size_t)(live ? c : last_c) * K + (live ? lk : K - 1));
    d[half] = wp[(sg + half) * 64];
    if (half) {
      ++lk;
      if (lk == K) { lk = 0; ++lci; }
    }
  };
  auto load_a = [&](f32x4 (&d)[2]) { load_a_half(d, 0); load_a_half(d, 1); };
  load_a(a[0]);
  load_a(a[1]);
  int bi = b;
  asm volatile("" : "+v"(bi));
  int len_raw = 0x7fffffff, rag_raw = 0x7fffffff;
  if (P.in_mask || P.out_mask || P.skip_len) len_raw = P.len[bi];
  int rag_cap = 0x7fffffff;
  if (P.rag) { rag_raw = P.rag[bi]; rag_cap = P.rag[P.B]; }

  // slab geometry of this group: one load instruction = vector v16 (+ 16 in the second pass) of rows 4 j + r4: 4 instructions stage a
  // 16-row slab of up to 64 columns, 8 one of up to 96
  const int row_len = 32 + (K - 1) * dil;
  const int nvec = (row_len + 3) >> 2;
  const bool two = nvec > 16;  // block-uniform
  const int t_base = n0 - G.pad_l;
  const int r4 = lane >> 4, v16 = lane & 15;
  int tvp[2], tcp[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    tvp[ps] = t_base + 4 * (v16 + 16 * ps);
    const int tc = tvp[ps] < 0 ? 0 : tvp[ps];
    tcp[ps] = tc > P.Tin - 4 ? P.Tin - 4 : tc;  // clamped (always valid) vector start; exact for interior tiles
  }
  const unsigned lrow = (unsigned)(r4 * P.Tin_stride);
  float* sl = lds + wave * (CONV_CI_T * WP_PITCH);
  float* sw = sl + r4 * WP_PITCH + 4 * v16;  // + 4 j rows, + 64 columns in the second pass
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float slope = P.in_slope;
  f32x4 v0[4], v1[4];
  const float* xb2 = G.x2 ? G.x2 + (long long)b * P.x_bstride : xb;  // channel-concatenated second input (x_split): chunks at or beyond the split
  const int split_chunk = P.x_split ? P.x_split / CONV_CI_T : 0x7fffffff;
  auto load_slab = [&](int ci) {
    const int c = wave + NW * ci;
    const float* xc = c >= split_chunk ? xb2 + (long long)(c - split_chunk) * CONV_CI_T * P.Tin_stride : xb + (long long)c * CONV_CI_T * P.Tin_stride;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      v0[j] = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(xc + 4 * j * P.Tin_stride) + (lrow + (unsigned)tcp[0]) * 4u);
    if (two) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        v1[j] = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(xc + 4 * j * P.Tin_stride) + (lrow + (unsigned)tcp[1]) * 4u);
    }
  };
  if (my_chunks > 0) load_slab(0);
  // epilogue operands (bias, residual) of the NE accumulator elements this wave finishes: requested now, consumed after the reduction
  // (their cold misses would otherwise sit on the critical path of the tail: ~1 us per launch)
  const int ecol = n0 + l31;
  const int ecolc = ecol < P.Tout ? ecol : P.Tout - 1;
  long long eoff[NE];
  bool eok[NE];
  float ebias[NE], eres[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = NE * wave + i;
    const int r = m0 + 4 * h + (e & 3) + 8 * (e >> 2);
    const int rc = r < P.Cout ? r : P.Cout - 1;
    eok[i] = ecol < P.Tout && r < P.Cout;
    eoff[i] = (long long)b * P.y_bstride + (long long)rc * P.Tout_stride + ecolc;
    // UNCONDITIONAL loads from always-valid addresses + selects: behind `if (ptr)` hipcc sinks the load to its use, i.e. puts the
    // cold miss of the residual back at the end of the kernel (measured: 3.5 k cycles between the reduction barrier and the stores)
    const float bv = (G.bias ? G.bias : G.w)[rc];
    const float rv = (G.res ? G.res : G.y)[eoff[i]];
    ebias[i] = G.bias ? bv : 0.f;
    eres[i] = G.res ? rv : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  CONV_DBG(1);
  int t_lim = P.Tin;
  if (P.in_mask) { const int l = __builtin_amdgcn_readfirstlane(len_raw); t_lim = l < t_lim ? l : t_lim; }
  if (P.skip_len && n0 >= __builtin_amdgcn_readfirstlane(len_raw)) return;  // masked stage: the whole tile lies in this item's padding
  if (P.rag) {
    const int rl = __builtin_amdgcn_readfirstlane(rag_raw);
    const int rc = __builtin_amdgcn_readfirstlane(rag_cap);
    if (n0 >= conv_rag_limit(rl, rc, P.rag_out_mul, P.rag_out_add, P.rag_out_cap_add)) return;  // whole tile is padding of this item (block-uniform)
    const int il = conv_rag_limit(rl, rc, P.rag_in_mul, P.rag_in_add);
    t_lim = il < t_lim ? il : t_lim;
  }
  const bool interior = t_base >= 0 && t_base + 4 * nvec <= t_lim;
  // leaky-relu (0 <= slope <= 1) once per element, then the slab.  Edge tiles (first / last columns, mask or ragged limit inside
  // the window): every element they need lies inside the clamped vector, shifted by sh = wanted - loaded start (|sh| <= 3)
  auto put = [&](f32x4 (&v)[4], int ps) {
    if (v16 + 16 * ps >= nvec) return;
    if (interior && slope == 1.f) {  // (block-uniform) plain copy: no VALU work between the load and the slab
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(sw + 4 * j * WP_PITCH + 64 * ps) = v[j];
    } else if (interior) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = fmaxf(v[j][q], v[j][q] * slope);
        *reinterpret_cast<f32x4*>(sw + 4 * j * WP_PITCH + 64 * ps) = o;
      }
    } else {
      const int sh = tvp[ps] - tcp[ps];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = tvp[ps] + q, idx = q + sh;
          const float x = idx <= 0 ? v[j][0] : (idx == 1 ? v[j][1] : (idx == 2 ? v[j][2] : v[j][3]));
          o[q] = (t >= 0 && t < t_lim) ? fmaxf(x, x * slope) : 0.f;  // select: stale padding may hold NaN
        }
        *reinterpret_cast<f32x4*>(sw + 4 * j * WP_PITCH + 64 * ps) = o;
      }
    }
  };
  auto stage = [&]() {
    put(v0, 0);
    if (two) put(v1, 1);
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // ---- 8 K MFMAs on the slab: B fragment of (tap, channel pair p) = rows 2p + h at columns l31 + tap * dil, read one tap ahead.
  //      PAR = which weight slot holds the chunk's first tap (flips per chunk when the tap count is odd).
  const float* bl = sl + h * WP_PITCH + l31;
  auto read_b = [&](float (&d)[8], int tap) {
#pragma unroll
    for (int p = 0; p < 8; ++p) d[p] = bl[2 * p * WP_PITCH + tap * dil];
  };
  auto mfma8 = [&](const f32x4 (&w)[2], const float (&d)[8]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[p >> 2][p & 3], d[p], acc, 0, 0, 0);
  };
  auto taps = [&](auto par) {
    constexpr int PAR = decltype(par)::value;
    float bb[2][8];
    read_b(bb[0], 0);
    if (KT) {
      // program order pinned per tap (hipcc otherwise sinks every load to its use: B fragments one MFMA pair ahead, weights one
      // tap ahead): [B of tap + 1] | 4 MFMAs, first weight vector of tap + 2, 4 MFMAs, second weight vector |
#pragma unroll
      for (int tap = 0; tap < (KT ? KT : 1); ++tap) {
        f32x4 (&w)[2] = a[(tap + PAR) & 1];
        const float (&d)[8] = bb[tap & 1];
        if (tap + 1 < KT) read_b(bb[(tap + 1) & 1], tap + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[0][p], d[p], acc, 0, 0, 0);
        load_a_half(w, 0);
#pragma unroll
        for (int p = 4; p < 8; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[1][p & 3], d[p], acc, 0, 0, 0);
        load_a_half(w, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 1
      for (int tap = 0; tap < K; tap += 2) {
        read_b(bb[1], tap + 1 < K ? tap + 1 : tap);
        mfma8(a[PAR], bb[0]);
        load_a(a[PAR]);
        if (tap + 1 < K) {
          read_b(bb[0], tap + 2 < K ? tap + 2 : tap + 1);
          mfma8(a[PAR ^ 1], bb[1]);
          load_a(a[PAR ^ 1]);
        }
      }
    }
  };
  constexpr bool FLIP = KT & 1;  // static tap counts: the odd ones alternate the slot parity per chunk
#pragma unroll 1
  for (int ci = 0; ci < my_chunks; ci += 2) {
    // (VALU issue on a SIMD is arbitrated by priority, then age: the SECOND wave of a SIMD gets its ~60 staging instructions through
    // only when the first one's MFMA stream ends -- slab complete at 13.4 k cycles instead of 7.5 k.  Raising the staging priority
    // with s_setprio fixes that and makes the kernel SLOWER: two interleaved dependency-paced MFMA chains run at 111 cycles per
    // MFMA on the shared pipe, one after the other at 74-82.  Left as the hardware schedules it.)
    stage();
    if (ci + 1 < my_chunks) load_slab(ci + 1);
    if (ci == 0) CONV_DBG(2);
    taps(wp_int<0>{});
    if (!KT && (K & 1)) {  // run-time odd tap count: the next chunk's first tap sits in slot 1 -> swap once per chunk
      const f32x4 t0 = a[0][0], t1 = a[0][1];
      a[0][0] = a[1][0]; a[0][1] = a[1][1];
      a[1][0] = t0; a[1][1] = t1;
    }
    if (ci + 1 < my_chunks) {
      stage();
      if (ci + 2 < my_chunks) load_slab(ci + 2);
      taps(wp_int<FLIP ? 1 : 0>{});
      if (!KT && (K & 1)) {
        const f32x4 t0 = a[0][0], t1 = a[0][1];
        a[0][0] = a[1][0]; a[0][1] = a[1][1];
        a[1][0] = t0; a[1][1] = t1;
      }
    }
  }
  CONV_DBG(3);
  // ---- cross-wave reduction: the partial tile goes into this wave's own (dead) slab, one barrier, NE elements per wave
#pragma unroll
  for (int e = 0; e < 16; ++e) sl[e * 64 + lane] = acc[e];
  __syncthreads();
  CONV_DBG(4);
  float sum[NE];
#pragma unroll
  for (int ee = 0; ee < NE; ++ee) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += lds[w * (CONV_CI_T * WP_PITCH) + (NE * wave + ee) * 64 + lane];
    sum[ee] = r;
  }
  if (P.bias_b || P.scale_b || P.relu) {  // (kernel-uniform) per-item bias / gate, output activation: the shared epilogue
    const int lenb = P.out_mask ? len_raw : 0x7fffffff;
    conv_epilogue_frag<EPI_STORE, NE>(P, G, b, lenb, m0 + 4 * h, NE * wave, ecol, sum);
    CONV_DBG(5);
    return;
  }
  // plain epilogue on the prefetched operands: bias, mask, residual
  const bool masked = P.out_mask && ecol >= len_raw;
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    float r = sum[i] + ebias[i];
    if (masked) r = 0.f;
    r += eres[i];
    if (eok[i]) {
      G.y[eoff[i]] = r;
      if (G.y2) G.y2[eoff[i]] = r;
    }
  }
  CONV_DBG(5);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 4) conv_wp_kernel(const ConvParams P) {
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  CONV_DBG(0);
  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  const ConvGroup& G = P.g[grp];
  switch (G.K) {  // block-uniform
    case 3: conv_wp_body<NW, 3>(P, G, lds, mt, nt, b, wave, lane); break;
    case 7: conv_wp_body<NW, 7>(P, G, lds, mt, nt, b, wave, lane); break;
    case 11: conv_wp_body<NW, 11>(P, G, lds, mt, nt, b, wave, lane); break;
    default: conv_wp_body<NW, 0>(P, G, lds, mt, nt, b, wave, lane); break;
  }
}

