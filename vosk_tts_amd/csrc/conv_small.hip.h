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
    for (int cb = wave * rpi; cb < Cin; cb += step * C16_SB) {
      float v[C16_SB];
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
#pragma unroll
      for (int k = 0; k < C16_SB; ++k) {
        const int c = cb + k * step + rsub;
        const float o = tok ? conv_act_in(v[k], scale, slope) : 0.f;  // select, not multiply: stale padding may hold NaN
        if (jok && c < Cin) lds[c * ROWP + j] = o;
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
  if (tid >= 256) return;
  const int row = tid >> 4, col = n0 + (tid & 15);
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) v += red[(w * 4 + (row & 3)) * 64 + (row >> 2) * 16 + (tid & 15)];
  const int r = m0 + row;
  if (col >= P.Tout || r >= P.Cout) return;
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
    G.y[o] = v;
    if (G.y2) G.y2[o] = v;
    CONV_DBG(5);
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

// =============================================================================================================================
// conv_ls_kernel — the single-utterance DECODER regime (T = 600..2400 columns, C = 128..512, 3..11 taps, up to 3 grouped convs):
// 32 x 32 output tile per workgroup on v_mfma_f32_32x32x2_f32, the contraction split over 16 waves (4 per SIMD) by taps like
// the K-split kernel, but with that kernel's two per-MFMA global loads gone:
//   * B operand: the workgroup stages ALL input channels x (32 columns + halo) once in LDS, cooperatively and coalesced
//     (leaky-relu, MRF mean of up to three inputs, mask / ragged limit, ReflectionPad applied once per element); every tap of
//     every wave reads shifted columns of that tile (ds_read_b32, 32 consecutive columns per half-wave: conflict-free);
//   * A operand: each wave requests ALL its weight fragments (<= MAXT taps x 2 dwordx4, same packing as the other 32x32
//     kernels) before the staging barrier, so the weight stream flies under the staging phase and the MFMA loop touches no
//     global memory at all.  The L1 / texture-address path, which the K-split kernel loads with 512 B per MFMA, carries only
//     the 256 B of weights.
// Partial tiles meet in LDS (the staged tile is dead by then), 1024 threads run the shared STORE epilogue one element each.
// Serves (B = 1): conv_pre, polyphase ConvTranspose1d, ResBlock1 convs (grouped k = 3/7/11), subband_conv_post / conv_post
// (models.py:983-1040, modules.py:210-223).
template <int MAXT, int NIN>
__global__ void __launch_bounds__(1024) conv_ls_kernel(const ConvParams P) {
  constexpr int NW = 16;
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  CONV_DBG(0);
  // block -> (group, batch item, column tile, M tile): heaviest group first (the launcher sorts groups by taps)
  int id = blockIdx.x;
  const int mt = id % P.ntiles_m; id /= P.ntiles_m;
  const int nt = id % P.ntiles_n; id /= P.ntiles_n;
  const int b = id % P.B;
  const int grp = id / P.B;
  const ConvGroup& G = P.g[grp];
  const int n0 = nt * 32, m0 = mt * 32;
  const int K = G.K, dil = G.dil;
  int tap_base = 0;
  if (P.ups_u) tap_base = P.ups_shift[m0 / P.ups_cout];
  int len_raw = 0x7fffffff, rag_raw = 0x7fffffff;
  if (P.in_mask || P.out_mask) len_raw = P.len[b];
  if (P.rag) rag_raw = P.rag[b];
  const int ROWC = P.row_len;            // columns a tile needs: 32 + the launch's largest halo
  const int ROW = (ROWC + 3) & ~3;       // LDS row pitch (multiple of 4: rows are written with ds_write_b128 on the fast path)
  const int nchunks = P.Cin / CONV_CI_T;
  const int total_taps = nchunks * K;
  const int my_taps = wave < total_taps ? (total_taps - wave + NW - 1) / NW : 0;

  // ---- 1. load order (phase stamps of the first version, tools/convdbg.py: with all 22 KB of a wave's weights queued ahead of
  //         the staging loads of the other waves, the B tile was complete only after 19 k cycles and the MFMA phase -- which
  //         then runs at ~100 % of the matrix pipe -- started 8 us into a 20 us kernel):
  //           a. the weight fragments of this wave's FIRST TWO taps,
  //           b. the whole B tile [C_in][ROW] in ONE batch (element e = tid, tid + 1024, ... -> (channel, column) by an
  //              incremental cursor; registers are free, the bulk of the weights is not live yet),
  //           c. staging values -> LDS (loads return in order, so a. has landed too),
  //           d. the rest of the weight stream, which then arrives while the MFMA loop is already consuming taps in order.
  //         Staging geometry: wave w owns channels w, w + 16, ...; a lane owns column lane (and lane + 64 when the row is wider):
  //         the per-column work (reflection, clamping, validity) is done ONCE per lane, a row costs one uniform base + one load
  //         per 64 columns.  (The first version walked a flat element index with ~25 VALU instructions per element: with 4 waves
  //         per SIMD the load ISSUE alone took 11 k cycles.)
  constexpr int RB = 16;  // rows per register batch
  constexpr int NA0 = MAXT < 2 ? MAXT : 2;
  const int Cin = P.Cin;
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float* xb2 = (NIN > 1 && G.x2) ? G.x2 + (long long)b * P.x_bstride : xb;
  const float* xb3 = (NIN > 1 && G.x3) ? G.x3 + (long long)b * P.x_bstride : xb2;
  const float s3 = (NIN > 1 && G.x3) ? 1.f : 0.f;
  const float slope = P.in_slope, scale = P.in_scale;
  const int t_base = n0 - G.pad_l;
  const bool wide = ROW > 64;  // block-uniform
  unsigned toff[2];
  int tt[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int t = t_base + lane + 64 * q;
    if (P.reflect && t == -1) t = (P.Tin > 1) ? 1 : 0;
    tt[q] = t;
    toff[q] = (unsigned)(t < 0 ? 0 : (t >= P.Tin ? P.Tin - 1 : t)) * 4u;
  }
  int mb = m0 >> 5;
  if (mb >= (P.M >> 5)) mb = 0;
  const f32x4* wp = reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64 + lane;
  f32x4 a[MAXT][2];
#pragma unroll
  for (int i = 0; i < NA0; ++i) {
    const int q = wave + NW * i;
    const int qc = q < total_taps ? q : total_taps - 1;
    a[i][0] = wp[(size_t)(2 * qc) * 64];
    a[i][1] = wp[(size_t)(2 * qc + 1) * 64];
  }
  // Interior tiles (no edge, mask or ragged limit inside the window -- all but the first and last column tile of a conv): rows are
  // fetched with UNALIGNED dwordx4 loads, 64 / nvec rows per wave-instruction.  The texture-address path of a CU retires about
  // one vector memory instruction per ~14 cycles whatever its width (measured: 16 waves x 36 dword loads kept the B tile of a
  // k = 11 conv incomplete for 19 k cycles), so a staged row should be ONE wide instruction, not two narrow ones.
  bool fast = false;
  {
    int t_lim = P.Tin;
    if (P.in_mask) t_lim = __builtin_amdgcn_readfirstlane(len_raw) < t_lim ? __builtin_amdgcn_readfirstlane(len_raw) : t_lim;
    if (P.rag) { const int il = __builtin_amdgcn_readfirstlane(rag_raw) * P.rag_in_mul + P.rag_in_add; t_lim = il < t_lim ? il : t_lim; }
    fast = t_base >= 0 && t_base + ROW <= t_lim && ROW <= 256;
  }
  if (fast) {
    const int nvec = ROW >> 2, rpi = 64 / nvec;            // vectors per row, rows per wave-instruction
    const int rsub = lane / nvec, jv = lane - rsub * nvec;  // per-lane constants
    const bool lok = rsub < rpi;
    const int ngroups = (Cin + rpi - 1) / rpi;              // groups of rpi consecutive channels; wave w owns groups w, w + 16, ...
    constexpr int GB = 8;                                   // groups per register batch
    for (int g0 = wave; g0 < ngroups; g0 += NW * GB) {
      f32x4 v[GB];
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int cch = (g0 + NW * k) * rpi + rsub;
        const int cl = (cch < Cin && lok) ? cch : Cin - 1;
        const unsigned off = (unsigned)(cl * P.Tin_stride + t_base + 4 * jv) * 4u;
        f32x4 x = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb) + off);
        if (NIN > 1) {
          const f32x4 x2 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb2) + off);
          const f32x4 x3 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(xb3) + off);
          x = x + x2 + s3 * x3;
        }
        v[k] = x;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g0 == wave) CONV_DBG(1);
#pragma unroll
      for (int k = 0; k < GB; ++k) {
        const int cch = (g0 + NW * k) * rpi + rsub;
        if (cch < Cin && lok) {
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = conv_act_in(v[k][q], scale, slope);
          *reinterpret_cast<f32x4*>(lds + cch * ROW + 4 * jv) = o;
        }
      }
    }
  } else
  for (int r0 = 0; r0 * NW < Cin; r0 += RB) {
    float v0[RB], v1[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int cch = wave + NW * (r0 + r);
      const int cl = cch < Cin ? cch : Cin - 1;  // wave-uniform
      const long long ro = (long long)cl * P.Tin_stride;
      float x0 = ks_ld(xb + ro, toff[0]);
      if (NIN > 1) x0 = x0 + ks_ld(xb2 + ro, toff[0]) + s3 * ks_ld(xb3 + ro, toff[0]);
      v0[r] = x0;
      float x1 = 0.f;
      if (wide) {
        x1 = ks_ld(xb + ro, toff[1]);
        if (NIN > 1) x1 = x1 + ks_ld(xb2 + ro, toff[1]) + s3 * ks_ld(xb3 + ro, toff[1]);
      }
      v1[r] = x1;
    }
    __builtin_amdgcn_sched_barrier(0);  // every load above is issued before anything waits for len[b] / rag[b]
    if (r0 == 0) CONV_DBG(1);
    int t_lim = P.Tin;
    if (P.in_mask) t_lim = len_raw < t_lim ? len_raw : t_lim;
    if (P.rag) { const int il = rag_raw * P.rag_in_mul + P.rag_in_add; t_lim = il < t_lim ? il : t_lim; }
    const bool ok0 = lane < ROW && tt[0] >= 0 && tt[0] < t_lim;
    const bool ok1 = lane + 64 < ROW && tt[1] >= 0 && tt[1] < t_lim;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int cch = wave + NW * (r0 + r);
      if (cch < Cin) {  // wave-uniform
        if (lane < ROW) lds[cch * ROW + lane] = ok0 ? conv_act_in(v0[r], scale, slope) : 0.f;  // select: stale padding may hold NaN
        if (wide && lane + 64 < ROW) lds[cch * ROW + lane + 64] = ok1 ? conv_act_in(v1[r], scale, slope) : 0.f;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the bulk of the weight stream is requested only now
#pragma unroll
  for (int i = NA0; i < MAXT; ++i) {
    const int q = wave + NW * i;
    const int qc = q < total_taps ? q : total_taps - 1;
    a[i][0] = wp[(size_t)(2 * qc) * 64];
    a[i][1] = wp[(size_t)(2 * qc + 1) * 64];
  }
  // ragged batch: the whole tile is padding of this item (block-uniform; decided after the loads were issued)
  if (P.rag && n0 >= rag_raw * P.rag_out_mul + P.rag_out_add) return;
  CONV_DBG(2);
  __syncthreads();
  CONV_DBG(3);

  // ---- 3. MFMAs: tap q = (chunk, kk): 8 k-steps p over channel pairs 16*chunk + 2p + h, B read one tap ahead
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  {
    const float* bl = lds + h * ROW + l31 + tap_base;
    int uc = wave / K, uk = wave - (wave / K) * K;
    const int step_c = NW / K, step_k = NW - step_c * K;
    // (4 waves per SIMD cover each other's LDS latency; no per-wave double buffering: the register file is full of weights)
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
      if (i < my_taps) {
        const float* bp = bl + uc * (CONV_CI_T * ROW) + uk * dil;
        float bc[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) bc[p] = bp[2 * p * ROW];
#pragma unroll
        for (int p = 0; p < 8; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][p >> 2][p & 3], bc[p], acc, 0, 0, 0);
        uk += step_k;
        uc += step_c + (uk >= K ? 1 : 0);
        uk -= uk >= K ? K : 0;
      }
    }
  }

  // ---- 4. cross-wave reduction through LDS (the staged tile is dead) + one element per thread through the shared epilogue
  CONV_DBG(4);
  __syncthreads();
  float* red = lds;  // [wave][e][lane]
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
  __syncthreads();
  float v[1] = {0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w) v[0] += red[(w * 16 + wave) * 64 + lane];
  const int lenb = P.out_mask ? len_raw : 0x7fffffff;
  CONV_DBG(5);
  conv_epilogue_frag<EPI_STORE, 1>(P, G, b, lenb, m0 + 4 * h, wave, n0 + l31, v);
  CONV_DBG(6);
}
