// conv_mfma.hip.h — the hot kernel of the path: Conv1d / polyphase ConvTranspose1d as an
// implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32,
// 157 TFLOP/s peak), for channel-major [B,C,T] activations.
//
//   out[b, co, t] = bias[co] + sum_{ci,kk} W[co,ci,kk] * act(x[b, ci, t + kk*dil - pad_l])
//
// GEMM view: M = C_out rows, N = time columns, K = C_in * taps.
//   * B operand (activations): a CI_T-channel x (N_T + halo) window is staged ONCE in LDS per
//     chunk (leaky-relu / mask / MRF-sum / reflection applied at staging) and re-read for every
//     tap with a shifted column offset -> HBM/L2 sees each activation once per M-tile row.
//     ds_read_b32 with 32 consecutive columns per half-wave is bank-conflict free.
//   * A operand (weights): pre-packed on the host in exact MFMA-fragment order
//     [mblock32][stepgroup][lane64][4 k-steps] so every wave streams its fragments with fully
//     coalesced 16-byte loads straight from L2 (weights of one conv are <= 2.9 MB, L2-resident),
//     double-buffered in registers one tap ahead.  No LDS traffic for weights.
//   * Wave tile = (MI*32) x (NI*32) of 32x32 MFMA accumulators; WG = WM x WN waves (256 threads).
//   * Up to 3 independent convolutions ("groups": the k=3/7/11 ResBlocks of one MRF stage, same
//     input) share one launch so a single utterance still fills 256 CUs.
//   * 1-D grid with a bijective XCD remap: all M-tiles and groups of one time tile run on the
//     same XCD (they share the staged activation window in that XCD's L2).
//
// Reference ops served by this kernel: every nn.Conv1d on the inference path
// (training/vits2/models.py:983,1000, modules.py:126-141,190-206, attentions.py:133-136,292-293),
// ConvTranspose1d (models.py:986-990) in polyphase form, fused with
// fused_add_tanh_sigmoid_multiply (commons.py:100-107), the WN res/skip update
// (modules.py:168-175) and the coupling-layer tail (models.py:390-392).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CONV_CI_T 16        // input channels per LDS chunk (8 MFMA k-steps per tap)
#define CONV_MAX_HALO 64    // (K-1)*dil <= 64 (largest on the path: (11-1)*5 = 50)
#define CONV_MAX_GROUPS 3

enum ConvEpilogue {
  EPI_STORE = 0,    // y = [relu](acc + bias + bias_b[b]) [*mask] [+ res]
  EPI_GATE = 1,     // WN gate: y[c] = tanh(a_t + g_t) * sigmoid(a_s + g_s)          (MI == 2)
  EPI_RESSKIP = 2,  // WN res/skip: rows < H: io = (io + v) * mask ; rows >= H: skip (+)= v
  EPI_COUPLE = 3,   // coupling tail with the following Flip folded in (see engine.hip)
};

struct ConvGroup {
  const float* x;     // input [B, C, Tin]
  const float* x2;    // optional 2nd/3rd inputs summed at staging (MRF mean, models.py:1030-1036)
  const float* x3;
  const float* w;     // packed weights (pack_conv_weights)
  const float* bias;  // [Cout] or null
  float* y;           // output
  const float* res;   // residual added after mask (EPI_STORE) or null
  int K;              // taps
  int dil;            // dilation
  int pad_l;          // input index = t + kk*dil - pad_l
  int n_sg;           // step-groups per m-block = Cin/16 * 2K
};

struct ConvParams {
  ConvGroup g[CONV_MAX_GROUPS];
  int n_groups;
  int B;
  int Cin;            // contraction channels, multiple of 16
  int x_ch_off;       // input channel row = x_ch_off + ci * x_ch_sign (Flip folded into the read)
  int x_ch_sign;
  long long x_bstride;
  int Tin;            // valid input length
  int Tin_stride;     // input row stride
  int M;              // packed rows (multiple of 32)
  int Cout;           // rows actually stored
  int Tout;           // output columns (for polyphase: input positions q)
  int Tout_stride;
  long long y_bstride;
  float in_slope;     // leaky-relu slope applied at staging (1 = identity)
  float in_scale;     // multiplies the (summed) input at staging
  int in_mask;        // zero input where t >= len[b]
  int reflect;        // ReflectionPad1d((1,0)) folded into staging: index -1 reads index 1
  const int* len;     // [B] lengths for masks
  int relu;
  int out_mask;
  const float* bias_b;  // per-batch bias [B][bias_b_stride] or null (cond(g) terms)
  int bias_b_stride;
  int bias_b_off;
  int ups_u;          // 0, or polyphase factor: packed row = phase*ups_cout + co, output t = u*q + phase
  int ups_cout;
  int ups_shift[8];   // per-phase tap base (already includes +pad_l)
  float* io;          // EPI_RESSKIP: x in/out ; EPI_COUPLE: new z
  const float* u;     // EPI_COUPLE: previous z
  float* skip;        // EPI_RESSKIP
  int H;              // EPI_RESSKIP: split row ; EPI_COUPLE: half channels ; EPI_GATE: hidden
  int first;          // EPI_RESSKIP: first layer (store skip instead of accumulate)
  int last;           // EPI_RESSKIP: last layer (M == H, everything is skip; apply mask)
  int ntiles_m, ntiles_n;
  int row_len;        // LDS row = N_T + halo
};

__device__ __forceinline__ float conv_act_in(float v, float scale, float slope) {
  v *= scale;
  return v > 0.f ? v : v * slope;
}

// ---- shared epilogue: one accumulator element (row, col) of the output tile --------------------
template <int EPI>
__device__ __forceinline__ void conv_epilogue_elem(const ConvParams& P, const ConvGroup& G, int b, int lenb, int row, int col, float v) {
  if (row >= P.Cout || col >= P.Tout) return;
  if (EPI == EPI_STORE) {
    if (P.ups_u) {
      const int phase = row / P.ups_cout, co = row - phase * P.ups_cout;
      if (G.bias) v += G.bias[co];
      G.y[(long long)b * P.y_bstride + (long long)co * P.Tout_stride + (long long)col * P.ups_u + phase] = v;
    } else {
      if (G.bias) v += G.bias[row];
      if (P.bias_b) v += P.bias_b[(long long)b * P.bias_b_stride + P.bias_b_off + row];
      if (P.relu) v = v > 0.f ? v : 0.f;
      if (P.out_mask && col >= lenb) v = 0.f;
      const long long o = (long long)b * P.y_bstride + (long long)row * P.Tout_stride + col;
      if (G.res) v += G.res[o];
      G.y[o] = v;
    }
  } else if (EPI == EPI_RESSKIP) {
    v += G.bias[row];
    const bool valid = col < lenb;
    if (P.last || row >= P.H) {
      const int sr = P.last ? row : row - P.H;
      const long long o = (long long)b * P.y_bstride + (long long)sr * P.Tout_stride + col;
      float s = P.first ? v : P.skip[o] + v;
      if (P.last && !valid) s = 0.f;  // output * x_mask (modules.py:176)
      P.skip[o] = s;
    } else {
      const long long o = (long long)b * P.y_bstride + (long long)row * P.Tout_stride + col;
      P.io[o] = valid ? P.io[o] + v : 0.f;  // x = (x + res_acts) * x_mask (modules.py:171)
    }
  } else if (EPI == EPI_COUPLE) {
    // previous z = u (before the Flip that precedes this layer); this layer's logical input is
    // flip(u): x0[c] = u[I-1-c], x1[c] = u[half-1-c].  new z = cat(x0, (x1 - m)*mask).
    v += G.bias[row];
    const int half = P.H, I2 = 2 * P.H;
    const long long bo = (long long)b * P.y_bstride + col;
    const float x1 = P.u[bo + (long long)(half - 1 - row) * P.Tout_stride];
    const float x0 = P.u[bo + (long long)(I2 - 1 - row) * P.Tout_stride];
    P.io[bo + (long long)(half + row) * P.Tout_stride] = col < lenb ? (x1 - v) : 0.f;
    P.io[bo + (long long)row * P.Tout_stride] = x0;
  }
}
// WN gate (commons.py:100-107): channel ch of batch b at column col from the tanh / sigmoid pre-activations
__device__ __forceinline__ void conv_epilogue_gate(const ConvParams& P, const ConvGroup& G, int b, int ch, int col, float at, float as) {
  if (ch >= P.H || col >= P.Tout) return;
  at += G.bias[ch];
  as += G.bias[P.H + ch];
  if (P.bias_b) {
    const float* bb = P.bias_b + (long long)b * P.bias_b_stride + P.bias_b_off;
    at += bb[ch];
    as += bb[P.H + ch];
  }
  const float tv = tanhf(at);
  const float sv = 1.0f / (1.0f + __expf(-as));
  G.y[(long long)b * P.y_bstride + (long long)ch * P.Tout_stride + col] = tv * sv;
}

template <int WM, int WN, int MI, int NI, int EPI>
__global__ void __launch_bounds__(256) conv_mfma_kernel(const ConvParams P) {
  static_assert(WM * WN == 4, "256-thread workgroups");
  constexpr int M_T = WM * MI * 32;
  constexpr int N_T = WN * NI * 32;
  constexpr int JT = (N_T + CONV_MAX_HALO + 63) / 64;
  extern __shared__ float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;

  // ---- block decode (bijective XCD remap: block L runs on XCD L%8; give each XCD a contiguous
  // range of logical ids so tiles sharing an activation window share an L2)
  int id;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, within = L >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = id % P.ntiles_m; id /= P.ntiles_m;
  const int grp = id % P.n_groups; id /= P.n_groups;
  const int nt = id % P.ntiles_n;
  const int b = id / P.ntiles_n;
  const ConvGroup& G = P.g[grp];

  const int ROW = P.row_len;
  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  int tap_base = 0;
  if (P.ups_u) tap_base = P.ups_shift[m0 / P.ups_cout];
  const int nchunks = P.Cin / CONV_CI_T;
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }

  // ---- staging: wave w owns chunk rows w, w+4, w+8, w+12; lanes stride over columns
  float stg[4][JT];
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float* xb2 = G.x2 ? G.x2 + (long long)b * P.x_bstride : nullptr;
  const float* xb3 = G.x3 ? G.x3 + (long long)b * P.x_bstride : nullptr;
  const float in_scale = P.in_scale, in_slope = P.in_slope;
  const int t_base = n0 - G.pad_l;

  auto load_chunk = [&](int c) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int ci = c * CONV_CI_T + wave + 4 * rr;
      const long long roff = (long long)(P.x_ch_off + ci * P.x_ch_sign) * P.Tin_stride;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        int t = t_base + col;
        if (P.reflect && t == -1) t = (P.Tin > 1) ? 1 : 0;
        float v = 0.f;
        if (col < ROW && t >= 0 && t < t_lim) {
          v = xb[roff + t];
          if (xb2) v += xb2[roff + t];
          if (xb3) v += xb3[roff + t];
          v = conv_act_in(v, in_scale, in_slope);
        }
        stg[rr][j] = v;
      }
    }
  };
  auto store_chunk = [&](int buf) {
    float* dst = lds + buf * (CONV_CI_T * ROW);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        if (col < ROW) dst[(wave + 4 * rr) * ROW + col] = stg[rr][j];
      }
    }
  };

  // ---- accumulators and weight stream
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int n_mblocks = P.M >> 5;
  const f32x4* wp[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + wm * MI + mi;
    if (mb >= n_mblocks) mb = 0;  // padded tile: compute on valid memory, never stored
    wp[mi] = reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64 + lane;
  }
  const int n_sg = G.n_sg;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  f32x4 a_cur[MI][2], a_nxt[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    a_cur[mi][0] = wp[mi][0];
    a_cur[mi][1] = wp[mi][64];
  }
  int sg = 2;  // next step-group to fetch

  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) load_chunk(c + 1);
    const float* lb = lds + (c & 1) * (CONV_CI_T * ROW) + h * ROW + wn * (NI * 32) + l31 + tap_base;
#pragma unroll 1
    for (int kk = 0; kk < K; ++kk) {
      if (sg < n_sg) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          a_nxt[mi][0] = wp[mi][(size_t)sg * 64];
          a_nxt[mi][1] = wp[mi][(size_t)(sg + 1) * 64];
        }
      }
      sg += 2;
      const float* lk = lb + kk * dil;
      float bv[8][NI];
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[p][ni] = lk[2 * p * ROW + ni * 32];
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mi][p >> 2][p & 3], bv[p][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        a_cur[mi][0] = a_nxt[mi][0];
        a_cur[mi][1] = a_nxt[mi][1];
      }
    }
    if (c + 1 < nchunks) store_chunk((c + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int lenb = (P.out_mask || EPI == EPI_RESSKIP || EPI == EPI_COUPLE) ? P.len[b] : 0x7fffffff;
  if (EPI == EPI_GATE) {
    // packed m-blocks alternate [tanh 32 rows | sigmoid 32 rows] of the same 32 channels
    const int j = (m0 >> 6) + wm;  // channel block of 32
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + wn * (NI * 32) + ni * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        conv_epilogue_gate(P, G, b, j * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col, acc[0][ni][e], acc[MI - 1][ni][e]);
    }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + wn * (NI * 32) + ni * 32 + l31;
      const int rbase = m0 + (wm * MI + mi) * 32 + 4 * h;
#pragma unroll
      for (int e = 0; e < 16; ++e) conv_epilogue_elem<EPI>(P, G, b, lenb, rbase + (e & 3) + 8 * (e >> 2), col, acc[mi][ni][e]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Small-N variant ("K-split"): one workgroup owns a (MI*32) x (NI*32) output tile and its 4 waves
// split the CONTRACTION (input-channel chunks c = wave, wave+4, ...).  Every wave stages its own
// chunks into a wave-private LDS region and streams its own weight fragments with a 3-deep register
// prefetch, so the main loop has no workgroup barrier and a single utterance (N = 50..2400 columns)
// still spreads over hundreds of waves, each pulling a distinct weight slab from L2/HBM.  The four
// partial accumulators are summed through LDS; wave w then finishes rows e in [4w, 4w+4) of every
// 32x32 fragment through the shared epilogue.
// ---------------------------------------------------------------------------------------------
template <int MI, int NI, int EPI>
__global__ void __launch_bounds__(256) conv_mfma_ks_kernel(const ConvParams P) {
  constexpr int M_T = MI * 32;
  constexpr int N_T = NI * 32;
  constexpr int JT = (N_T + CONV_MAX_HALO + 63) / 64;
  extern __shared__ float lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  int id;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, within = L >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int mt = id % P.ntiles_m; id /= P.ntiles_m;
  const int grp = id % P.n_groups; id /= P.n_groups;
  const int nt = id % P.ntiles_n;
  const int b = id / P.ntiles_n;
  const ConvGroup& G = P.g[grp];

  const int ROW = P.row_len;
  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  int tap_base = 0;
  if (P.ups_u) tap_base = P.ups_shift[m0 / P.ups_cout];
  const int nchunks = P.Cin / CONV_CI_T;
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }

  float stg[CONV_CI_T][JT];
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float* xb2 = G.x2 ? G.x2 + (long long)b * P.x_bstride : nullptr;
  const float* xb3 = G.x3 ? G.x3 + (long long)b * P.x_bstride : nullptr;
  const float in_scale = P.in_scale, in_slope = P.in_slope;
  const int t_base = n0 - G.pad_l;
  float* wlds = lds + wave * (2 * CONV_CI_T * ROW);  // wave-private double buffer

  auto load_chunk = [&](int c) {
#pragma unroll
    for (int rr = 0; rr < CONV_CI_T; ++rr) {
      const int ci = c * CONV_CI_T + rr;
      const long long roff = (long long)(P.x_ch_off + ci * P.x_ch_sign) * P.Tin_stride;
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        int t = t_base + col;
        if (P.reflect && t == -1) t = (P.Tin > 1) ? 1 : 0;
        float v = 0.f;
        if (col < ROW && t >= 0 && t < t_lim) {
          v = xb[roff + t];
          if (xb2) v += xb2[roff + t];
          if (xb3) v += xb3[roff + t];
          v = conv_act_in(v, in_scale, in_slope);
        }
        stg[rr][j] = v;
      }
    }
  };
  auto store_chunk = [&](int buf) {
    float* dst = wlds + buf * (CONV_CI_T * ROW);
#pragma unroll
    for (int rr = 0; rr < CONV_CI_T; ++rr) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        if (col < ROW) dst[rr * ROW + col] = stg[rr][j];
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int n_mblocks = P.M >> 5;
  const f32x4* wp[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + mi;
    if (mb >= n_mblocks) mb = 0;
    wp[mi] = reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64 + lane;
  }
  // this wave's taps in order: (c, kk) for c = wave, wave+4, ... ; tap -> first step-group 2*(c*K+kk)
  const int my_chunks = wave < nchunks ? (nchunks - wave + 3) / 4 : 0;
  const int my_taps = my_chunks * K;
  auto tap_sg = [&](int tp) { const int ci = tp / K, kk = tp - ci * K; return 2 * ((wave + 4 * ci) * K + kk); };
  f32x4 a0[MI][2], a1[MI][2], a2[MI][2];  // taps tp, tp+1, tp+2
  auto fetch = [&](f32x4 (&dst)[MI][2], int tp) {
    if (tp < my_taps) {
      const int sg = tap_sg(tp);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        dst[mi][0] = wp[mi][(size_t)sg * 64];
        dst[mi][1] = wp[mi][(size_t)(sg + 1) * 64];
      }
    }
  };
  fetch(a0, 0);
  fetch(a1, 1);
  if (my_chunks > 0) { load_chunk(wave); store_chunk(0); }
  int tp = 0;
  for (int ci = 0; ci < my_chunks; ++ci) {
    const int cn = wave + 4 * (ci + 1);
    if (ci + 1 < my_chunks) load_chunk(cn);
    const float* lb = wlds + (ci & 1) * (CONV_CI_T * ROW) + h * ROW + l31 + tap_base;
#pragma unroll 1
    for (int kk = 0; kk < K; ++kk, ++tp) {
      fetch(a2, tp + 2);
      const float* lk = lb + kk * dil;
      float bv[8][NI];
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[p][ni] = lk[2 * p * ROW + ni * 32];
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mi][p >> 2][p & 3], bv[p][ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        a0[mi][0] = a1[mi][0]; a0[mi][1] = a1[mi][1];
        a1[mi][0] = a2[mi][0]; a1[mi][1] = a2[mi][1];
      }
    }
    if (ci + 1 < my_chunks) store_chunk((ci + 1) & 1);
  }

  // ---- cross-wave reduction through LDS (staging regions are dead after the barrier)
  __syncthreads();
  float* red = lds;  // [wave][mi][ni][e][64]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(((wave * MI + mi) * NI + ni) * 16 + e) * 64 + lane] = acc[mi][ni][e];
  __syncthreads();
  const int lenb = (P.out_mask || EPI == EPI_RESSKIP || EPI == EPI_COUPLE) ? P.len[b] : 0x7fffffff;
  float sum[MI][NI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int e = 4 * wave + ee;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) a += red[(((w * MI + mi) * NI + ni) * 16 + e) * 64 + lane];
        sum[mi][ni][ee] = a;
      }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + ni * 32 + l31;
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
      const int e = 4 * wave + ee;
      const int rin = (e & 3) + 8 * (e >> 2) + 4 * h;  // row inside the 32-row fragment
      if (EPI == EPI_GATE) {
        conv_epilogue_gate(P, G, b, (m0 >> 6) * 32 + rin, col, sum[0][ni][ee], sum[MI - 1][ni][ee]);
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) conv_epilogue_elem<EPI>(P, G, b, lenb, m0 + mi * 32 + rin, col, sum[mi][ni][ee]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Host side: weight packing into MFMA-fragment order.
//   src(row, ci, kk) -> packed[((mb * n_sg + sg) * 64 + lane) * 4 + s4]
//   with step S = sg*4 + s4, chunk = S / (8K), kk = (S % 8K) / 8, p = S % 8,
//   ci = chunk*16 + 2p + (lane>>5), row = mb*32 + (lane&31).
// A (32x32x2 MFMA): lane l holds A[i = l&31][k = l>>5]; B: lane l holds B[k = l>>5][j = l&31].
// ---------------------------------------------------------------------------------------------
template <typename F>
static void pack_conv_weights(float* dst, int Mpad, int Cin, int K, F src /* float(int row,int ci,int kk) */) {
  const int n_sg = Cin / CONV_CI_T * 2 * K;
  for (int mb = 0; mb < Mpad / 32; ++mb)
    for (int sg = 0; sg < n_sg; ++sg)
      for (int lane = 0; lane < 64; ++lane)
        for (int s4 = 0; s4 < 4; ++s4) {
          const int S = sg * 4 + s4;
          const int chunk = S / (8 * K), within = S % (8 * K);
          const int kk = within / 8, p = within % 8;
          const int ci = chunk * CONV_CI_T + 2 * p + (lane >> 5);
          const int row = mb * 32 + (lane & 31);
          dst[(((size_t)mb * n_sg + sg) * 64 + lane) * 4 + s4] = src(row, ci, kk);
        }
}
