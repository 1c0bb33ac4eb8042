// kernels_misc.hip.h — the non-GEMM kernels of the VITS2 inference path (gfx950, wave64).
// All tensors are channel-major [B,C,T] fp32; threads map to consecutive t so every global
// access is coalesced along time.  Reference lines are cited per kernel
// (paths relative to /root/reference/training/vits2/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PI_F 3.14159265358979323846f

// ----------------------------------------------------------------------------- Philox normals
// Counter-based noise for the two randn draws of infer() (models.py:96, :1700) when the caller
// does not inject noise.  (The CPU checker under oracle/ restates the same definition.)
__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ inline float philox_normal(uint64_t seed, uint32_t stream, uint32_t row, uint32_t t) {
  uint32_t c[4] = {t, row, stream, 0};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  float u1 = ((float)c[0] + 0.5f) * (1.0f / 4294967296.0f);
  float u2 = ((float)c[1] + 0.5f) * (1.0f / 4294967296.0f);
  if (u1 < 1e-12f) u1 = 1e-12f;
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// Per-call scalars of a forward that is replayed as a captured hipGraph: the kernels that need them read this block from
// device memory instead of taking them by value, so ONE graph serves requests with different scales / seeds
// (vits_synthesize fast path, engine.hip).  Kernels take a nullable pointer: null = use the by-value argument.
struct SynthDev {
  float scales[3];            // [noise_scale, length_scale, noise_scale_w]  (onnx_export.py:62-64)
  float pcm_scale;            // Synth.synth_audio's `scale` (vosk_tts/synth.py:128) for the int16 output
  unsigned long long seed;    // Philox seed
};

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ----------------------------------------------------------------------------- small utilities
__global__ void lengths_to_i32_kernel(const int64_t* in, int* out, int n, int clamp_max) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    long long v = in[i];
    if (v < 0) v = 0;
    if (v > clamp_max) v = clamp_max;
    out[i] = (int)v;
  }
}

// x[b,c,t] = emb[ids[b,t]][c] * sqrt(H) * mask  (models.py:318-322).  err[0] set on bad id.
__global__ void embed_kernel(const int64_t* ids, const int* len, const float* emb, float* x, int H, int T, int n_vocab,
                             float scale, int* err) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.z;
  if (t >= T) return;
  long long id = ids[(long long)b * T + t];
  const bool valid = t < len[b];
  if (valid && (id < 0 || id >= n_vocab)) { atomicOr(err, 1); id = 0; }
  for (int c = blockIdx.y; c < H; c += gridDim.y)
    x[((long long)b * H + c) * T + t] = valid ? emb[id * H + c] * scale : 0.f;
}

// out[b][r] = bias[r] + sum_j W[r][j] * emb_g[sid[b]][j]   — every cond(g) / Linear(g) on the path
// in one launch: enc spk_emb_linear (attentions.py:52-56), dp.cond (models.py:60), WN cond_layer
// (modules.py:152-153).  W is the row-concatenation built at load time.  One wave per row.
// len64 / len32 (optional): the int64 -> clamped int32 conversion of the feed's lengths rides along (one launch less)
__global__ void cond_gemv_kernel(const float* W, const float* bias, const float* emb_g, const int64_t* sid, float* out,
                                 int rows, int G, int n_speakers, int* err, const int64_t* len64, int* len32, int clamp_max) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), b = blockIdx.y;
  if (len64 && blockIdx.x == 0 && threadIdx.x == 0) {
    long long v = len64[b];
    v = v < 0 ? 0 : (v > clamp_max ? clamp_max : v);
    len32[b] = (int)v;
  }
  if (row >= rows) return;
  long long s = sid ? sid[b] : 0;
  if (s < 0 || s >= n_speakers) { if (lane == 0) atomicOr(err, 2); s = 0; }
  const float* g = emb_g + s * G;
  float a = 0.f;
  for (int j = lane; j < G; j += 64) a += W[(long long)row * G + j] * g[j];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if (lane == 0) out[(long long)b * rows + row] = a + bias[row];
}

// x = (x + v[b][c]) * mask   (attentions.py:55-56)
__global__ void add_vec_mask_kernel(float* x, const float* v, int v_stride, int v_off, const int* len, int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const long long o = ((long long)b * C + c) * T + t;
  x[o] = t < len[b] ? x[o] + v[(long long)b * v_stride + v_off + c] : 0.f;
}

// ----------------------------------------------------------------------------- LayerNorm over C
// modules.LayerNorm (modules.py:29-32): y = LN_c(a [+ b]) ; optional GELU ; optional out = base + y ;
// optional mask.  Block = 256 threads = 16 time lanes x 16 channel groups: every thread keeps its
// C/16 channel values in registers (one coalesced 64-byte segment per 16 lanes per channel), the
// 16 groups combine through LDS (two-pass mean / variance, like F.layer_norm).
#define LN_TL 16
#define LN_CG 16
#define LN_MAXV 24  // C <= 384
struct LNParams {
  const float* a; const float* b; const float* base; float* y;
  const float* gamma; const float* beta; const int* len;
  int C, T; int gelu; int mask;
  int skip_len;  // ragged batch: blocks that start at or beyond len[b] do nothing
  // adaLN (DiT blocks, stabletts diffusion_transformer.py:111,120-122): when > 0 the affine part is per batch item,
  // y = LN(x) * (1 + gamma[b*mod_stride + c]) + beta[b*mod_stride + c]   (LayerNorm without elementwise affine + modulate)
  int mod_stride;
  float eps;  // 1e-5 (modules.LayerNorm / nn.LayerNorm default), 1e-12 for the BERT encoder
  // FiLM folded in front (stabletts DitWrapper, components/decoder.py:15-16,31-33): when pre != nullptr the input is first
  // mapped to x' = (pre[c] * x + pre[C + c]) * [t < len[b]], x' is written to pre_out (the block's residual stream) and the
  // LayerNorm runs on x'
  const float* pre; float* pre_out;
  // K-sliced producer (round 5, the BERT encoder's second FFN matrix: stts.hip.h bert_forward): `a` is the first of np partial tensors
  // pstride floats apart, summed here in fixed order before everything else (0 / 1: a is the whole tensor)
  int np; long long pstride;
};
template <int TL = LN_TL>
__device__ __forceinline__ float ln_group_sum(float v, float* red, int tl, int cg) {
  constexpr int CG = 256 / TL;
  red[cg * TL + tl] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < CG; ++g) s += red[g * TL + tl];
  __syncthreads();
  return s;
}
// MAXV = channels per thread (C <= CG * MAXV): instantiated for 12 / 24 / 48 so narrow tensors do not issue dead loads.
// TL = time lanes per block (CG = 256 / TL channel groups): 16 by default; 4 for tensors of a few columns (round 5: a 768 x 16 BERT
// tensor was ONE block, a 384 x 304 estimator tensor 20 -- 9 us of latency per launch, 21 / 68 launches per request)
template <int MAXV, int TL = LN_TL>
__global__ void __launch_bounds__(256) layernorm_c_kernel(const LNParams P) {
  constexpr int LN_CGT = 256 / TL;
  __shared__ float red[256];
  kernarg_warm<sizeof(LNParams)>();
  const int tl = threadIdx.x & (TL - 1), cg = threadIdx.x / TL, b = blockIdx.y;
  if (P.skip_len && (int)(blockIdx.x * TL) >= P.len[b]) return;  // block-uniform
  const int t = blockIdx.x * TL + tl;
  const bool in = t < P.T;
  const long long o0 = (long long)b * P.C * P.T + (in ? t : 0);
  float v[MAXV];
  float sum = 0.f;
  // every load unconditional with a clamped channel index (validity is a select afterwards): a per-element `if` around
  // the load makes hipcc wait for each one in turn -- 12..24 dependent L2 round trips instead of one
  if (P.b) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = cg + i * LN_CGT, cc = c < P.C ? c : P.C - 1;
      v[i] = P.a[o0 + (long long)cc * P.T] + P.b[o0 + (long long)cc * P.T];
    }
  } else {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = cg + i * LN_CGT, cc = c < P.C ? c : P.C - 1;
      v[i] = P.a[o0 + (long long)cc * P.T];
    }
  }
  if (P.np > 1) {  // kernel-uniform
    for (int j = 1; j < P.np; ++j) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int c = cg + i * LN_CGT, cc = c < P.C ? c : P.C - 1;
        v[i] += P.a[(long long)j * P.pstride + o0 + (long long)cc * P.T];
      }
    }
  }
  if (P.pre) {
    const bool live = in && t < P.len[b];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = cg + i * LN_CGT, cc = c < P.C ? c : P.C - 1;
      v[i] = live ? P.pre[cc] * v[i] + P.pre[P.C + cc] : 0.f;
      if (in && c < P.C) P.pre_out[o0 + (long long)c * P.T] = v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = cg + i * LN_CGT;
    v[i] = (in && c < P.C) ? v[i] : 0.f;
    sum += v[i];
  }
  const float mean = ln_group_sum<TL>(sum, red, tl, cg) / (float)P.C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = cg + i * LN_CGT;
    if (c < P.C) { const float d = v[i] - mean; sq += d * d; }
  }
  const float rstd = 1.0f / sqrtf(ln_group_sum<TL>(sq, red, tl, cg) / (float)P.C + P.eps);
  if (!in) return;
  const bool zero = P.mask && t >= P.len[b];
  const float* gp = P.gamma + (P.mod_stride ? (long long)b * P.mod_stride : 0);
  const float* bp = P.beta + (P.mod_stride ? (long long)b * P.mod_stride : 0);
  const float g1 = P.mod_stride ? 1.0f : 0.0f;  // adaLN: 1 + scale
  float ga[MAXV], be[MAXV], ba[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = cg + i * LN_CGT, cc = c < P.C ? c : P.C - 1;
    ga[i] = gp[cc]; be[i] = bp[cc];
    ba[i] = P.base ? P.base[o0 + (long long)cc * P.T] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = cg + i * LN_CGT;
    float x = (v[i] - mean) * rstd * (g1 + ga[i]) + be[i];
    if (P.gelu) x = gelu_erf(x);
    x += ba[i];
    if (c < P.C) P.y[o0 + (long long)c * P.T] = zero ? 0.f : x;
  }
}

static void launch_layernorm(hipStream_t st, const LNParams& P, int B) {
  static const int small_blocks = getenv("VITS_LN_SMALL") ? atoi(getenv("VITS_LN_SMALL")) : 64;  // A/B: 0 = never the 4-lane form
  if ((long)((P.T + LN_TL - 1) / LN_TL) * B < small_blocks && P.C <= 12 * 64) {  // few columns: 4 time lanes x 64 channel groups, 4 x the blocks
    const dim3 g4((P.T + 3) / 4, B);
    if (P.C <= 6 * 64) hipLaunchKernelGGL((layernorm_c_kernel<6, 4>), g4, dim3(256), 0, st, P);
    else hipLaunchKernelGGL((layernorm_c_kernel<12, 4>), g4, dim3(256), 0, st, P);
    return;
  }
  const dim3 grid((P.T + LN_TL - 1) / LN_TL, B);
  if (P.C <= 12 * LN_CG) hipLaunchKernelGGL(layernorm_c_kernel<12>, grid, dim3(256), 0, st, P);
  else if (P.C <= 24 * LN_CG) hipLaunchKernelGGL(layernorm_c_kernel<24>, grid, dim3(256), 0, st, P);
  else hipLaunchKernelGGL(layernorm_c_kernel<48>, grid, dim3(256), 0, st, P);
}

// DDSConv first half (modules.py:100-102): y = gelu(LN1(dwconv_k,dil(x * mask))), same block shape;
// the depthwise conv result stays in registers between the statistics and the normalisation.
struct DwLnParams {
  const float* x; float* y; const float* w; const float* bias; const float* gamma; const float* beta; const int* len;
  int C, T, K, dil;
  int skip_len;
};
__global__ void __launch_bounds__(256) dwconv_ln_gelu_kernel(const DwLnParams P) {
  __shared__ float red[LN_CG * LN_TL];
  const int tl = threadIdx.x & (LN_TL - 1), cg = threadIdx.x >> 4, b = blockIdx.y;
  if (P.skip_len && (int)(blockIdx.x * LN_TL) >= P.len[b]) return;  // block-uniform
  const int t = blockIdx.x * LN_TL + tl;
  const bool in = t < P.T;
  const int L = P.len[b] < P.T ? P.len[b] : P.T, pad = (P.K * P.dil - P.dil) / 2;
  const long long o0 = (long long)b * P.C * P.T;
  float v[LN_MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = cg + i * LN_CG;
    float a = 0.f;
    if (in && c < P.C) {
      a = P.bias[c];
      for (int k = 0; k < P.K; ++k) {
        const int s = t + k * P.dil - pad;
        if (s >= 0 && s < L) a += P.w[c * P.K + k] * P.x[o0 + (long long)c * P.T + s];
      }
    }
    v[i] = a;
    sum += a;
  }
  const float mean = ln_group_sum(sum, red, tl, cg) / (float)P.C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = cg + i * LN_CG;
    if (c < P.C) { const float d = v[i] - mean; sq += d * d; }
  }
  const float rstd = 1.0f / sqrtf(ln_group_sum(sq, red, tl, cg) / (float)P.C + 1e-5f);
  if (!in) return;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = cg + i * LN_CG;
    if (c < P.C) P.y[o0 + (long long)c * P.T + t] = gelu_erf((v[i] - mean) * rstd * P.gamma[c] + P.beta[c]);
  }
}

// One whole DDSConv layer (modules.py:96-108) per launch for D <= 256 channels:
//   y = conv_sep(x * mask) (depthwise, K taps, dilation K^i) -> LN1 -> GELU -> conv_1x1 -> LN2 -> GELU ; x = (x + y) * mask
// A workgroup owns DDS_TL columns and ALL channels (thread = channel), so both channel LayerNorms are block
// reductions and the 1x1 conv is a [D x D] mat-vec per column on the VALU: thread co walks the TRANSPOSED weight
// matrix wt[ci][co] (coalesced) against the activations broadcast from LDS.  At these sizes (T_x tokens, D = 256:
// 3.3 MFLOP per layer) the three launches this replaces were pure launch latency.  Input and output buffers differ
// (the depthwise taps read neighbouring columns owned by other workgroups).
#define DDS_TL 8
struct DdsParams {
  const float* x; float* y;
  const float* sw; const float* sb; const float* g1; const float* b1;
  const float* wt; const float* pb; const float* g2; const float* b2;
  const int* len;
  int D, T, K, dil, skip_len;
};
__device__ __forceinline__ void dds_block_sum(float (&v)[DDS_TL], float* red, int lane, int wave) {
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[j] += __shfl_xor(v[j], off);
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < DDS_TL; ++j) red[wave * DDS_TL + j] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) v[j] = red[j] + red[DDS_TL + j] + red[2 * DDS_TL + j] + red[3 * DDS_TL + j];
  __syncthreads();
}
__device__ __forceinline__ void dds_ln_gelu(float (&v)[DDS_TL], bool live, float gamma, float beta, float invD, float* red, int lane, int wave) {
  float s[DDS_TL];
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) s[j] = live ? v[j] : 0.f;
  dds_block_sum(s, red, lane, wave);
  float q[DDS_TL];
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) { s[j] *= invD; const float d = v[j] - s[j]; q[j] = live ? d * d : 0.f; }
  dds_block_sum(q, red, lane, wave);
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) v[j] = gelu_erf((v[j] - s[j]) * (1.0f / sqrtf(q[j] * invD + 1e-5f)) * gamma + beta);
}
// weights of the 1x1 conv arrive as wt4[ci/4][co] = float4{W[co][ci..ci+3]}: one dwordx4 per thread per 4 input
// channels.  The whole matrix (D*D*4 bytes, 256 KB at D = 256) is streamed by EVERY workgroup and comes from HBM / MALL
// on first touch, so the stream is software-pipelined in two register batches of 16 float4 (64 input channels): batch 0
// is issued before the depthwise/LayerNorm phase, batch k+1 while batch k is consumed.
#define DDS_WB 16
__device__ __forceinline__ void dds_load_batch(float4 (&w)[DDS_WB], const float4* wt4, int D, int k, int co) {
#pragma unroll
  for (int i = 0; i < DDS_WB; ++i) w[i] = wt4[(size_t)(k * DDS_WB + i) * D + co];
}
__device__ __forceinline__ void dds_use_batch(const float4 (&w)[DDS_WB], const float* act, int k, float (&acc)[DDS_TL]) {
#pragma unroll
  for (int i = 0; i < DDS_WB; ++i) {
    const float wv[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* ap = act + ((k * DDS_WB + i) * 4 + q) * DDS_TL;
      const float4 a0 = *reinterpret_cast<const float4*>(ap);
      const float4 a1 = *reinterpret_cast<const float4*>(ap + 4);
      acc[0] += wv[q] * a0.x; acc[1] += wv[q] * a0.y; acc[2] += wv[q] * a0.z; acc[3] += wv[q] * a0.w;
      acc[4] += wv[q] * a1.x; acc[5] += wv[q] * a1.y; acc[6] += wv[q] * a1.z; acc[7] += wv[q] * a1.w;
    }
  }
}
__global__ void __launch_bounds__(256) dds_layer_kernel(const DdsParams P) {
  __shared__ __attribute__((aligned(16))) float act[256 * DDS_TL];
  __shared__ float red[4 * DDS_TL];
  const int c = threadIdx.x, lane = c & 63, wave = c >> 6, b = blockIdx.y, t0 = blockIdx.x * DDS_TL;
  const int L = P.len[b] < P.T ? P.len[b] : P.T;
  if (P.skip_len && t0 >= L) return;  // block-uniform: the whole tile is padding of a masked stage
  const int D = P.D, T = P.T, pad = (P.K * P.dil - P.dil) / 2;
  const bool live = c < D;
  const int cc = live ? c : 0;
  const float invD = 1.0f / (float)D;
  const int nb = D / (4 * DDS_WB);  // weight batches of 64 input channels (D is a multiple of 64, <= 256)
  const float4* wt4 = reinterpret_cast<const float4*>(P.wt);
  float4 wa[DDS_WB], wb[DDS_WB];
  dds_load_batch(wa, wt4, D, 0, cc);
  const float* xr = P.x + ((long long)b * D + cc) * T;
  float v[DDS_TL];
  {
    // every load unconditional with a clamped index, validity applied as a select: a per-element `if` around the load
    // would serialise the K * DDS_TL row reads behind s_waitcnt vmcnt(0)
    const float bias = P.sb[cc];
#pragma unroll
    for (int j = 0; j < DDS_TL; ++j) v[j] = bias;
    const int Lc = L > 0 ? L - 1 : 0;
    for (int k = 0; k < P.K; ++k) {
      const float wk = P.sw[cc * P.K + k];
      float xv[DDS_TL];
#pragma unroll
      for (int j = 0; j < DDS_TL; ++j) {
        const int s = t0 + j + k * P.dil - pad;
        xv[j] = xr[s < 0 ? 0 : (s > Lc ? Lc : s)];
      }
#pragma unroll
      for (int j = 0; j < DDS_TL; ++j) {
        const int s = t0 + j + k * P.dil - pad;
        v[j] += (s >= 0 && s < L) ? wk * xv[j] : 0.f;
      }
    }
  }
  const float g2c = P.g2[cc], b2c = P.b2[cc];
  float xres[DDS_TL];  // residual input, fetched early
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) { const int t = t0 + j; xres[j] = xr[t < T ? t : T - 1]; }
  dds_ln_gelu(v, live, P.g1[cc], P.b1[cc], invD, red, lane, wave);
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) act[c * DDS_TL + j] = live ? v[j] : 0.f;
  __syncthreads();
  float acc[DDS_TL];
  {
    const float pb = P.pb[cc];
#pragma unroll
    for (int j = 0; j < DDS_TL; ++j) acc[j] = pb;
  }
  if (nb > 1) dds_load_batch(wb, wt4, D, 1, cc);
  dds_use_batch(wa, act, 0, acc);
  if (nb > 2) dds_load_batch(wa, wt4, D, 2, cc);
  if (nb > 1) dds_use_batch(wb, act, 1, acc);
  if (nb > 3) dds_load_batch(wb, wt4, D, 3, cc);
  if (nb > 2) dds_use_batch(wa, act, 2, acc);
  if (nb > 3) dds_use_batch(wb, act, 3, acc);
  dds_ln_gelu(acc, live, g2c, b2c, invD, red, lane, wave);
  if (!live) return;
  float* yr = P.y + ((long long)b * D + c) * T;
#pragma unroll
  for (int j = 0; j < DDS_TL; ++j) {
    const int t = t0 + j;
    if (t < T) yr[t] = t < L ? xres[j] + acc[j] : 0.f;
  }
}

// ----------------------------------------------------------------------------- attention
// MultiHeadAttention.attention (attentions.py:165-196) with the relative-position key/value terms
// (attentions.py:198-260) in exact banded form: O(T) memory, no [T,2T-1] skew.
//   s[i,j] = q~_i.k_j + (|j-i|<=W ? q~_i.E_k[j-i+W] : 0),  masked keys (-1e4) carry weight exp(-1e4-m) == 0
//   out_i  = sum_j p_ij v_j + sum_{|j-i|<=W} p_ij E_v[j-i+W]
// Block = 256 threads = 16 queries x 16 key lanes; K/V tiles of 64 keys staged in LDS (coalesced
// along time), q in registers, online softmax per query with 16-lane shuffles, fp32 throughout.
// Rows past len[b] are written as 0 (the reference's uniform-softmax junk there is multiplied by
// x_mask before it can reach any valid position; see DESIGN.md "masked rows").
// qkv: [B, 3H, T] (q rows [0,H), k rows [H,2H), v rows [2H,3H)), out [B,H,T].
#define ATT_TQ 16
#define ATT_KL 16
#define ATT_TK 64
template <int DK>
__global__ void __launch_bounds__(256) relpos_attention_kernel(const float* qkv, const float* ek, const float* ev,
                                                                const int* len, float* out, int H, int T, int W) {
  __shared__ float kt[DK * ATT_TK];
  __shared__ float vt[DK * ATT_TK];
  const int tid = threadIdx.x, kl = tid & (ATT_KL - 1), qi = tid >> 4;
  const int hd = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * ATT_TQ + qi;
  const int L = len[b] < T ? len[b] : T;
  const float* qb = qkv + ((long long)b * 3 * H + (long long)hd * DK) * T;
  const float* kb = qb + (long long)H * T;
  const float* vb = kb + (long long)H * T;
  const bool active = i < L;
  const int ic = active ? i : 0;
  const float scale = 1.0f / sqrtf((float)DK);
  float q[DK];
#pragma unroll
  for (int d = 0; d < DK; ++d) q[d] = qb[(long long)d * T + ic] * scale;
  // relative-key logits: lane kl < 2W+1 holds q~ . E_k[kl]
  float qe = 0.f;
  if (kl <= 2 * W) {
#pragma unroll
    for (int d = 0; d < DK; ++d) qe += q[d] * ek[kl * DK + d];
  }
  float acc[DK];
#pragma unroll
  for (int d = 0; d < DK; ++d) acc[d] = 0.f;
  float m = -3.0e38f, l = 0.f;
  for (int j0 = 0; j0 < L; j0 += ATT_TK) {
    __syncthreads();
    for (int e = tid; e < DK * ATT_TK; e += 256) {
      const int d = e / ATT_TK, jj = e % ATT_TK;
      const int j = j0 + jj;
      const bool ok = j < L;
      kt[e] = ok ? kb[(long long)d * T + j] : 0.f;
      vt[e] = ok ? vb[(long long)d * T + j] : 0.f;
    }
    __syncthreads();
    float s[ATT_TK / ATT_KL];
    float tmax = -3.0e38f;
#pragma unroll
    for (int mm = 0; mm < ATT_TK / ATT_KL; ++mm) {
      const int jj = kl + ATT_KL * mm;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DK; ++d) a += q[d] * kt[d * ATT_TK + jj];
      const int r = j0 + jj - i;
      const int rc = r + W < 0 ? 0 : (r + W > 15 ? 15 : r + W);
      const float bias = __shfl(qe, rc, ATT_KL);  // executed by all lanes
      if (r >= -W && r <= W) a += bias;
      if (j0 + jj >= L) a = -3.0e38f;
      s[mm] = a;
      tmax = fmaxf(tmax, a);
    }
#pragma unroll
    for (int o = ATT_KL / 2; o > 0; o >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, o, ATT_KL));
    const float mn = fmaxf(m, tmax);
    const float alpha = __expf(m - mn);
    l *= alpha;
#pragma unroll
    for (int d = 0; d < DK; ++d) acc[d] *= alpha;
#pragma unroll
    for (int mm = 0; mm < ATT_TK / ATT_KL; ++mm) {
      const int jj = kl + ATT_KL * mm;
      const float p = (j0 + jj < L) ? __expf(s[mm] - mn) : 0.f;
      l += p;
#pragma unroll
      for (int d = 0; d < DK; ++d) acc[d] += p * vt[d * ATT_TK + jj];
    }
    m = mn;
  }
  // relative-value band (attentions.py:191-194): lane kl < 2W+1 owns offset r = kl - W; its probability
  // is recomputed against the final max and folded into this lane's partial accumulator
  if (active && kl <= 2 * W) {
    const int j = i + kl - W;
    if (j >= 0 && j < L) {
      float a = qe;
#pragma unroll
      for (int d = 0; d < DK; ++d) a += q[d] * kb[(long long)d * T + j];
      const float p = __expf(a - m);
#pragma unroll
      for (int d = 0; d < DK; ++d) acc[d] += p * ev[kl * DK + d];
    }
  }
#pragma unroll
  for (int o = ATT_KL / 2; o > 0; o >>= 1) l += __shfl_xor(l, o, ATT_KL);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  // reduce the 16 partial accumulators of each query; lane kl writes channels d == kl (mod 16)
  float* ob = out + ((long long)b * H + (long long)hd * DK) * T + i;
#pragma unroll
  for (int d = 0; d < DK; ++d) {
    float a = acc[d];
#pragma unroll
    for (int o = ATT_KL / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, ATT_KL);
    if ((d & (ATT_KL - 1)) == kl && i < T) ob[(long long)d * T] = active ? a * inv : 0.f;
  }
}

// ----------------------------------------------------------------------------- attention (MFMA)
// Same math as relpos_attention_kernel, on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), flash style.
// One workgroup = 32 queries of one (batch, head); its 4 waves split the KEY tiles (jt = wave, wave+4, ..)
// and merge their (m, l, O) through LDS at the end, so a single short utterance still uses every wave
// and long-form (T_y = 6000) has ~47 tiles per wave.
//
// "Swapped" formulation — everything is computed transposed so that a LANE owns a QUERY:
//   S^T[key][q] = sum_d K[key][d] Q^T[d][q]        A = K fragment (coalesced from [d][T]), B = Q^T (registers)
//   C/D layout: lane (h = lane>>5, l31 = lane&31) register e holds row kappa(e,h) = (e&3)+8(e>>2)+4h, column l31
//   -> lane l31 holds 16 of the 32 scores of ITS query; row max / sum are in-register + one xor-32 exchange.
//   O^T[d][q] += sum_key V[d][key] P^T[key][q]      the contraction order of an MFMA is free, so k-step s is
//   DEFINED to cover keys kappa(s,0), kappa(s,1): then the B fragment of step s is exactly accumulator
//   register s of S^T — P never moves.  A = V fragment read from a padded LDS tile.
//   relative keys:   QE^T[r][q] = E_k[r] . q~  by one MFMA pass, kept in LDS [r][q]
//   relative values: O^T[d][q] += sum_r E_v^T[d][r] Prel^T[r][q], Prel gathered through LDS on the <= 3 tiles
//   that touch the diagonal band.
template <int DK>
__global__ void __launch_bounds__(256, 2) relpos_attention_mfma_kernel(const float* qkv, const float* ek, const float* ev,
                                                                     const int* len, float* out, int H, int T, int W) {
  constexpr int NS = DK / 2, ND = DK / 32, VS = 33;
  constexpr int WREG = DK * VS + 10 * 32 + 9 * 32;  // per-wave LDS: V tile | Prel | QE
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, h = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hd = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * 32;
  const int L = len[b] < T ? len[b] : T;
  const int i = i0 + l31;
  float* ob = out + ((long long)b * H + (long long)hd * DK) * T;
  if (i0 >= L) {  // whole query tile is padding: write zeros (see "masked rows" in DESIGN.md)
    if (i < T)
      for (int d = (threadIdx.x >> 5); d < DK; d += 8) ob[(long long)d * T + i] = 0.f;
    return;
  }
  const bool active = i < L;
  const int ic = active ? i : L - 1;
  const float* qb = qkv + ((long long)b * 3 * H + (long long)hd * DK) * T;
  const float* kb = qb + (long long)H * T;
  const float* vb = kb + (long long)H * T;
  float* vt = lds + wave * WREG;
  float* prel = vt + DK * VS;
  float* qes = prel + 10 * 32;
  const float scale = 1.0f / sqrtf((float)DK);
  const int NW = 2 * W + 1;

  // B operand of every QK^T step: Q^T[d = 2s + h][query], pre-scaled
  float qf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) qf[s] = qb[(long long)(2 * s + h) * T + ic] * scale;
  // ek == nullptr: plain scaled-dot-product attention (StableTTS DiT blocks, BERT) -- no relative-position work at all
  const bool rel = ek != nullptr;
  // QE^T[r][q] (attentions.py:175-177): A = E_k[r = l31][d = 2s + h]
  if (rel) {
    f32x16 qe;
#pragma unroll
    for (int e = 0; e < 16; ++e) qe[e] = 0.f;
    const int rr = l31 < NW ? l31 : 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float a = l31 < NW ? ek[rr * DK + 2 * s + h] : 0.f;
      qe = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qf[s], qe, 0, 0, 0);
    }
    // rows r = kappa(e,h) < 9: h=0 -> e 0..3 (r 0..3) and e 4 (r 8); h=1 -> e 0..3 (r 4..7)
#pragma unroll
    for (int e = 0; e < 5; ++e) {
      const int r = (e & 3) + 8 * (e >> 2) + 4 * h;
      if (r < 9) qes[r * 32 + l31] = qe[e];
    }
  }

  f32x16 O[ND];
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
  float m = -3.0e38f, l = 0.f;

  const int ntiles = (L + 31) >> 5;
  // K fragments of the current tile live in registers and are prefetched one tile ahead (issued before the
  // PV MFMAs of the previous tile); the V tile's loads are issued at the top of the tile and fly under the
  // 48 QK^T MFMAs before they are written to LDS.
  float kf[NS];
  auto load_k = [&](int jt_) {
    const int j = jt_ * 32 + l31;
    const int jc = j < L ? j : L - 1;
    const float* kp = kb + (long long)h * T + jc;
#pragma unroll
    for (int s = 0; s < NS; ++s) kf[s] = kp[(long long)(2 * s) * T];
  };
  if (wave < ntiles) load_k(wave);
  for (int jt = wave; jt < ntiles; jt += 4) {
    const int j0 = jt * 32;
    // ---- V tile: issue the loads now (keys beyond L read as 0) ...
    float vst[NS];
    const bool vok = j0 + l31 < L;
    {
      const int jc = vok ? j0 + l31 : L - 1;
      const float* vp = vb + (long long)h * T + jc;
#pragma unroll
      for (int s = 0; s < NS; ++s) vst[s] = vp[(long long)(2 * s) * T];
    }
    // ---- S^T = K Q^T on the prefetched K fragments
    f32x16 S;
#pragma unroll
    for (int e = 0; e < 16; ++e) S[e] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], S, 0, 0, 0);
    // ... and park V in the wave-private LDS tile [d][33] once QK^T has been issued
#pragma unroll
    for (int s = 0; s < NS; ++s) vt[(2 * s + h) * VS + l31] = vok ? vst[s] : 0.f;
    // next tile's K fragments fly under the softmax and the PV MFMAs
    if (jt + 4 < ntiles) load_k(jt + 4);
    // ---- relative-key bias on the diagonal band, key mask, tile max
    const bool near = rel && (j0 + 31 >= i0 - W) && (j0 <= i0 + 31 + W);
    float mx = -3.0e38f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      float sv = S[e];
      if (near) {
        const int r = key - i + W;
        const int rc = r < 0 ? 0 : (r > 8 ? 8 : r);
        const float bias = qes[rc * 32 + l31];
        sv += (r >= 0 && r < NW) ? bias : 0.f;
      }
      sv = key < L ? sv : -3.0e38f;
      S[e] = sv;
      mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const float pe = key < L ? __expf(S[e] - mn) : 0.f;
      S[e] = pe;  // S now holds P^T
      psum += pe;
    }
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) O[dt][e] *= alpha;
    // ---- O^T += V P^T  (k-step s <-> keys kappa(s,0), kappa(s,1): B fragment == S[s])
#pragma unroll
    for (int dt = 0; dt < ND; ++dt) {
      const float* vrow = vt + (dt * 32 + l31) * VS + 4 * h;
#pragma unroll
      for (int s = 0; s < 16; ++s) O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[(s & 3) + 8 * (s >> 2)], S[s], O[dt], 0, 0, 0);
    }
    // ---- relative values (attentions.py:191-194) on band tiles: Prel^T[r][q] through LDS, then 5 k-steps
    if (near) {
#pragma unroll
      for (int r5 = 0; r5 < 5; ++r5) prel[(5 * h + r5) * 32 + l31] = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = j0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int r = key - i + W;
        if (r >= 0 && r < NW && key < L) prel[r * 32 + l31] = S[e];
      }
#pragma unroll
      for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) {
          const int r = 2 * s5 + h;
          const float a = r < NW ? ev[(r < NW ? r : 0) * DK + dt * 32 + l31] : 0.f;
          O[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, prel[r * 32 + l31], O[dt], 0, 0, 0);
        }
    }
  }

  // ---- merge the 4 waves' partial (m, l, O) through LDS; wave w finishes values idx == w (mod 4)
  __syncthreads();
  constexpr int NV = ND * 16 + 2;
  float* comb = lds;  // [wave][NV][64]
  {
    float* c = comb + (wave * NV) * 64 + lane;
    c[0] = m;
    c[64] = l;
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) c[(2 + dt * 16 + e) * 64] = O[dt][e];
  }
  __syncthreads();
  float mw[4], sc[4];
  float ms = -3.0e38f;
#pragma unroll
  for (int w = 0; w < 4; ++w) { mw[w] = comb[(w * NV) * 64 + lane]; ms = fmaxf(ms, mw[w]); }
  float lt = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) { sc[w] = __expf(mw[w] - ms); lt += comb[(w * NV + 1) * 64 + lane] * sc[w]; }
  lt += __shfl_xor(lt, 32, 64);
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
#pragma unroll
  for (int k = 0; k < ND * 4; ++k) {
    const int idx = 4 * k + wave;  // dt*16 + e
    const int dt = idx >> 4, e = idx & 15;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) a += comb[(w * NV + 2 + idx) * 64 + lane] * sc[w];
    const int d = dt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (i < T) ob[(long long)d * T + i] = active ? a * inv : 0.f;
  }
}

// ----------------------------------------------------------------------------- duration predictor
// z[b,c,t] = noise * noise_scale_w  (models.py:96)
// solo != 0 (VITS_FLAG_SOLO_BATCH): item b draws what a single-utterance call with seed + b would draw
// item_seeds (optional, solo batches): per-item seeds instead of seed + b (requests batched by a server keep their own draw)
__global__ void dp_init_z_kernel(float* z, const float* noise, float nsw, uint64_t seed, int T, int solo, const SynthDev* dv,
                                 const unsigned long long* item_seeds) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  if (dv) { nsw = dv->scales[2]; seed = dv->seed; }
  const long long o = ((long long)b * 2 + c) * T + t;
  const float e = noise ? noise[o] : (solo ? philox_normal(item_seeds ? item_seeds[b] : seed + (uint64_t)b, 1, (uint32_t)c, (uint32_t)t)
                                            : philox_normal(seed, 1, (uint32_t)(b * 2 + c), (uint32_t)t));
  z[o] = e * nsw;
}

// ConvFlow head (modules.py:365-366): h = pre(x0) + g   (Conv1d(1,D,1) is a per-channel affine; the
// DDSConv that follows starts with x = x + g, modules.py:97-98)
__global__ void convflow_pre_kernel(const float* z, int x0_row, const float* pw, const float* pb, const float* cond,
                                    float* h, int D, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const long long o = ((long long)b * D + c) * T + t;
  h[o] = pw[c] * z[((long long)b * 2 + x0_row) * T + t] + pb[c] + cond[o];
}

__device__ __forceinline__ float softplus_f(float v) { return v > 20.f ? v : log1pf(expf(v)); }

// Inverse rational-quadratic spline with linear tails for ONE element (transforms.py:55-177, inverse branch 152-167;
// searchsorted :47-52).  pp(i) = row i of proj(h)*mask at this element's column (3*nb - 1 rows).  Shared by the launch path
// (spline_inverse_kernel) and the persistent duration-predictor kernel (persist.hip.h) so both run the same arithmetic.
template <typename PP>
__device__ __forceinline__ float spline_inverse_elem(float y, PP pp, int nb, float bound, float inv_sqrt_d) {
  if (!(y >= -bound && y <= bound)) return y;  // identity outside the interval (transforms.py:65-77)
  const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
  constexpr int NB = 16;
  float w[NB], cw[NB + 1], hh[NB], ch[NB + 1];
  float mx = -3.0e38f;
  for (int i = 0; i < nb; ++i) { w[i] = pp(i) * inv_sqrt_d; mx = fmaxf(mx, w[i]); }
  float sum = 0.f;
  for (int i = 0; i < nb; ++i) { w[i] = expf(w[i] - mx); sum += w[i]; }
  float acc = 0.f;
  cw[0] = -bound;
  for (int i = 0; i < nb; ++i) {
    acc += min_w + (1.f - min_w * nb) * (w[i] / sum);
    cw[i + 1] = 2.f * bound * acc - bound;
  }
  cw[nb] = bound;
  mx = -3.0e38f;
  for (int i = 0; i < nb; ++i) { hh[i] = pp(nb + i) * inv_sqrt_d; mx = fmaxf(mx, hh[i]); }
  sum = 0.f;
  for (int i = 0; i < nb; ++i) { hh[i] = expf(hh[i] - mx); sum += hh[i]; }
  acc = 0.f;
  ch[0] = -bound;
  for (int i = 0; i < nb; ++i) {
    acc += min_h + (1.f - min_h * nb) * (hh[i] / sum);
    ch[i + 1] = 2.f * bound * acc - bound;
  }
  ch[nb] = bound;
  int bin = -1;
  for (int i = 0; i <= nb; ++i) {
    const float loc = ch[i] + (i == nb ? 1e-6f : 0.f);
    if (y >= loc) bin++;
  }
  bin = bin < 0 ? 0 : (bin > nb - 1 ? nb - 1 : bin);
  float in_cw = 0.f, in_w = 1.f, in_ch = 0.f, in_h = 1.f;
  for (int i = 0; i < nb; ++i)
    if (i == bin) { in_cw = cw[i]; in_w = cw[i + 1] - cw[i]; in_ch = ch[i]; in_h = ch[i + 1] - ch[i]; }
  const float cst = logf(expf(1.f - min_d) - 1.f);
  const float ud0 = (bin == 0) ? cst : pp(2 * nb + bin - 1);
  const float ud1 = (bin == nb - 1) ? cst : pp(2 * nb + bin);
  const float d0 = min_d + softplus_f(ud0), d1 = min_d + softplus_f(ud1);
  const float delta = in_h / in_w;
  const float t1 = (y - in_ch) * (d0 + d1 - 2.f * delta);
  const float a = t1 + in_h * (delta - d0);
  const float bq = in_h * d0 - t1;
  const float c = -delta * (y - in_ch);
  const float disc = bq * bq - 4.f * a * c;
  const float root = (2.f * c) / (-bq - sqrtf(disc));
  return root * in_w + in_cw;
}

// one element per thread, then cat(x0,x1)*mask (modules.py:386).  pr: [B, 3*nb-1 (padded rows ignored), T] = proj(h)*mask.
__global__ void spline_inverse_kernel(float* z, int x0_row, const float* pr, int pr_rows, const int* len, int T, int nb,
                                      float bound, float inv_sqrt_d) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const int x1_row = 1 - x0_row;
  float* p0 = &z[((long long)b * 2 + x0_row) * T + t];
  float* p1 = &z[((long long)b * 2 + x1_row) * T + t];
  if (t >= len[b]) { *p0 = 0.f; *p1 = 0.f; return; }
  const float y = *p1;
  if (!(y >= -bound && y <= bound)) return;
  const float* pp = pr + (long long)b * pr_rows * T + t;
  *p1 = spline_inverse_elem(y, [&](int i) { return pp[(long long)i * T]; }, nb, bound, inv_sqrt_d);
}

// ElementwiseAffine reverse + logw = z0 (modules.py:293-295, models.py:99-100)
__global__ void ea_logw_kernel(const float* z, int row, const float* m, const float* logs, const int* len, float* logw, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  // `row` is the physical row currently holding logical channel 0 (Flips are row relabels)
  const float v = (z[((long long)b * 2 + row) * T + t] - m[0]) * expf(-logs[0]);
  logw[(long long)b * T + t] = t < len[b] ? v : 0.f;
}

// ----------------------------------------------------------------------------- length regulator
// w = exp(logw)*mask*length_scale; w_ceil; y_len = max(1, sum) (models.py:1689-1691); inclusive
// cumsum for generate_path (commons.py:128-143).  One block per batch item.
// ea_z (optional): logw is not read but computed here from the duration flow's z, i.e. ea_logw_kernel folded in
// (ElementwiseAffine reverse, modules.py:293-295: logw = (z[row] - m) * exp(-logs), masked)
__global__ void durations_kernel(const float* logw, const int* forced, const int* len, float length_scale, int T,
                                 int* dur, int* cum, int* ylen32, int64_t* ylen64, int Tcap, int* err, const SynthDev* dv,
                                 const float* ea_z, int ea_row, const float* ea_m, const float* ea_logs) {
  __shared__ int part[256];
  if (dv) length_scale = dv->scales[1];
  const float ea_mu = ea_z ? ea_m[0] : 0.f, ea_sc = ea_z ? expf(-ea_logs[0]) : 0.f;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int L = len[b];
  const int per = (T + 255) / 256;
  const int t0 = tid * per;
  int s = 0;
  for (int t = t0; t < t0 + per && t < T; ++t) {
    int d = 0;
    if (t < L) {
      if (forced) d = forced[(long long)b * T + t];
      else {
        const float lw = ea_z ? (ea_z[((long long)b * 2 + ea_row) * T + t] - ea_mu) * ea_sc : logw[(long long)b * T + t];
        d = (int)ceilf(expf(lw) * length_scale);
      }
    }
    if (d < 0) d = 0;
    dur[(long long)b * T + t] = d;
    s += d;
  }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { int v = part[i]; part[i] = run; run += v; }
    int yl = run < 1 ? 1 : run;
    if (Tcap > 0 && yl > Tcap) { atomicOr(err, 4); yl = Tcap; }
    ylen32[b] = yl;
    if (ylen64) ylen64[b] = yl;
  }
  __syncthreads();
  int run = part[tid];
  for (int t = t0; t < t0 + per && t < T; ++t) {
    run += dur[(long long)b * T + t];
    cum[(long long)b * T + t] = run;
  }
}

// expand m_p/logs_p to frame rate by gather (instead of the reference's one-hot matmul,
// models.py:1696-1698) and sample the prior z_p = m_p + eps*exp(logs_p)*noise_scale (:1700).
// stats: [B, 2I, Tx] (m rows [0,I), logs rows [I,2I)).  Frames >= y_len: z_p = eps*noise_scale.
// Block = 64 frames x 4 channel lanes; blockIdx.y strides the channels so a single utterance (T_y ~ 150) still spreads over
// dozens of workgroups and every thread handles only a few channels after ONE binary search.
#define EXPAND_CPB 16  // channels per block (4 lanes x 4 channels)
__global__ void __launch_bounds__(256) expand_prior_kernel(const float* stats, const int* cum, const int* ylen, const float* noise,
                                                           long long noise_stride, float noise_scale, uint64_t seed, float* z_p,
                                                           int I, int Tx, int Ty, int solo, const SynthDev* dv,
                                                           const unsigned long long* item_seeds) {
  const int f = blockIdx.x * 64 + (threadIdx.x & 63), cl = threadIdx.x >> 6, b = blockIdx.z;
  if (f >= Ty) return;
  if (dv) { noise_scale = dv->scales[0]; seed = dv->seed; }  // device parameter block (graph replay) instead of by value
  const int* cb = cum + (long long)b * Tx;
  int tok = -1;
  if (f < ylen[b] && f < cb[Tx - 1]) {
    int lo = 0, hi = Tx - 1;  // first j with cum[j] > f
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cb[mid] > f) hi = mid; else lo = mid + 1; }
    tok = lo;
  }
  const int tk = tok >= 0 ? tok : 0;
#pragma unroll
  for (int i = 0; i < EXPAND_CPB / 4; ++i) {
    const int c = blockIdx.y * EXPAND_CPB + i * 4 + cl;
    if (c >= I) break;
    const float mu_ = stats[((long long)b * 2 * I + c) * Tx + tk];
    const float ls_ = stats[((long long)b * 2 * I + I + c) * Tx + tk];
    const float mu = tok >= 0 ? mu_ : 0.f, ls = tok >= 0 ? ls_ : 0.f;
    const float e = noise ? noise[((long long)b * I + c) * noise_stride + f]
                          : (solo ? philox_normal(item_seeds ? item_seeds[b] : seed + (uint64_t)b, 2, (uint32_t)c, (uint32_t)f)
                                  : philox_normal(seed, 2, (uint32_t)(b * I + c), (uint32_t)f));
    z_p[((long long)b * I + c) * Ty + f] = mu + e * expf(ls) * noise_scale;
  }
}

// tile_start[b] = sum_{b' < b} ceil(min(cap, len[b']*mul + add) / tile)  (B+1 entries): the compact tile map of
// one ragged conv launch (conv_decode_block).  One thread; B <= a few hundred.
// has_cap: len has B + 1 entries, len[B] = where the padded batch tensor ends in frames (the decoder's rag array, conv_rag_limit)
__global__ void ragged_tiles_kernel(const int* len, int B, int mul, int add, int cap, int tile, int* tile_start, int has_cap, int cap_add) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int run = 0;
  if (has_cap) { const int c = len[B] * mul + cap_add; cap = c < cap ? c : cap; }
  for (int b = 0; b < B; ++b) {
    tile_start[b] = run;
    int cols = len[b] * mul + add;
    cols = cols < cap ? cols : cap;
    cols = cols < 0 ? 0 : cols;
    run += (cols + tile - 1) / tile;
  }
  tile_start[B] = run;
}

// The decoder's ragged limits.  rag[b] = min(T_end, len_y[b] + halo) frames for b < B and rag[B] = T_end, where T_end = min(Ty, max_b len_y[b])
// is where the reference's padded batch tensor ends (Ty may be a larger capacity bucket: beyond the longest item the decoder must
// see the tensor edge, i.e. zeros, exactly like the exact-size run).  tail[b] = min(len_y[b] * tail_mul + tail_add, T_end * tail_mul):
// the columns of the last conv's output that exist for item b (what the iSTFT / PQMF / tanh tail may read).
__global__ void ragged_len_kernel(const int* len_y, int* rag, int* tail, int B, int Ty, int halo, int tail_mul, int tail_add) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > B) return;
  int mx = 0;
  for (int i = 0; i < B; ++i) mx = len_y[i] > mx ? len_y[i] : mx;
  const int end = mx < Ty ? mx : Ty;
  if (b == B) { rag[B] = end; return; }
  const int v = len_y[b] + halo;
  rag[b] = v < end ? v : end;
  const int t = len_y[b] * tail_mul + tail_add, tc = end * tail_mul;  // (an inclusive column index for the iSTFT: the last conv makes tail + 1 columns)
  tail[b] = t < tc ? t : tc;
}

// Synth.audio_float_to_int16 after `audio * scale` (vosk_tts/synth.py:16-23,128-130) on the device:
// int16(clip(a * scale * 32767, -32767, 32767)), numpy's astype truncates toward zero like the C cast
__global__ void pcm16_kernel(const float* audio, long long a_bstride, int16_t* out, long long o_bstride, long long n, float scale,
                             const SynthDev* dv) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= n) return;
  if (dv) scale = dv->pcm_scale;
  float v = audio[(long long)b * a_bstride + i] * scale;
  v = v * 32767.0f;
  v = fminf(fmaxf(v, -32767.0f), 32767.0f);
  out[(long long)b * o_bstride + i] = (int16_t)(int)v;
}

// ----------------------------------------------------------------------------- decoder tail
// spec = exp(x[:, :, :cut]); phase = pi*sin(x[:, :, cut:]) (models.py:1043-1044);
// OnnxSTFT.inverse (stft.py:246-262): conv_transpose1d with the windowed pinv-DFT basis, stride hop,
// * n_fft/hop, trim n_fft/2 both sides.  One thread per sub-band output sample.
// post: [B, S*(N+2), Tp]; mb: [B, S, Tm], Tm = (Tp-1)*hop.  basis: [N+2][N].
__global__ void istft_kernel(const float* post, const float* basis, float* mb, int S, int N, int hop, int Tp, int Tm,
                             const int* rag, int rag_mul) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y, b = blockIdx.z;
  if (n >= Tm) return;
  if (rag && n >= rag[b] * rag_mul) return;  // ragged batch: beyond this item's length + halo
  const int cut = N / 2 + 1, C = S * (N + 2);
  const int np = n + N / 2;
  int t_hi = np / hop;
  if (t_hi > Tp - 1) t_hi = Tp - 1;
  if (rag) {  // conv_post computed columns [0, rag*rag_mul/hop]; later ones were never written
    const int lim = rag[b] * rag_mul / hop;
    if (t_hi > lim) t_hi = lim;
  }
  int t_lo = (np - N + hop) / hop;  // ceil((np-N+1)/hop)
  if (np - N + 1 <= 0) t_lo = 0;
  const float* pb = post + ((long long)b * C + (long long)s * (N + 2)) * Tp;
  float a = 0.f;
  for (int t = t_lo; t <= t_hi; ++t) {
    const int j = np - t * hop;
    for (int k = 0; k < cut; ++k) {
      const float mag = expf(pb[(long long)k * Tp + t]);
      const float ph = PI_F * sinf(pb[(long long)(cut + k) * Tp + t]);
      float sn, cs;
      sincosf(ph, &sn, &cs);
      a += mag * cs * basis[k * N + j] + mag * sn * basis[(cut + k) * N + j];
    }
  }
  mb[((long long)b * S + s) * Tm + n] = a * ((float)N / (float)hop);
}

// PQMF.synthesis (pqmf.py:105-116) in polyphase form: zero-stuffing by S with gain S, pad taps/2,
// FIR [1,S,taps+1].  One thread per output sample: (taps+1)/S * S MACs instead of S*(taps+1).
__global__ void pqmf_synthesis_kernel(const float* mb, const float* filt, float* audio, int S, int taps, int Tm,
                                      long long audio_bstride, const int* rag, int rag_mul) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  const int To = Tm * S, L = taps + 1, padl = taps / 2;
  if (t >= To) return;
  if (rag && t >= rag[b] * rag_mul) { audio[(long long)b * audio_bstride + t] = 0.f; return; }  // padding: defined zeros
  const int j0 = ((padl - t) % S + S) % S;
  const int u_lim = rag ? (rag[b] * rag_mul < To ? rag[b] * rag_mul : To) : To;  // sub-band samples that exist
  float a = 0.f;
  for (int s = 0; s < S; ++s) {
    const float* xb = mb + ((long long)b * S + s) * Tm;
    for (int j = j0; j < L; j += S) {
      const int u = t + j - padl;
      if (u >= 0 && u < u_lim) a += filt[s * L + j] * (xb[u / S] * (float)S);
    }
  }
  audio[(long long)b * audio_bstride + t] = a;
}

// Fused decoder tail for one utterance or a batch: exp / sin of subband_conv_post's output, OnnxSTFT.inverse and
// PQMF.synthesis in ONE launch (the two kernels above stay as the independently written cross-check used by tests via
// vits_debug_tail_impl).  A block owns TAIL_MB sub-band samples of every sub-band (= TAIL_MB*S output samples):
//   1. mag*cos(phase), mag*sin(phase) of the frames its samples touch -> LDS (each (frame, bin) evaluated once instead of
//      once per overlapping output sample: 16x fewer transcendentals),
//   2. the sub-band samples [m0 - HM, m0 + TAIL_MB + HM) by the same windowed-basis sum as istft_kernel -> LDS (and -> mb),
//   3. the polyphase PQMF FIR over LDS -> audio.
// Same operand order as the separate kernels, so results agree to rounding of the re-used products.
#define TAIL_MB 64
struct TailParams {
  const float* post; const float* basis; const float* filt; float* mb; float* audio;
  int S, N, hop, Tp, Tm, taps;
  long long audio_bstride;
  const int* rag; int rag_mul;  // item b is valid for rag[b]*rag_mul sub-band samples (ragged batches), null = dense
};
__global__ void __launch_bounds__(256) istft_pqmf_kernel(const TailParams P) {
  extern __shared__ float sm[];
  kernarg_warm<sizeof(TailParams)>();
  const int tid = threadIdx.x, b = blockIdx.y, m0 = blockIdx.x * TAIL_MB;
  const int S = P.S, N = P.N, hop = P.hop, Tp = P.Tp, Tm = P.Tm;
  const int cut = N / 2 + 1, C = S * (N + 2), L = P.taps + 1, padl = P.taps / 2;
  const int HM = (padl + S - 1) / S + 1;
  const int n_lo = m0 - HM, nsub = TAIL_MB + 2 * HM;
  const int FR = (nsub + N) / hop + 2;
  const int n_valid = P.rag ? (P.rag[b] * P.rag_mul < Tm ? P.rag[b] * P.rag_mul : Tm) : Tm;  // sub-band samples that exist
  int f_hi_lim = Tp - 1;
  if (P.rag) { const int lim = P.rag[b] * P.rag_mul / hop; f_hi_lim = lim < f_hi_lim ? lim : f_hi_lim; }  // conv_post columns that were computed
  int f_lo = n_lo + N / 2 - N + 1;
  f_lo = f_lo <= 0 ? 0 : (f_lo + hop - 1) / hop;
  int f_hi = (n_lo + nsub - 1 + N / 2) / hop;
  f_hi = f_hi < f_hi_lim ? f_hi : f_hi_lim;
  const int nfr = f_hi - f_lo + 1;
  float* re = sm;                       // [S*cut][FR]
  float* im = re + S * cut * FR;
  float* sub = im + S * cut * FR;       // [S][nsub]
  float* bas = sub + S * nsub;          // [(N+2)][N]   windowed inverse basis
  float* flt = bas + (N + 2) * N;       // [S][L]       PQMF synthesis filters
  for (int i = tid; i < (N + 2) * N; i += 256) bas[i] = P.basis[i];
  for (int i = tid; i < S * L; i += 256) flt[i] = P.filt[i];
  for (int i = tid; i < S * cut * nfr; i += 256) {
    const int fr = i % nfr, sk = i / nfr, s = sk / cut, k = sk - s * cut;
    const float* pb = P.post + ((long long)b * C + (long long)s * (N + 2)) * Tp + f_lo + fr;
    const float mag = expf(pb[(long long)k * Tp]);
    const float ph = PI_F * sinf(pb[(long long)(cut + k) * Tp]);
    float sn, cs;
    sincosf(ph, &sn, &cs);
    re[sk * FR + fr] = mag * cs;
    im[sk * FR + fr] = mag * sn;
  }
  __syncthreads();
  for (int i = tid; i < S * nsub; i += 256) {
    const int s = i / nsub, q = i - s * nsub, n = n_lo + q;
    float a = 0.f;
    if (n >= 0 && n < n_valid) {
      const int np = n + N / 2;
      int t_hi = np / hop;
      t_hi = t_hi < f_hi_lim ? t_hi : f_hi_lim;
      int t_lo = (np - N + hop) / hop;
      if (np - N + 1 <= 0) t_lo = 0;
      for (int t = t_lo; t <= t_hi; ++t) {
        const int j = np - t * hop;
        const float* rp = re + (s * cut) * FR + (t - f_lo);
        const float* ip = im + (s * cut) * FR + (t - f_lo);
        for (int k = 0; k < cut; ++k) a += rp[k * FR] * bas[k * N + j] + ip[k * FR] * bas[(cut + k) * N + j];
      }
      a *= (float)N / (float)hop;
      if (P.mb && q >= HM && q < HM + TAIL_MB) P.mb[((long long)b * S + s) * Tm + n] = a;
    }
    sub[i] = a;
  }
  __syncthreads();
  const int To = Tm * S;
  for (int r = 0; r * 256 < TAIL_MB * S; ++r) {
    const int o = r * 256 + tid;
    const int t = m0 * S + o;
    if (o >= TAIL_MB * S || t >= To) continue;
    float a = 0.f;
    if (t < n_valid * S) {
      const int j0 = ((padl - t) % S + S) % S;
      for (int s = 0; s < S; ++s) {
        const float* xb = sub + s * nsub - n_lo;
        for (int j = j0; j < L; j += S) {
          const int u = t + j - padl;  // a multiple of S by construction of j0
          const int m = u >= 0 ? u / S : -((-u) / S);
          a += flt[s * L + j] * (xb[m] * (float)S);
        }
      }
    }
    P.audio[(long long)b * P.audio_bstride + t] = a;
  }
}

// plain HiFi-GAN tail: tanh (models.py:889)
// rag (optional): item b is valid for rag[b] * rag_mul columns, beyond that the output is 0 (those tiles were never computed)
__global__ void tanh_copy_kernel(const float* x, float* y, int T, long long x_bstride, long long y_bstride, const int* rag, int rag_mul) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  const bool live = !rag || t < rag[b] * rag_mul;
  y[(long long)b * y_bstride + t] = live ? tanhf(x[(long long)b * x_bstride + t]) : 0.f;
}

// Streaming decode: copies the frame window [start, start+W) of z [C, T] (row stride T) into a dense [C, W] buffer.
__global__ void window_copy_kernel(const float* __restrict__ z, long long zstride, int start, int W, float* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (t < W) dst[(size_t)c * W + t] = z[(size_t)c * zstride + start + t];
}

// the same for a batch of windows of one utterance (streaming: k windows decoded per graph replay): item blockIdx.z starts at starts.v[z]
struct WindowStarts { int v[8]; };
__global__ void window_copy_batch_kernel(const float* __restrict__ z, long long zstride, WindowStarts starts, int W, int I, float* __restrict__ dst) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t < W) dst[((size_t)b * I + c) * W + t] = z[(size_t)c * zstride + starts.v[b] + t];
}

// Monotonic alignment search (monotonic_align/core.pyx:7-42), one workgroup per item.
// Forward: rows are dependent, columns of one row are not -> threads own columns, the previous row's running
// scores live in a double-buffered LDS row, one barrier per frame row.  Instead of keeping the whole Q matrix for
// the backtrack, each cell records the one bit the reference's backtrack reads: Q[y-1,x] < Q[y-1,x-1] (strict).
// Backtrack: a single lane walks t_y steps over the byte map (dependent loads served from L2).
// The band  max(0, t_x+y-t_y) <= x < min(t_x, y+1)  is the reference's; cells outside it are never read.
__global__ void __launch_bounds__(256) mas_kernel(const float* __restrict__ values, const int* __restrict__ t_ys,
                                                  const int* __restrict__ t_xs, int Ty, int Tx, unsigned char* __restrict__ diag,
                                                  int* __restrict__ paths) {
  extern __shared__ float rows[];  // [2][Tx]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int t_y = t_ys[b], t_x = t_xs[b];
  const float* v = values + (size_t)b * Ty * Tx;
  unsigned char* dg = diag + (size_t)b * Ty * Tx;
  int* path = paths + (size_t)b * Ty * Tx;
  for (size_t i = tid; i < (size_t)Ty * Tx; i += nt) path[i] = 0;
  if (t_y <= 0 || t_x <= 0) return;
  const float max_neg_val = -1e9f;
  for (int y = 0; y < t_y; ++y) {
    const float* prev = rows + ((y + 1) & 1) * Tx;
    float* cur = rows + (y & 1) * Tx;
    const int x0 = t_x + y - t_y > 0 ? t_x + y - t_y : 0, x1 = t_x < y + 1 ? t_x : y + 1;
    for (int x = x0 + tid; x < x1; x += nt) {
      const float v_cur = x == y ? max_neg_val : prev[x];
      const float v_prev = x == 0 ? (y == 0 ? 0.f : max_neg_val) : prev[x - 1];
      cur[x] = v[(size_t)y * Tx + x] + (v_prev > v_cur ? v_prev : v_cur);
      // what the backtrack at row y asks about row y-1 (core.pyx:31); only read where both cells are in the band
      dg[(size_t)y * Tx + x] = (x != 0 && x != y && v_cur < v_prev) ? 1 : 0;
    }
    __syncthreads();
  }
  __threadfence_block();
  if (tid == 0) {
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
      path[(size_t)y * Tx + index] = 1;
      if (index != 0 && (index == y || dg[(size_t)y * Tx + index])) index -= 1;
    }
  }
}

// ============================================================================ StableTTS / Matcha (stts.hip.h)
// Small, launch-latency-bound pieces of MatchaTTS.synthesise; the contractions run on the shared MFMA conv kernels.

// TextEncoder.forward streams (text_encoder.py:113-127): x[b][0:E] = emb[ids[b,0,t]]*sqrt(E), four auxiliary streams
// x[b][E + (s-1)*Pd + c] = punc_emb[ids[b,s,t]]*sqrt(Pd); the bert_proj rows are written by a 1x1 conv launch.
__global__ void stts_embed_kernel(const int64_t* ids, const float* emb, const float* pemb, float* x, int H, int E, int Pd, int T,
                                  int n_vocab, float es, float ps, int* err) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const int s = c < E ? 0 : 1 + (c - E) / Pd;
  long long id = ids[((long long)b * 5 + s) * T + t];
  if (id < 0 || id >= n_vocab) { atomicOr(err, 1); id = 0; }
  x[((long long)b * H + c) * T + t] = c < E ? emb[id * E + c] * es : pemb[id * Pd + (c - E) % Pd] * ps;
}
__global__ void mask_rows_kernel(float* x, const int* len, int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t < T && t >= len[b]) x[((long long)b * C + c) * T + t] = 0.f;
}
// RotaryPositionalEmbeddings (diffusion_transformer.py:124-198) on the q and k rows of a fused [B,3H,T] qkv buffer:
// d = dk/2 rotated features per head, pairs (j, j + d/2), theta_j = 10000^(-2j/d), position = column index.
__global__ void rope_kernel(float* qkv, int H, int T, int heads, int dk) {
  const int d = dk / 2, d2 = d / 2;
  const int t = blockIdx.x * blockDim.x + threadIdx.x, hj = blockIdx.y, b = blockIdx.z >> 1, which = blockIdx.z & 1;
  if (t >= T) return;
  const int h = hj / d2, j = hj % d2;
  const float theta = 1.0f / powf(10000.0f, (float)(2 * j) / (float)d);
  const float ang = (float)t * theta, cs = cosf(ang), sn = sinf(ang);
  float* r0 = qkv + (((long long)b * 3 + which) * H + h * dk + j) * T + t;
  float* r1 = r0 + (long long)d2 * T;
  const float x0 = *r0, x1 = *r1;
  *r0 = x0 * cs - x1 * sn;
  *r1 = x1 * cs + x0 * sn;
}
// DitWrapper.time_fusion (components/decoder.py:15-16,31-33): h = (gamma[c] * h + beta[c]) * mask, film = [gamma | beta]
__global__ void film_mask_kernel(const float* h, float* out, const float* film, const int* len, int H, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const long long o = ((long long)b * H + c) * T + t;
  out[o] = t < len[b] ? film[c] * h[o] + film[H + c] : 0.f;
}
// y[n][r] = act(bias[r] + sum_j W[r][j] * x[n][j]); one wave per (row, n).  act: 0 none, 1 SiLU
__global__ void gemv_rows_kernel(const float* W, const float* bias, const float* x, int x_stride, float* y, int y_stride, int rows,
                                 int cols, int act) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n = blockIdx.y;
  if (row >= rows) return;
  float a = 0.f;
  for (int j = lane; j < cols; j += 64) a += W[(long long)row * cols + j] * x[(long long)n * x_stride + j];
  for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
  if (lane == 0) {
    a += bias ? bias[row] : 0.f;
    y[(long long)n * y_stride + row] = act == 1 ? a / (1.0f + __expf(-a)) : a;
  }
}
__global__ void copy_rows_kernel(const float* src, long long sb, float* dst, long long db, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y, b = blockIdx.z;
  if (t < T) dst[(long long)b * db + (long long)r * T + t] = src[(long long)b * sb + (long long)r * T + t];
}
// z = randn * temperature (flow_matching.py:52) into the state rows of both CFG batch items
// Batch of B utterances (blockIdx.z): item b's state also lives in CFG item B + b when cfg != 0; its library noise uses
// the Philox stream of seed + b, i.e. exactly what a single-utterance call with that seed draws.
// per-call scalars of the graph-replayed single-utterance path (stts.hip.h "fast path"): read from device memory so that one captured
// graph serves every request of its shape bucket
struct SttsDev { float temperature; float pad; unsigned long long seed; };
__global__ void cfm_init_kernel(float* cat, long long cat_b, const float* noise, long long nstride, float temperature, uint64_t seed,
                                int NF, int T, int B, int cfg, const SttsDev* dv, const unsigned long long* item_seeds = nullptr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  if (dv) { temperature = dv->temperature; seed = dv->seed; }
  const float v = (noise ? noise[(long long)c * nstride + t]
                         : philox_normal(item_seeds ? item_seeds[b] : seed + (uint64_t)b, 3u, (uint32_t)c, (uint32_t)t)) * temperature;
  cat[(long long)b * cat_b + (long long)c * T + t] = v;
  if (cfg) cat[(long long)(B + b) * cat_b + (long long)c * T + t] = v;
}
// solve_euler step with classifier-free guidance (flow_matching.py:84-93,177-189):
// x <- x + dt * (d0 + g * (d0 - d1)); the state lives in rows [0,NF) of every batch item of the in_proj input
__global__ void cfm_euler_kernel(float* cat, long long cat_b, const float* d, float dt, float g, int NF, int T, int B, int cfg) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const long long o = (long long)c * T + t;
  float d0 = d[(long long)b * NF * T + o];
  if (cfg) d0 = d0 + g * (d0 - d[(long long)(B + b) * NF * T + o]);
  const float x = cat[(long long)b * cat_b + o] + dt * d0;
  cat[(long long)b * cat_b + o] = x;
  if (cfg) cat[(long long)(B + b) * cat_b + o] = x;
}
// generate_path + matmul as a gather (matcha_tts.py:163-174): frame t belongs to the token j with cum[j-1] <= t < cum[j]
__global__ void stts_expand_kernel(const float* x, const int* cum, int Tx, float* mu_y, int CC, int T, const float* pde, float* pau) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  cum += (long long)b * Tx;
  int lo = 0, hi = Tx;  // first j with cum[j] > t (padded tokens repeat the last cumulative count, so they are never hit)
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (cum[mid] > t) hi = mid; else lo = mid + 1; }
  const bool valid = lo < Tx;
  mu_y[((long long)b * CC + c) * T + t] = valid ? x[((long long)b * CC + c) * Tx + lo] : 0.f;
  if (c == 0) pau[(long long)b * T + t] = (valid && pde) ? pde[(long long)b * Tx + lo] : 0.f;
}
// decoder_outputs[:, :, :y_len] with the frames of forced pauses replaced by frame 0 (matcha_tts.py:180-192), denormalised
// batch: item b reads its state from cat + b*cat_b, writes mel [B, NF, Tm] (zeros beyond its own length len[b])
__global__ void stts_mel_kernel(const float* cat, long long cat_b, int T, const float* pau, float* mel, int NF, int Tm, const int* len,
                                float mel_std, float mel_mean) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
  if (t >= Tm) return;
  const float* cb = cat + (long long)b * cat_b;
  float v = 0.f;
  if (t < len[b]) v = (pau[(long long)b * T + t] > 0.f ? cb[(long long)c * T] : cb[(long long)c * T + t]) * mel_std + mel_mean;
  mel[((long long)b * NF + c) * Tm + t] = v;
}
// dst[b][0..G) = table[idx[b]][0..G)  (speaker embedding lookup with the id in device memory; out-of-range ids raise bit 2 of *err)
__global__ void gather_rows_kernel(float* dst, const float* table, const int64_t* idx, int G, int n_rows, int* err) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (j >= G) return;
  long long r = idx[b];
  if (r < 0 || r >= n_rows) { if (j == 0) atomicOr(err, 2); r = 0; }
  dst[(long long)b * G + j] = table[r * G + j];
}
// dst[r][t] = vec[r % C]  (fake_content.repeat over frames and batch items, flow_matching.py:183-184)
__global__ void fill_rows_kernel(float* dst, const float* vec, int T, int C) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (t < T) dst[(long long)r * T + t] = vec[r % C];
}
__global__ void clamp_kernel(float* a, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = fminf(1.f, fmaxf(-1.f, a[i]));
}

// BertEmbeddings (transformers modeling_bert): x[c][t] = word[ids[t]][c] + position[t][c] + token_type[types[t]][c]
__global__ void bert_embed_kernel(const int64_t* ids, const int64_t* types, const float* we, const float* pe, const float* te, float* x,
                                  int H, int T, int vocab, int type_vocab, int* err) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (t >= T) return;
  long long id = ids[t], ty = types ? types[t] : 0;
  if (id < 0 || id >= vocab || ty < 0 || ty >= type_vocab) { atomicOr(err, 1); id = 0; ty = 0; }
  x[(long long)c * T + t] = we[id * H + c] + pe[(long long)t * H + c] + te[ty * H + c];
}
__global__ void transpose_ct_kernel(const float* x, float* y, int C, int T) {  // [C,T] -> [T,C]
  const int c = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
  if (c < C) y[(long long)t * C + c] = x[(long long)c * T + t];
}

// ----------------------------------------------------------------------------- attention, few-column form
// Same math as relpos_attention_mfma_kernel (attentions.py:165-260, exact banded relative positions) for SHORT sequences
// (a single utterance: T_x ~ 50 tokens, T_y ~ 150 frames), where that kernel's 32-query tiles leave 4..10 workgroups on
// the chip.  Here a workgroup owns 16 queries of one (batch, head) on v_mfma_f32_16x16x4_f32 and its NW waves split the key
// tiles of 16:  S^T[key][q] = K Q^T  (A = K fragment straight from global, B = pre-scaled Q^T kept in registers),
// a lane owns one query column and 4 of the 16 keys, so P^T's accumulator registers ARE the B operand of the PV step when
// k-step r is defined to cover keys {4g + r}; V goes through a wave-private LDS tile [key][d]; relative keys by one MFMA
// pass (Q E_k^T -> LDS), relative values by 3 k-steps over a gathered band; waves merge (m, l, O) through LDS.
// Every load address depends on T only (clamped), len[b] is first used after Q and the first K tile are in flight.
template <int DK, int NW>
__global__ void __launch_bounds__(NW * 64) relpos_attention16_kernel(const float* qkv, const float* ek, const float* ev,
                                                                      const int* len, float* out, int H, int T, int W) {
  constexpr int NS = DK / 4, ND = DK / 16, DS = DK + 4;
  constexpr int WREG = 16 * DS + 12 * 16 + 12 * 16;  // per-wave LDS: V tile [16 keys][DS] | Prel [12][16] | QE [12][16]
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, g = lane >> 4, l15 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hd = blockIdx.y, b = blockIdx.z, i0 = blockIdx.x * 16;
  const int len_raw = len[b];
  const int i = i0 + l15;
  const int ic = i < T ? i : T - 1;
  const float* qb = qkv + ((long long)b * 3 * H + (long long)hd * DK) * T;
  const float* kb = qb + (long long)H * T;
  const float* vb = kb + (long long)H * T;
  float* ob = out + ((long long)b * H + (long long)hd * DK) * T;
  float* vt = lds + wave * WREG;
  float* prel = vt + 16 * DS;
  float* qes = prel + 12 * 16;
  const float scale = 1.0f / sqrtf((float)DK);
  const int NWR = 2 * W + 1;
  const bool rel = ek != nullptr;

  // Q^T[d = 4s + g][query], K fragments of this wave's first tile: requested before len[b] is needed
  float qf[NS], kf[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) qf[s] = qb[(long long)(4 * s + g) * T + ic];
  auto load_k = [&](int jt_) {
    const int j = jt_ * 16 + l15;
    const int jc = j < T ? j : T - 1;
    const float* kp = kb + (long long)g * T + jc;
#pragma unroll
    for (int s = 0; s < NS; ++s) kf[s] = kp[(long long)(4 * s) * T];
  };
  load_k(wave);
  float eka[NS];
  if (rel) {
#pragma unroll
    for (int s = 0; s < NS; ++s) eka[s] = l15 < NWR ? ek[(l15 < NWR ? l15 : 0) * DK + 4 * s + g] : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  const int L = len_raw < T ? len_raw : T;
  if (i0 >= L) {  // whole query tile is padding: zeros (see "masked rows" in DESIGN.md)
    if (i < T)
      for (int d = (threadIdx.x >> 4); d < DK; d += NW * 4) ob[(long long)d * T + i] = 0.f;
    return;
  }
  const bool active = i < L;
#pragma unroll
  for (int s = 0; s < NS; ++s) qf[s] *= scale;
  if (rel) {  // QE^T[r][q] (attentions.py:175-177): A = E_k[r = l15][d = 4s + g]; rows r = 4g + e
    f32x4 qe = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) qe = __builtin_amdgcn_mfma_f32_16x16x4f32(eka[s], qf[s], qe, 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * g + e < 12) qes[(4 * g + e) * 16 + l15] = qe[e];
  }
  f32x4 O[ND];
#pragma unroll
  for (int db = 0; db < ND; ++db) O[db] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, l = 0.f;  // l: this lane's partial row sum (its 4 keys per tile); the 4 lane groups add up at the end
  const int ntiles = (L + 15) >> 4;
  for (int jt = wave; jt < ntiles; jt += NW) {
    const int j0 = jt * 16;
    // V tile: lane (key = l15, d = 4s + g); keys beyond L are written as 0 (select: stale memory may hold NaN)
    float vst[NS];
    const bool vok = j0 + l15 < L;
    {
      const int jc = j0 + l15 < T ? j0 + l15 : T - 1;
      const float* vp = vb + (long long)g * T + jc;
#pragma unroll
      for (int s = 0; s < NS; ++s) vst[s] = vp[(long long)(4 * s) * T];
    }
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) S = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[s], S, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) vt[l15 * DS + 4 * s + g] = vok ? vst[s] : 0.f;
    if (jt + NW < ntiles) load_k(jt + NW);
    const bool near = rel && (j0 + 15 >= i0 - W) && (j0 <= i0 + 15 + W);
    float mx = -3.0e38f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = j0 + 4 * g + e;
      float sv = S[e];
      if (near) {
        const int r = key - i + W;
        const int rc = r < 0 ? 0 : (r > 11 ? 11 : r);
        const float bias = qes[rc * 16 + l15];
        sv += (r >= 0 && r < NWR) ? bias : 0.f;
      }
      sv = key < L ? sv : -3.0e38f;
      S[e] = sv;
      mx = fmaxf(mx, sv);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = j0 + 4 * g + e;
      const float pe = key < L ? __expf(S[e] - mn) : 0.f;
      S[e] = pe;  // S now holds P^T
      psum += pe;
    }
    l = l * alpha + psum;
    m = mn;
#pragma unroll
    for (int db = 0; db < ND; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) O[db][e] *= alpha;
    // O^T += V P^T: k-step r <-> keys {4g + r}: B fragment == S[r]; A = V[d = 16 db + l15][key 4g + r]
#pragma unroll
    for (int db = 0; db < ND; ++db)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        O[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt[(4 * g + r) * DS + db * 16 + l15], S[r], O[db], 0, 0, 0);
    if (near) {  // relative values (attentions.py:191-194): Prel^T[r][q] gathered through LDS, 3 k-steps of 4
#pragma unroll
      for (int r3 = 0; r3 < 3; ++r3) prel[(3 * g + r3) * 16 + l15] = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = j0 + 4 * g + e;
        const int r = key - i + W;
        if (r >= 0 && r < NWR && key < L) prel[r * 16 + l15] = S[e];
      }
#pragma unroll
      for (int db = 0; db < ND; ++db)
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
          const int r = 4 * s3 + g;
          const float a = r < NWR ? ev[(r < NWR ? r : 0) * DK + db * 16 + l15] : 0.f;
          O[db] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, prel[r * 16 + l15], O[db], 0, 0, 0);
        }
    }
  }
  // ---- merge the waves' partial (m, l, O) through LDS; wave w finishes output registers idx == w (mod NW)
  __syncthreads();
  constexpr int NV = ND * 4 + 2;
  float* comb = lds;  // [wave][NV][64]
  {
    float* c = comb + (wave * NV) * 64 + lane;
    c[0] = m;
    c[64] = l;
#pragma unroll
    for (int db = 0; db < ND; ++db)
#pragma unroll
      for (int e = 0; e < 4; ++e) c[(2 + db * 4 + e) * 64] = O[db][e];
  }
  __syncthreads();
  float sc[NW];
  float ms = -3.0e38f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { sc[w] = comb[(w * NV) * 64 + lane]; ms = fmaxf(ms, sc[w]); }
  float lt = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { sc[w] = __expf(sc[w] - ms); lt += comb[(w * NV + 1) * 64 + lane] * sc[w]; }
  lt += __shfl_xor(lt, 16, 64);
  lt += __shfl_xor(lt, 32, 64);
  const float inv = lt > 0.f ? 1.0f / lt : 0.f;
  for (int idx = wave; idx < ND * 4; idx += NW) {
    const int db = idx >> 2, e = idx & 3;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) a += comb[(w * NV + 2 + idx) * 64 + lane] * sc[w];
    const int d = db * 16 + 4 * g + e;
    if (i < T) ob[(long long)d * T + i] = active ? a * inv : 0.f;
  }
}

// ---- shader clock under load (vits_debug_clock_probe, include/vits_mi355_debug.h): one wave per workgroup sleeps on its CU for
// `ticks` of the constant 100 MHz clock and reports shader-clock cycles per wall nanosecond over that interval.  s_memtime counts the
// shader clock of the CU's XCD whether or not this wave is issuing, so the figure is the clock the kernels running NEXT to the probe
// (another stream) see.  (Round 6: dense conv launches on N(0,1) operands run at 1.86 - 2.11 GHz, the bench's forwards at 2.35 - 2.40.)
__global__ void __launch_bounds__(64) clock_probe_kernel(double* ghz, long long ticks) {
  const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  long long w1 = w0;
  while (w1 - w0 < ticks) { __builtin_amdgcn_s_sleep(64); w1 = wall_clock64(); }
  const long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) ghz[blockIdx.x] = (double)(c1 - c0) / ((double)(w1 - w0) * 10.0);  // 100 MHz ticks -> ns
}
