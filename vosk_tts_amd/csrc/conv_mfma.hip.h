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
#include <type_traits>
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
  const float* w16;   // the same weights in 16x16x4 fragment order (conv_small.hip.h) or null
  const void* wb;     // the same weights split into bf16 (hi, lo) pieces in 32x32x16 fragment order (conv_bf3.hip.h) or null
  const float* bias;  // [Cout] or null
  float* y;           // output
  float* y2;          // optional second copy of the output (EPI_STORE, same layout): saves a device-to-device copy
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
  int x_split;        // input channels >= x_split come from g.x2 (channel ci - x_split; K-split kernel: NIN == 2):
                      // a conv over cat((x, x2), dim=1) without materialising the concatenation; multiple of 16
  int in_mask;        // zero input where t >= len[b]
  int reflect;        // ReflectionPad1d((1,0)) folded into staging: index -1 reads index 1
  const int* len;     // [B] lengths for masks
  int relu;           // 1: ReLU, 2: SiLU (stabletts FFN / cond_proj), 3: GELU-erf (BERT) on acc + bias
  const float* scale_b;  // per-batch per-row gate [B][scale_b_stride] applied after the mask, before the residual
  int scale_b_stride;    // (adaLN-Zero gates: x + gate * f(x) * mask, diffusion_transformer.py:112-113)
  int scale_b_off;
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
  // 11: CU-mate order of a grouped single-utterance launch (conv_decode_block); 12: plain order without the per-XCD pairing of
  // M-tiles (A/B: VITS_PAIR_MTILES=0); 0: the plain orders
  int xcd_mode;
  // DDSConv prologue of the small-tile kernel (conv_small.hip.h, PRO == 1): the B operand of this 1x1 conv is computed from
  // the previous layer's raw tensors instead of being read:
  //   x_in = dds_y2 ? (x + gelu(LN(dds_y2; dds_g2, dds_b2))) * mask : x * mask          (modules.py:105-107 of the layer before)
  //   B    = dds_sw ? gelu(LN(depthwise_conv(x_in; dds_sw, dds_sb, dds_dil); dds_g1, dds_b1)) : x_in   (modules.py:100-102)
  // dds_xout (optional) receives x_in.  x is g.x; all tensors [B, C_in, T]; P.len gives the mask.
  const float* dds_y2; const float* dds_g2; const float* dds_b2;
  const float* dds_sw; const float* dds_sb; const float* dds_g1; const float* dds_b1;
  float* dds_xout;
  int dds_dil;
  // first layer of a ConvFlow's DDSConv (dds_y2 == null): x is the conditioning tensor and the layer input is
  // x_in = dds_pw[c] * dds_z[b][t] + dds_pb[c] + x[c][t]   (ConvFlow.pre, a Conv1d(1, C, 1), + g: modules.py:365-366,97-98)
  const float* dds_z; const float* dds_pw; const float* dds_pb;
  long long dds_z_bstride;
  // LayerNorm prologue of the small-tile kernel (PRO == 2): the staged tile holds the RAW tensor y (all channels of the tile's
  // columns); before the MFMAs it is replaced by  ((LN_c(y; ln_g, ln_b) + ln_vec[b][c] + ln_base[c][t]) * [valid column])
  // -- modules.LayerNorm (modules.py:29-32) folded into its consumer; ln_out (optional) receives the normalised tensor for
  // the tile's own 16 columns (written once per column tile, by the workgroups of M-tile 0) since the residual path needs it.
  const float* ln_g; const float* ln_b; const float* ln_base; const float* ln_vec;
  int ln_vec_stride, ln_vec_off;
  float* ln_out;
  // PRO == 3: the channel statistics come from the PRODUCER of y (ln_stat_in: per (batch item, 16-row block of y, column) the
  // block's mean and centred second moment, written by the producer conv's epilogue when its ln_stat_out is set); the consumer
  // merges the ln_nmb partials in fixed order and normalises while staging -- no statistics pass, no extra barriers, and the
  // ~C_out/16 workgroups that share a column tile no longer each redo the reduction.
  const float* ln_stat_in; float* ln_stat_out; int ln_nmb;
  int row_len;        // LDS row = N_T + halo
  long long* dbg;     // optional phase cycle stamps (tools/ only); null in production
  // Ragged batches (the mask-free decoder): rag[b] = item b's length in frames, rag[B] = where the padded batch tensor ends.  Input columns
  // >= min(rag[b]*rag_in_mul + rag_in_add, rag[B]*rag_in_mul) read as 0 and output tiles starting at or beyond
  // min(rag[b]*rag_out_mul + rag_out_add, rag[B]*rag_out_mul) are skipped: the adds are what THIS layer still has to produce beyond the
  // item's end for every later layer's receptive field (engine.hip run_decoder).  null = dense (reference-padded).
  const int* rag;
  int rag_in_mul, rag_in_add, rag_out_mul, rag_out_add;
  int rag_tab_add;      // host only: the add the launch's compact tile map is built with (>= rag_out_add; one map per decoder stage instead of one per layer)
  int rag_out_cap_add;  // output columns that exist beyond rag[B] * rag_out_mul (1 for the reflection-padded conv_post: T + 1 columns)
  // Masked stages (encoder / duration predictor / flow) of a ragged batch: every consumer of this conv's
  // output is either column-local or masks its input at len[b], so tiles that start at or beyond len[b]
  // are not computed at all (their memory keeps whatever it held; see DESIGN.md "ragged batches").
  int skip_len;
  // Compact tile map of a ragged launch: tile_start[b] = number of column tiles of items < b (B+1 entries,
  // built on the device by ragged_tiles_kernel for this launch's tile width).  Working tiles get the lowest
  // block ids so they are dispatched first and spread over all CUs; ids >= tile_start[B] * ... exit at once.
  const int* tile_start;
};

// limit of a ragged launch in columns: rag[b] * mul + add, capped where the padded batch tensor ends (rag[B] frames)
__device__ __forceinline__ int conv_rag_limit(int rl, int rcap, int mul, int add, int cap_add = 0) {
  const int v = rl * mul + add, c = rcap * mul + cap_add;
  return v < c ? v : c;
}

// Kernel-argument warm-up.  ConvParams travels by value (~700 bytes = 11 scalar-cache lines) and hipcc fetches its fields lazily,
// one s_load + s_waitcnt lgkmcnt(0) per region right before first use: measured (tools/ddsdbg.py stamps) 2.4 us pass between
// kernel entry and the first operand load of a small conv -- four to five SERIALISED cold misses of the scalar cache.  Touching
// one dword of every 64-byte line of the kernarg segment at entry makes those misses overlap (one memory round trip); every
// later s_load of the compiler then hits the scalar cache.
// (ONE asm statement: the loads complete asynchronously, so the dummy destination register must not be visible to the
//  register allocator until the s_waitcnt inside the same statement has retired them.)
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  static_assert(BYTES <= 1024, "extend kernarg_warm");
  constexpr int LAST = (BYTES - 1) / 64 * 64;
#define KW_OFF(i) ((i) * 64 < LAST ? (i) * 64 : LAST)
  const unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  int d;
  asm volatile(
      "s_load_dword %0, %1, %2\n s_load_dword %0, %1, %3\n s_load_dword %0, %1, %4\n s_load_dword %0, %1, %5\n"
      "s_load_dword %0, %1, %6\n s_load_dword %0, %1, %7\n s_load_dword %0, %1, %8\n s_load_dword %0, %1, %9\n"
      "s_load_dword %0, %1, %10\n s_load_dword %0, %1, %11\n s_load_dword %0, %1, %12\n s_load_dword %0, %1, %13\n"
      "s_load_dword %0, %1, %14\n s_load_dword %0, %1, %15\n s_load_dword %0, %1, %16\n s_load_dword %0, %1, %17\n"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(d)
      : "s"(kp), "n"(KW_OFF(0)), "n"(KW_OFF(1)), "n"(KW_OFF(2)), "n"(KW_OFF(3)), "n"(KW_OFF(4)), "n"(KW_OFF(5)), "n"(KW_OFF(6)),
        "n"(KW_OFF(7)), "n"(KW_OFF(8)), "n"(KW_OFF(9)), "n"(KW_OFF(10)), "n"(KW_OFF(11)), "n"(KW_OFF(12)), "n"(KW_OFF(13)),
        "n"(KW_OFF(14)), "n"(KW_OFF(15))
      : "memory");
#undef KW_OFF
}

// block id -> (m tile, group, column tile, batch item); returns false when the block has no work
// Plain dispatch order with the M-tiles of one column tile on ONE XCD: workgroup L runs on XCD L % 8, so consecutive ids (the
// M-tiles of a column tile, which stage the same activation window) would land on ntiles_m different L2s and each fetch the
// window over the fabric.  Inside every run of 8 * ntiles_m ids, XCD x takes ids [x * ntiles_m, (x + 1) * ntiles_m) in its own
// dispatch order; the tail that does not fill a run keeps the identity.  The spread of tiles over the XCDs is unchanged.
__device__ __forceinline__ int conv_pair_mtiles(int ntiles_m, int L = blockIdx.x, int nblk = gridDim.x) {
  const int run = 8 * ntiles_m;
  if (ntiles_m < 2 || L >= (nblk / run) * run) return L;
  const int base = L / run * run, r = L - base, x = r & 7, w = r >> 3;  // w-th workgroup of XCD x inside this run
  return base + x * ntiles_m + w;
}
// (Lb / nblk: the block id and grid size to decode for -- the launch's own by default; a persistent kernel that walks VIRTUAL blocks
//  passes its own, conv_sk.hip.h)
__device__ __forceinline__ bool conv_decode_block(const ConvParams& P, int& mt, int& grp, int& nt, int& b, int Lb = blockIdx.x, int nblk_ = gridDim.x) {
  if (P.tile_start) {
    // plain dispatch order (no XCD-contiguous remap: working tiles must be spread over all XCDs), M-tiles paired per XCD
    int id = P.xcd_mode == 12 ? Lb : conv_pair_mtiles(P.ntiles_m, Lb, nblk_);
    mt = id % P.ntiles_m; id /= P.ntiles_m;
    const int total = P.tile_start[P.B];
    int q;
    if (P.n_groups > 1) { q = id % (P.ntiles_n * P.B); grp = id / (P.ntiles_n * P.B); }  // heaviest group first
    else { q = id; grp = 0; }
    if (q >= total) return false;
    int lo = 0, hi = P.B - 1;  // last b with tile_start[b] <= q
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (P.tile_start[mid] <= q) lo = mid; else hi = mid - 1; }
    b = lo;
    nt = q - P.tile_start[b];
    return true;
  }
  if (P.xcd_mode == 11) {
    // three grouped convs of one utterance, heaviest first (11 / 7 / 3 taps).  Two workgroups share a CU and run there mostly one
    // after the other (profiles/r3_blocktrace_c2.txt), the w-th and (w + 32)-th workgroup of an XCD's stream are CU mates, and the
    // launch lasts as long as the CU with the heaviest pair.  XCD x takes tiles x, x + 8, ... of every group in an order that puts a
    // 3-tap tile under every 11-tap one (14 units) and 7-tap tiles under each other (14) instead of 11 + 7 (18).
    const int x = Lb & 7, w = Lb >> 3;
    const int per = P.ntiles_m * P.ntiles_n, n = (per - x + 7) >> 3;  // tiles of one group on this XCD
    const int a = n < 32 ? n : 32;            // first pass over the 32 CUs: 11-tap tiles, then 7-tap ones
    const int b7 = 32 - a < n ? 32 - a : n;   // 7-tap tiles that fit in the first pass
    int idx;
    if (w < a) { grp = 0; idx = w; }
    else if (w < a + b7) { grp = 1; idx = w - a; }
    else if (w < a + b7 + n) { grp = 2; idx = w - a - b7; }
    else if (w < a + 2 * n) { grp = 1; idx = w - a - n; }       // the remaining 7-tap tiles
    else { grp = 0; idx = w - 2 * n; }                          // 11-tap tiles beyond the first 32 (large launches)
    if (idx >= n) return false;
    const int t = idx * 8 + x;
    mt = t % P.ntiles_m; nt = t / P.ntiles_m;
    b = 0;
    return true;
  }
  int id;
  {
    // bijective XCD remap: block L runs on XCD L%8; give each XCD a contiguous range of logical ids so
    // tiles sharing an activation window share an L2
    const int nblk = nblk_, L = Lb;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, within = L >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  if (P.n_groups > 1) {
    // grouped launch (k = 11/7/3 ResBlocks, sorted heaviest first by the launcher): plain dispatch order
    // with the group outermost, so the long blocks start first and the launch tail is made of short ones
    id = (P.B > 1 && P.xcd_mode != 12) ? conv_pair_mtiles(P.ntiles_m, Lb, nblk_) : Lb;  // (one utterance: an XCD keeps its M-tile's weight rows, profiles/r3_xcd_map.txt)
    mt = id % P.ntiles_m; id /= P.ntiles_m;
    nt = id % P.ntiles_n; id /= P.ntiles_n;
    b = id % P.B;
    grp = id / P.B;
  } else {
    mt = id % P.ntiles_m; id /= P.ntiles_m;
    nt = id % P.ntiles_n;
    b = id / P.ntiles_n;
    grp = 0;
  }
  return true;
}
#ifdef CONV_TIMING
#define CONV_DBG(k) do { if (P.dbg && blockIdx.x == 0 && lane == 0) { P.dbg[wave * 8 + (k)] = __builtin_readcyclecounter(); if ((k) == 0) P.dbg[wave * 8 + 7] = __builtin_amdgcn_s_getreg(63492); } \
    /* block trace (timing build): wall-clock start / end (100 MHz) and hardware id of every workgroup, wave 0 */ \
    if (P.dbg && wave == 0 && lane == 0 && blockIdx.x < 4000 && ((k) == 0 || (k) == 5)) { \
      P.dbg[128 + blockIdx.x * 4 + ((k) == 0 ? 0 : 1)] = wall_clock64(); \
      if ((k) == 0) { P.dbg[128 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492); P.dbg[128 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20); } } } while (0)
#define CONV_DBG_DO(x) x
#else
#define CONV_DBG(k) do { } while (0)
#define CONV_DBG_DO(x)
#endif

__device__ __forceinline__ float conv_act_in(float v, float scale, float slope) {
  v *= scale;
  return v > 0.f ? v : v * slope;
}

// ---- shared epilogue over one accumulator fragment ------------------------------------------------
// NE elements of one output column `col` at rows row0 + (e&3) + 8*(e>>2) (e = e0 .. e0+NE-1 of the
// 32x32 C/D layout).  Every load uses a clamped (always valid) address and every wave-uniform
// condition is hoisted out of the element loops, so the loads of a fragment issue back to back and
// are waited for once (a per-element `if (ptr) v += ptr[..]` makes hipcc serialise them).
template <int EPI, int NE>
__device__ __forceinline__ void conv_epilogue_frag(const ConvParams& P, const ConvGroup& G, int b, int lenb, int row0, int e0,
                                                   int col, float (&v)[NE]) {
  const bool colok = col < P.Tout;
  const int colc = colok ? col : P.Tout - 1;
  int row[NE];
  bool ok[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = e0 + i;
    const int r = row0 + (e & 3) + 8 * (e >> 2);
    ok[i] = colok && r < P.Cout;
    row[i] = r < P.Cout ? r : P.Cout - 1;
  }
  if (EPI == EPI_STORE) {
    if (P.ups_u) {
#pragma unroll
      for (int i = 0; i < NE; ++i) {
        const int phase = row[i] / P.ups_cout, co = row[i] - phase * P.ups_cout;
        const float bv = G.bias ? G.bias[co] : 0.f;
        if (ok[i]) G.y[(long long)b * P.y_bstride + (long long)co * P.Tout_stride + (long long)colc * P.ups_u + phase] = v[i] + bv;
      }
      return;
    }
    long long o[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) o[i] = (long long)b * P.y_bstride + (long long)row[i] * P.Tout_stride + colc;
    if (G.bias) {
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] += G.bias[row[i]];
    }
    if (P.bias_b) {
      const float* bb = P.bias_b + (long long)b * P.bias_b_stride + P.bias_b_off;
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] += bb[row[i]];
    }
    if (P.relu == 1) {
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    } else if (P.relu == 2) {
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] = v[i] / (1.0f + __expf(-v[i]));
    } else if (P.relu == 3) {  // GELU (erf), BERT intermediate
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] = 0.5f * v[i] * (1.0f + erff(v[i] * 0.70710678118654752440f));
    }
    if (P.out_mask && col >= lenb) {
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] = 0.f;
    }
    if (P.scale_b) {
      const float* sb = P.scale_b + (long long)b * P.scale_b_stride + P.scale_b_off;
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] *= sb[row[i]];
    }
    if (G.res) {
      float r[NE];
#pragma unroll
      for (int i = 0; i < NE; ++i) r[i] = G.res[o[i]];
#pragma unroll
      for (int i = 0; i < NE; ++i) v[i] += r[i];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i)
      if (ok[i]) G.y[o[i]] = v[i];
    if (G.y2) {
#pragma unroll
      for (int i = 0; i < NE; ++i)
        if (ok[i]) G.y2[o[i]] = v[i];
    }
  } else if (EPI == EPI_RESSKIP) {
    const bool valid = col < lenb;
    float bv[NE], old[NE];
    long long o[NE];
    bool to_skip[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      to_skip[i] = P.last || row[i] >= P.H;
      const int sr = (P.last || row[i] < P.H) ? row[i] : row[i] - P.H;
      o[i] = (long long)b * P.y_bstride + (long long)sr * P.Tout_stride + colc;
      bv[i] = G.bias[row[i]];
    }
    // rows < H update x in place (modules.py:171); rows >= H (or every row of the last layer) feed the
    // skip accumulator (modules.py:172-175).  First layer stores, later layers accumulate.
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const float* src = to_skip[i] ? P.skip : P.io;
      old[i] = (to_skip[i] && P.first) ? 0.f : src[o[i]];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      float r = old[i] + v[i] + bv[i];
      if (to_skip[i]) {
        if (P.last && !valid) r = 0.f;  // output * x_mask (modules.py:176)
      } else if (!valid) {
        r = 0.f;  // x = (x + res_acts) * x_mask
      }
      float* dst = to_skip[i] ? P.skip : P.io;
      if (ok[i]) dst[o[i]] = r;
    }
  } else if (EPI == EPI_COUPLE) {
    // previous z = u (before the Flip that precedes this layer); this layer's logical input is
    // flip(u): x0[c] = u[I-1-c], x1[c] = u[half-1-c].  new z = cat(x0, (x1 - m)*mask)  (models.py:390-392)
    const int half = P.H, I2 = 2 * P.H;
    const long long bo = (long long)b * P.y_bstride + colc;
    float x0[NE], x1[NE], bv[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      bv[i] = G.bias[row[i]];
      x1[i] = P.u[bo + (long long)(half - 1 - row[i]) * P.Tout_stride];
      x0[i] = P.u[bo + (long long)(I2 - 1 - row[i]) * P.Tout_stride];
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      if (ok[i]) {
        P.io[bo + (long long)(half + row[i]) * P.Tout_stride] = col < lenb ? (x1[i] - (v[i] + bv[i])) : 0.f;
        P.io[bo + (long long)row[i] * P.Tout_stride] = x0[i];
      }
    }
  }
}
// WN gate (commons.py:100-107): channels ch0 + (e&3) + 8*(e>>2) of batch b at column col from the
// tanh / sigmoid pre-activation fragments
template <int NE>
__device__ __forceinline__ void conv_epilogue_gate(const ConvParams& P, const ConvGroup& G, int b, int ch0, int e0, int col,
                                                   float (&at)[NE], float (&as)[NE]) {
  const bool colok = col < P.Tout;
  const int colc = colok ? col : P.Tout - 1;
  const float* bb = P.bias_b ? P.bias_b + (long long)b * P.bias_b_stride + P.bias_b_off : G.bias;
  const float bscale = P.bias_b ? 1.f : 0.f;
  float b1[NE], b2[NE], c1[NE], c2[NE];
  int ch[NE];
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = e0 + i;
    const int c = ch0 + (e & 3) + 8 * (e >> 2);
    ch[i] = c < P.H ? c : P.H - 1;
    b1[i] = G.bias[ch[i]];
    b2[i] = G.bias[P.H + ch[i]];
    c1[i] = bb[ch[i]];
    c2[i] = bb[P.H + ch[i]];
  }
#pragma unroll
  for (int i = 0; i < NE; ++i) {
    const int e = e0 + i;
    const int c = ch0 + (e & 3) + 8 * (e >> 2);
    const float tv = tanhf(at[i] + b1[i] + bscale * c1[i]);
    const float sv = 1.0f / (1.0f + __expf(-(as[i] + b2[i] + bscale * c2[i])));
    if (colok && c < P.H) G.y[(long long)b * P.y_bstride + (long long)ch[i] * P.Tout_stride + colc] = tv * sv;
  }
}

// ---- branch-free staging -----------------------------------------------------------------------
// Column geometry of a tile is the same for every channel chunk: per staging column j the clamped
// input offset and a validity flag are computed once; chunk loads are then unconditional loads +
// selects, with the number of summed inputs (1, or 3 for the MRF mean) hoisted out of the loop.
#define CONV_STAGE_COLS(JT_)                                                                 \
  int toff[JT_];                                                                             \
  bool tok[JT_];                                                                             \
  _Pragma("unroll") for (int j = 0; j < JT_; ++j) {                                          \
    const int col = lane + 64 * j;                                                           \
    int t = t_base + col;                                                                    \
    if (P.reflect && t == -1) t = (P.Tin > 1) ? 1 : 0;                                       \
    tok[j] = col < ROW && t >= 0 && t < t_lim;                                               \
    toff[j] = t < 0 ? 0 : (t >= P.Tin ? P.Tin - 1 : t);                                      \
  }

// Buffer addressing for the big-tile kernel's streams: descriptor (4 SGPRs, wave-uniform base) + scalar byte offset (SGPR) +
// per-lane byte offset (VGPR) -> buffer_load ... offen with NO 64-bit VALU address math per load (hipcc re-associates a
// "uniform pointer + lane offset" global load into a per-lane 64-bit base + v_lshl_add_u64 per load).  Raw buffer, stride 0,
// no range limit (every offset used is in bounds by construction; clamped indices for the prefetch overrun).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bt_rsrc(const void* p) {
  // the base IS wave-uniform (block / wave indices only); say so, or the descriptor lands in VGPRs and every load gets a waterfall loop
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float bt_ld(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off, uni_off, 0));
}
__device__ __forceinline__ f32x4 bt_ld4(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uni_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uni_off, 0));
}

// ---- fragment-wide plain epilogue (bias, mask, residual) of the 128x128 / 64x128 kernels --------------------------------------
// Every ResBlock / encoder / flow STORE conv.  One 32x32 fragment at a time: ALL of a fragment's operands (4 x dwordx4 bias, 16
// residual values) are requested at once and the NEXT fragment's before this one's stores, on buffer addressing (descriptor +
// scalar row offset + one lane offset).  conv_epilogue_frag walks the 64 values of a wave four at a time with two dependent round
// trips (bias, residual) and a branch per element.  mw / nw: first row / column of the wave's fragments (wave-uniform).
__device__ __forceinline__ bool conv_epilogue_store_fast_ok(const ConvParams& P, const ConvGroup& G) {
  return !P.ups_u && !P.bias_b && !P.scale_b && P.relu == 0 && !G.y2 && (P.Cout & 7) == 0;
}
template <int MI, int NI>
__device__ __forceinline__ void conv_epilogue_store_fragments(const ConvParams& P, const ConvGroup& G, int b, int lenb, int mw, int nw, int h,
                                                              int l31, f32x16 (&acc)[MI][NI]) {
  constexpr int NF = MI * NI;
  const __amdgpu_buffer_rsrc_t ry = bt_rsrc(G.y + (long long)b * P.y_bstride);
  const __amdgpu_buffer_rsrc_t rr = bt_rsrc((G.res ? G.res : G.y) + (long long)b * P.y_bstride);
  const __amdgpu_buffer_rsrc_t rb = bt_rsrc(G.bias ? G.bias : G.w);
  const bool has_b = G.bias != nullptr, has_r = G.res != nullptr;
  f32x4 bq[2][4];
  float rv[2][16];
  auto frag_rows = [&](int f) { return mw + (f / NI) * 32; };  // wave-uniform first row of fragment f (lane rows: + 4 h)
  auto frag_col = [&](int f) { return nw + (f % NI) * 32 + l31; };
  auto frag_voff = [&](int f) {
    const int col = frag_col(f);
    return (unsigned)((4 * h) * P.Tout_stride + (col < P.Tout ? col : P.Tout - 1)) * 4u;
  };
  auto request = [&](int f, int slot) {
    const int R = frag_rows(f);
    const unsigned vo = frag_voff(f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int rq = R + 8 * q;  // rows rq + 4 h + (0..3); C_out is a multiple of 8, so the quad is valid for both half-waves or for neither
      if (rq > P.Cout - 8) rq = P.Cout - 8;
      bq[slot][q] = bt_ld4(rb, (unsigned)(4 * h) * 4u, (unsigned)rq * 4u);
#pragma unroll
      for (int i = 0; i < 4; ++i) rv[slot][4 * q + i] = bt_ld(rr, vo, (unsigned)(rq + i) * (unsigned)P.Tout_stride * 4u);
    }
  };
  request(0, 0);
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    if (f + 1 < NF) request(f + 1, (f + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    const int R = frag_rows(f), col = frag_col(f);
    const unsigned vo = frag_voff(f);
    const bool masked = P.out_mask && col >= lenb;
    float x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float t = acc[f / NI][f % NI][e] + (has_b ? bq[f & 1][e >> 2][e & 3] : 0.f);
      if (masked) t = 0.f;
      x[e] = t + (has_r ? rv[f & 1][e] : 0.f);
    }
    if (col < P.Tout) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (R + 8 * q < P.Cout) {  // wave-uniform
#pragma unroll
          for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[4 * q + i]), ry, vo, (unsigned)(R + 8 * q + i) * (unsigned)P.Tout_stride * 4u, 0);
        }
    }
  }
}

template <int WM, int WN, int MI, int NI, int EPI>
__global__ void __launch_bounds__(WM * WN * 64, 3) conv_mfma_kernel(const ConvParams P) {  // 3 waves per SIMD: 3 workgroups of 4 waves per CU, or 6 of 2
  // 256-thread workgroups, or (round 6) TWO-wave workgroups: the same wave tile on half the columns, so that a launch of 1 - 3 four-wave
  // workgroups per CU becomes 2 - 6 two-wave ones and quantises half as coarsely over the 256 CUs (the WaveNet gate conv of a 32-item
  // batch: 580 tiles of 128 x 64 = 2.27 per CU, 3 on the critical CUs -> 1160 of 128 x 32 = 4.5 / 5)
  static_assert(WM * WN == 4 || WM * WN == 2, "256- or 128-thread workgroups");
  constexpr int NWV = WM * WN;          // waves per workgroup
  constexpr int RPW = CONV_CI_T / NWV;  // chunk rows a wave stages
  constexpr int M_T = WM * MI * 32;
  constexpr int N_T = WN * NI * 32;
  constexpr int JT = N_T < 64 ? 1 : (N_T + CONV_MAX_HALO + 63) / 64;  // 32-column tiles: N_T + halo <= 64 (launcher's condition)
  static_assert(64 * (JT - 1) <= N_T, "staging column groups");
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row / weight-block address math stays scalar
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, l31 = lane & 31;

  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  // block-uniform by construction, but the ragged tile map is read with vector loads: tell the compiler, so that row / column /
  // item offsets stay in SGPRs (buffer addressing needs wave-uniform scalar offsets, else every access gets a waterfall loop)
  mt = __builtin_amdgcn_readfirstlane(mt); grp = __builtin_amdgcn_readfirstlane(grp);
  nt = __builtin_amdgcn_readfirstlane(nt); b = __builtin_amdgcn_readfirstlane(b);
  const ConvGroup& G = P.g[grp];
  CONV_DBG_DO(long long bt_c0 = 0; if (P.dbg && tid == 0 && blockIdx.x < 4000) {  // block trace (timing build): wall-clock start, hardware id, XCC id
    P.dbg[128 + blockIdx.x * 4 + 0] = wall_clock64();
    P.dbg[128 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492);
    P.dbg[128 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20);
    bt_c0 = __builtin_readcyclecounter();
  })
#ifdef CONV_TIMING
  // (round 6) the shader-clock cycles the workgroup lived go into bits 8.. of the XCC word: cycles / wall time = the clock this CU ran at
#define BT_BLOCK_END() do { if (P.dbg && tid == 0 && blockIdx.x < 4000) { P.dbg[128 + blockIdx.x * 4 + 1] = wall_clock64(); \
    P.dbg[128 + blockIdx.x * 4 + 3] |= (__builtin_readcyclecounter() - bt_c0) << 8; } } while (0)
#else
#define BT_BLOCK_END() do { } while (0)
#endif

  const int ROW = P.row_len;
  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  int tap_base = 0;
  if (P.ups_u) tap_base = P.ups_shift[m0 / P.ups_cout];
  const int nchunks = P.Cin / CONV_CI_T;
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }
  if (P.rag) {
    const int rl = P.rag[b], rc = P.rag[P.B];  // rag[B]: where the padded batch tensor ends (frames): no limit reaches beyond it
    if (n0 >= conv_rag_limit(rl, rc, P.rag_out_mul, P.rag_out_add, P.rag_out_cap_add)) return;  // whole tile is padding of this item (block-uniform)
    const int il = conv_rag_limit(rl, rc, P.rag_in_mul, P.rag_in_add);
    t_lim = il < t_lim ? il : t_lim;
  }
  if (P.skip_len && n0 >= P.len[b]) return;  // masked stage: the whole tile lies in this item's padding

  // ---- staging: wave w owns chunk rows w, w + NWV, ...; lanes stride over columns
  float stg[RPW][JT];
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float* xb2 = G.x2 ? G.x2 + (long long)b * P.x_bstride : nullptr;
  const float* xb3 = G.x3 ? G.x3 + (long long)b * P.x_bstride : nullptr;
  const float in_scale = P.in_scale, in_slope = P.in_slope;
  const int t_base = n0 - G.pad_l;

  CONV_STAGE_COLS(JT)
  unsigned tob[JT];  // byte offsets of the staging columns: row pointers are wave-uniform, so a load is base (SGPR pair) + tob (VGPR)
#pragma unroll
  for (int j = 0; j < JT; ++j) tob[j] = (unsigned)toff[j] * 4u;
  const __amdgpu_buffer_rsrc_t rx = bt_rsrc(xb), rx2 = bt_rsrc(xb2 ? xb2 : xb), rx3 = bt_rsrc(xb3 ? xb3 : xb);
  auto load_chunk = [&](int c) {
    unsigned roff[RPW];  // byte offset of the chunk's rows inside the item (wave-uniform; < 2^31 by the arena's size limits)
    // channel-concatenated second input (x_split): chunks at or beyond the split read g.x2 at channel ci - x_split
    const bool second = P.x_split && c * CONV_CI_T >= P.x_split;  // block-uniform
    const int cb = second ? c * CONV_CI_T - P.x_split : c * CONV_CI_T;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) roff[rr] = (unsigned)((long long)(P.x_ch_off + (cb + wave + NWV * rr) * P.x_ch_sign) * P.Tin_stride * 4);
    if (P.x_split) {
      const __amdgpu_buffer_rsrc_t rs = second ? rx2 : rx;
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int j = 0; j < JT; ++j) stg[rr][j] = bt_ld(rs, tob[j], roff[rr]);
    } else if (xb2) {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int j = 0; j < JT; ++j)
          stg[rr][j] = bt_ld(rx, tob[j], roff[rr]) + bt_ld(rx2, tob[j], roff[rr]) + (xb3 ? bt_ld(rx3, tob[j], roff[rr]) : 0.f);
    } else {
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
        for (int j = 0; j < JT; ++j) stg[rr][j] = bt_ld(rx, tob[j], roff[rr]);
    }
    // NOTE: nothing here may CONSUME the loaded values — the raw registers ride through the whole tap
    // loop and are activated only in store_chunk, otherwise every chunk starts with a memory-latency stall
  };
  auto store_chunk = [&](int buf) {
    float* dst = lds + buf * (CONV_CI_T * ROW);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        const float v = tok[j] ? conv_act_in(stg[rr][j], in_scale, in_slope) : 0.f;
        if (j < JT - 1 || col < ROW) dst[(wave + NWV * rr) * ROW + col] = v;  // 64 (JT - 1) <= N_T <= ROW: only the last group needs the test
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
  __amdgpu_buffer_rsrc_t wp[MI];  // wave-uniform descriptors of this wave's weight m-blocks: a fragment load is wp + step-group offset (scalar) + lane * 16
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + wm * MI + mi;
    if (mb >= n_mblocks) mb = 0;  // padded tile: compute on valid memory, never stored
    wp[mi] = bt_rsrc(reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64);
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  const int n_sg = G.n_sg;

  CONV_DBG_DO(long long dbg_t0 = 0; long long dbg_tap = 0; long long dbg_sync = 0; long long dbg_x = 0; if (P.dbg) dbg_t0 = __builtin_readcyclecounter();)

  // Weight fragments live in two STATIC slots: a tap computes on one and requests the next tap's fragments into the other,
  // taps run in pairs (slot 0 -> 1 -> 0), so nothing is copied between taps; an odd tap count leaves the next chunk's first
  // fragments in slot 1 and they are moved once per chunk.  (One slot pair + a rotation per tap cost 8 v_mov_b64 per 32 MFMAs.)
  f32x4 a[2][MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    a[0][mi][0] = bt_ld4(wp[mi], lane16, 0);
    a[0][mi][1] = bt_ld4(wp[mi], lane16, 1024);
  }
  int sg = 2;  // next step-group to fetch
  // (the first fragments are requested BEFORE the first activation chunk: one memory round trip in the tile's prologue, not two)
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  CONV_DBG_DO(if (P.dbg && blockIdx.x == 0 && lane == 0) P.dbg[wave * 8 + 0] = __builtin_readcyclecounter() - dbg_t0;)

  // one tap on slot CUR; the next tap's fragments are requested into the other slot first
  auto tap = [&](auto CUR, const float* lb, int kk) {
    constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
    {
      // unconditional (clamped) prefetch: a fixed number of loads per tap lets hipcc emit a counted
      // s_waitcnt vmcnt(N) instead of draining the prefetch it has just issued
      const int sgc = sg < n_sg ? sg : n_sg - 2;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        a[nxt][mi][0] = bt_ld4(wp[mi], lane16, (unsigned)sgc * 1024u);
        a[nxt][mi][1] = bt_ld4(wp[mi], lane16, (unsigned)sgc * 1024u + 1024u);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch AHEAD of this tap's MFMAs
    }
    sg += 2;
    const float* lk = lb + kk * dil;
    float bv[8][NI];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[p][ni] = lk[2 * p * ROW + ni * 32];
    __builtin_amdgcn_sched_barrier(0);  // all B-fragment reads of the tap in flight before its first MFMA
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][mi][p >> 2][p & 3], bv[p][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  using slot0 = std::integral_constant<int, 0>;
  using slot1 = std::integral_constant<int, 1>;

  for (int c = 0; c < nchunks; ++c) {
    CONV_DBG_DO(if (P.dbg) dbg_x = __builtin_readcyclecounter();)
    const float* lb = lds + (c & 1) * (CONV_CI_T * ROW) + h * ROW + wn * (NI * 32) + l31 + tap_base;
    // Tap 0 first, THEN the next chunk's activation loads: vmcnt retires in order, so a wait for weight fragments that were
    // requested behind the activation loads also waits for those.  Issued ahead of tap 0, the 12 loads' full latency sat in
    // front of the chunk's first MFMA; issued here the next wait that covers them is the one before tap 1 (>= 2048 MFMA cycles
    // later; for a 1-tap conv the fragment move below, counted exactly).
    tap(slot0{}, lb, 0);
    if (c + 1 < nchunks) load_chunk(c + 1);
    __builtin_amdgcn_sched_barrier(0);
    int kk = 1;
#pragma unroll 1
    for (; kk + 1 < K; kk += 2) {
      tap(slot1{}, lb, kk);
      tap(slot0{}, lb, kk + 1);
    }
    if (kk < K) {
      tap(slot1{}, lb, kk);  // even tap count: the next chunk's first fragments are in slot 0 already
    } else {                 // odd tap count (block-uniform): they are in slot 1
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        a[0][mi][0] = a[1][mi][0];
        a[0][mi][1] = a[1][mi][1];
      }
    }
    CONV_DBG_DO(if (P.dbg) { const long long t_ = __builtin_readcyclecounter(); dbg_tap += t_ - dbg_x; dbg_x = t_; })
    if (c + 1 < nchunks) store_chunk((c + 1) & 1);
    __syncthreads();
    CONV_DBG_DO(if (P.dbg) dbg_sync += __builtin_readcyclecounter() - dbg_x;)
  }
  CONV_DBG_DO(if (P.dbg && blockIdx.x == 0 && lane == 0) {
    P.dbg[wave * 8 + 1] = dbg_tap;
    P.dbg[wave * 8 + 2] = dbg_sync;
    P.dbg[wave * 8 + 3] = __builtin_readcyclecounter() - dbg_t0;
  })

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int lenb = (P.out_mask || EPI == EPI_RESSKIP || EPI == EPI_COUPLE) ? P.len[b] : 0x7fffffff;
  // general forms: 4 accumulator elements at a time -- 16-wide with 64-bit per-element offsets needs >100 live registers (offsets,
  // rows, residuals) and would set the whole kernel's allocation, i.e. its occupancy.  (The plain STORE form below is 16-wide on
  // buffer addressing: one lane offset + scalar row offsets.)
  if (EPI == EPI_GATE) {
    // packed m-blocks alternate [tanh 32 rows | sigmoid 32 rows] of the same 32 channels
    const int j = (m0 >> 6) + wm;  // channel block of 32
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {
        float at[4], as[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { at[i] = acc[0][ni][e0 + i]; as[i] = acc[MI - 1][ni][e0 + i]; }
        conv_epilogue_gate<4>(P, G, b, j * 32 + 4 * h, e0, n0 + wn * (NI * 32) + ni * 32 + l31, at, as);
      }
    BT_BLOCK_END();
    return;
  }
  if (EPI == EPI_STORE && conv_epilogue_store_fast_ok(P, G)) {  // block-uniform
    conv_epilogue_store_fragments<MI, NI>(P, G, b, lenb, m0 + wm * MI * 32, n0 + wn * (NI * 32), h, l31, acc);
    CONV_DBG_DO(if (P.dbg && blockIdx.x == 0 && lane == 0) P.dbg[wave * 8 + 4] = __builtin_readcyclecounter() - dbg_t0;)
    BT_BLOCK_END();
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][e0 + i];
        conv_epilogue_frag<EPI, 4>(P, G, b, lenb, m0 + (wm * MI + mi) * 32 + 4 * h, e0, n0 + wn * (NI * 32) + ni * 32 + l31, v);
      }
    }
  }
  CONV_DBG_DO(if (P.dbg && blockIdx.x == 0 && lane == 0) P.dbg[wave * 8 + 4] = __builtin_readcyclecounter() - dbg_t0;)
  BT_BLOCK_END();
}

// ---------------------------------------------------------------------------------------------
// Small-N variant ("K-split"): one workgroup owns a (MI*32) x (NI*32) output tile and its 4 waves
// split the CONTRACTION (input-channel chunks c = wave, wave+4, ...).  In this regime (one
// utterance: N = 50..2400 columns) a tile is touched by one workgroup only, so LDS staging buys
// no reuse and its instruction cost dwarfs the MFMAs (measured: ~3000 cycles of staging per
// 512 cycles of MFMA for a 1-tap conv).  Instead every wave loads its B fragments STRAIGHT from
// global/L1 — the 32x32x2 B layout (lane -> row k = lane>>5, column lane&31) is already coalesced
// along time — double-buffered in registers one tap ahead, next to a 3-deep ring of weight
// fragments.  No LDS and no barrier in the main loop; the four partial accumulators meet in LDS
// once, then wave w finishes rows e in [4w, 4w+4) of every fragment through the shared epilogue.
// ---------------------------------------------------------------------------------------------
// uniform base pointer + per-lane 32-bit byte offset -> global_load_dword v, v_off, s[base:base+1]
__device__ __forceinline__ float ks_ld(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
struct ks_true { static constexpr bool value = true; };
struct ks_false { static constexpr bool value = false; };
// NW = waves per workgroup (4, 8 or 16) that split the contraction at TAP granularity (tap q = chunk*K + kk belongs
// to wave q % NW).  More waves = shorter serial MFMA chains per wave and 2..4 waves per SIMD to hide the issue latency
// of the fragment loads; the register budget per wave shrinks accordingly (KS_U taps per pipeline stage).
template <int MI, int NI, int EPI, int NIN, int NW>
__global__ void __launch_bounds__(NW * 64) conv_mfma_ks_kernel(const ConvParams P) {
  constexpr int M_T = MI * 32;
  constexpr int N_T = NI * 32;
  constexpr int NE = 16 / NW;  // accumulator elements each wave finishes after the reduction
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps chunk/address math scalar
  const int h = lane >> 5, l31 = lane & 31;
  CONV_DBG(0);
  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  const ConvGroup& G = P.g[grp];

  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  int tap_base = 0;
  if (P.ups_u) tap_base = P.ups_shift[m0 / P.ups_cout];
  const int nchunks = P.Cin / CONV_CI_T;
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }
  if (P.rag) {
    const int rl = P.rag[b], rc = P.rag[P.B];  // rag[B]: where the padded batch tensor ends (frames): no limit reaches beyond it
    if (n0 >= conv_rag_limit(rl, rc, P.rag_out_mul, P.rag_out_add, P.rag_out_cap_add)) return;  // whole tile is padding of this item (block-uniform)
    const int il = conv_rag_limit(rl, rc, P.rag_in_mul, P.rag_in_add);
    t_lim = il < t_lim ? il : t_lim;
  }
  if (P.skip_len && n0 >= P.len[b]) return;  // masked stage: the whole tile lies in this item's padding
  const float in_scale = P.in_scale, in_slope = P.in_slope;

  // lane geometry.  Input addresses are  uniform_base(b, chunk, p)  +  lane_off(h, t)  with the
  // lane part a non-negative 32-bit element offset, so the loads use the SGPR-base + VGPR-offset form.
  int tcol[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) tcol[ni] = n0 + ni * 32 + l31 - G.pad_l + tap_base;
  const long long rs = (long long)P.x_ch_sign * P.Tin_stride;  // one input channel, in elements (may be < 0)
  const long long ubase = (long long)b * P.x_bstride + (long long)P.x_ch_off * P.Tin_stride + (rs < 0 ? rs : 0);
  const unsigned hoff = (unsigned)(h ? (rs < 0 ? 0 : rs) : (rs < 0 ? -rs : 0));  // row k = lane>>5
  const float* xu = G.x + ubase;
  const float* xu2 = G.x2 ? G.x2 + ubase : nullptr;
  const float* xu3 = G.x3 ? G.x3 + ubase : nullptr;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int n_mblocks = P.M >> 5;
  const f32x4* wp[MI];  // uniform per wave; + lane
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + mi;
    if (mb >= n_mblocks) mb = 0;
    wp[mi] = reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64;
  }
  const int total_taps = nchunks * K;
  const int my_taps = wave < total_taps ? (total_taps - wave + NW - 1) / NW : 0;
  const int q_last = wave + NW * (my_taps > 0 ? my_taps - 1 : 0);  // this wave's last tap (clamp target of dead prefetches)
  const int c_last = q_last / K, k_last = q_last - c_last * K;

  // One "tap" = 8 MFMA k-steps of one (chunk, kk).  Taps are processed in super-steps of KS_U taps,
  // double-buffered in registers: while super-step s computes, the A and B fragments of super-step
  // s+1 are in flight.  Every load is unconditional with clamped indices so hipcc can emit counted s_waitcnt.
  constexpr int KS_U = NW >= 16 ? 1 : (NW == 8 ? (MI * NI > 1 ? 2 : 4) : 4);  // 4 waves per SIMD leave 128 registers per wave
  struct TapBuf {
    f32x4 a[KS_U][MI][2];
    float bq[KS_U][8][NI];
    unsigned ok[KS_U];
  };
  int lc = wave / K, lk = wave - (wave / K) * K;  // load cursor (chunk, tap-in-chunk) of this wave's tap stream
  int lq = wave;
  const int step_c = NW / K, step_k = NW - step_c * K;  // cursor advance of NW taps, one division per wave
  const unsigned lane16 = (unsigned)lane * 16u;
  const float* xq2 = NIN > 2 ? xu2 : xu;               // NIN == 3: MRF mean of three inputs (x3 may be absent)
  const float* xq3 = NIN > 2 ? (xu3 ? xu3 : xu2) : xu;
  const float s3 = (NIN > 2 && xu3) ? 1.f : 0.f;
  const int split_chunk = NIN == 2 ? P.x_split / CONV_CI_T : 0x7fffffff;  // NIN == 2: channel-concatenated second input
  const int refl_t = P.reflect ? ((P.Tin > 1) ? 1 : 0) : -1;

  // Software pipeline inside ONE wave, pinned at source level: before every MFMA of the current
  // super-step the wave issues one slice of the NEXT super-step's loads (weights at p == 0, one B
  // fragment per k-step), and a sched_barrier after each pair keeps hipcc from regrouping them.
  // All loads are unconditional (clamped + select), which keeps the body one basic block with counted s_waitcnt.
  auto pipe_step = [&](auto has_cur, auto act, const TapBuf& cur, TapBuf& nxt) {
#pragma unroll
    for (int u = 0; u < KS_U; ++u) {
      const bool live = lq <= q_last;
      const int cc = live ? lc : c_last;
      const int kc = live ? lk : k_last;
      const int sg = 2 * (cc * K + kc);
      const bool second = NIN == 2 && cc >= split_chunk;  // wave-uniform
      const float* xsrc = second ? xu2 : xu;
      const long long coff = (long long)(second ? cc - split_chunk : cc) * CONV_CI_T * rs;
      unsigned lo[NI];
      unsigned okb = 0;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        int t = tcol[ni] + kc * dil;
        t = (t == -1 && refl_t >= 0) ? refl_t : t;
        okb |= ((t >= 0) & (t < t_lim) & live) ? (1u << ni) : 0u;
        lo[ni] = (hoff + (unsigned)(t < 0 ? 0 : (t >= P.Tin ? P.Tin - 1 : t))) * 4u;  // BYTE offset: keeps the saddr + 32-bit voffset form
      }
      nxt.ok[u] = okb;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        nxt.a[u][mi][0] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(wp[mi] + (size_t)sg * 64) + lane16);
        nxt.a[u][mi][1] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(wp[mi] + (size_t)(sg + 1) * 64) + lane16);
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const long long ro = coff + 2 * p * rs;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float v = ks_ld(xsrc + ro, lo[ni]);
          if (NIN > 2) v = (v + ks_ld(xq2 + ro, lo[ni]) + s3 * ks_ld(xq3 + ro, lo[ni])) * in_scale;  // scale only on the MRF mean
          nxt.bq[u][p][ni] = v;
        }
        if (decltype(has_cur)::value) {
          bool okb_[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) okb_[ni] = (cur.ok[u] >> ni) & 1u;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            // leaky-relu for 0 <= slope <= 1 is max(v, slope*v); the edge/mask flag is a select (stale padding may hold
            // NaN).  in_slope == 1 (every encoder / flow conv) skips the activation.
            const float x_ = cur.bq[u][p][ni];
            const float bv = okb_[ni] ? (decltype(act)::value ? fmaxf(x_, x_ * in_slope) : x_) : 0.f;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[u][mi][p >> 2][p & 3], bv, acc[mi][ni], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // advance the cursor by NW taps (scalar)
      lq += NW;
      lk += step_k;
      lc += step_c + (lk >= K ? 1 : 0);
      lk -= lk >= K ? K : 0;
    }
  };

  CONV_DBG(1);
  auto run_pipe = [&](auto act) {
    TapBuf t0, t1;
    pipe_step(ks_false{}, act, t1, t0);
#pragma unroll 1
    for (int tp = 0; tp < my_taps; tp += 2 * KS_U) {
      pipe_step(ks_true{}, act, t0, t1);
      if (tp + KS_U < my_taps) pipe_step(ks_true{}, act, t1, t0);
    }
  };
  if (my_taps > 0) {
    if (in_slope == 1.f) run_pipe(ks_false{});
    else run_pipe(ks_true{});
  }
  CONV_DBG(2);

  // ---- cross-wave reduction through LDS
  float* red = lds;  // [wave][mi][ni][e][64]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(((wave * MI + mi) * NI + ni) * 16 + e) * 64 + lane] = acc[mi][ni][e];
  __syncthreads();
  CONV_DBG(3);
  const int lenb = (P.out_mask || EPI == EPI_RESSKIP || EPI == EPI_COUPLE) ? P.len[b] : 0x7fffffff;
  float sum[MI][NI][NE];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int ee = 0; ee < NE; ++ee) {
        const int e = NE * wave + ee;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) a += red[(((w * MI + mi) * NI + ni) * 16 + e) * 64 + lane];
        sum[mi][ni][ee] = a;
      }
  CONV_DBG(4);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + ni * 32 + l31;
    if (EPI == EPI_GATE) {
      conv_epilogue_gate<NE>(P, G, b, (m0 >> 6) * 32 + 4 * h, NE * wave, col, sum[0][ni], sum[MI - 1][ni]);
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) conv_epilogue_frag<EPI, NE>(P, G, b, lenb, m0 + mi * 32 + 4 * h, NE * wave, col, sum[mi][ni]);
    }
  }
  CONV_DBG(5);
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
