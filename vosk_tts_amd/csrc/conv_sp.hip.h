// conv_sp.hip.h — the 64 x 64 tile of conv_mfma.hip.h re-built as a SOFTWARE-PIPELINED loop for launches that put only one to three
// workgroups on a CU (round 5).
//
// Why (block traces, tools/bt_conv.py, profiles/r5_bt_64x64.txt): conv_mfma_kernel<2,2,1,1> gives a wave ONE 32 x 32 accumulator, so a tap
// is 8 dependent MFMAs (512 cycles) behind 8 LDS reads, behind a weight fragment that was requested one tap -- i.e. less than one L2
// round trip -- earlier, and every 16-channel chunk ends in a staging pass + barrier.  With 7 workgroups on a CU the other waves hide
// all of that (76 % of the MFMA rate inside a CU); with one or two (a coalesced batch of 8 requests: 288 workgroups; the text side of a
// 32-item batch: 660) a tap takes ~1400 cycles and the CU runs at 37-50 %.  Those launches are the middle third of every batch-size
// forward.
//
// What changes, same tile, same LDS layout, same operand formats, same epilogues:
//   * a stage is FOUR 16-channel chunks (64 channels x (64 + halo) columns in LDS): a quarter of the barriers and staging passes, and a
//     stage's tap count 4 K is a multiple of four whatever K is, so that
//   * the weight fragments live in a 4-slot ring with STATIC slot indices (taps in unrolled groups of four): a fragment is requested
//     three taps (>= 1500 MFMA cycles) before its use and nothing is copied (a rolled ring makes hipcc move the registers behind an
//     s_waitcnt vmcnt(0));
//   * the B fragments of tap t + 1 are read from LDS BEFORE the MFMAs of tap t (two register sets);
//   * the contraction alternates between two accumulators (two independent MFMA chains), summed once at the end;
//   * the next stage's activation loads go out after the first tap of a stage, in straight-line code of their own (group 0 is peeled),
//     so every s_waitcnt in the steady state is counted exactly.
// Eligibility (launch_conv): C_in a multiple of 64, one input tensor, no polyphase / reflection / channel split; everything else
// (masks, ragged tile maps, grouped launches with their own K / dilation, Flip-folded channel order, all four epilogues' operands) is
// taken from ConvParams exactly like the kernel it stands in for.  tests: test_conv1d_* (every kernel on random shapes), the stage and
// end-to-end parity tests run it wherever the heuristic picks it.
#pragma once
#include "conv_mfma.hip.h"

#define SP_STAGE_CH 64  // channels per stage (4 chunks of CONV_CI_T)
// LDS layout of a staged chunk: [column][16 channels in the order 0 2 4 .. 14 | 1 3 .. 15], column pitch SP_PITCH floats.  The B operand of
// the 32x32x2 MFMA wants, per lane (h = lane >> 5, column = lane & 31), channel 2 p + h at k-step p: with this order a lane's eight values
// are CONTIGUOUS -- two ds_read_b128 per tap instead of eight ds_read_b32 + eight address adds.  With one wave on a SIMD every instruction
// issued beside the MFMAs costs ~6 cycles that nothing hides (profiles/history: mfmaprobe; MI355X_MICROARCH.md "one extra issue slot"), and
// a tap of the [channel][column] layout carried ~40 of them for its 8 MFMAs: 1050 cycles per tap whatever was prefetched or interleaved
// (profiles/r5_conv_sp.txt).  Pitch 20: the 16 lanes of a ds_read_b128 group land on 16 distinct 4-bank groups (20 i mod 64).
#define SP_PITCH 20

// One 64 x 64 tile's contraction over the stages [s_begin, s_end) of 64 channels (the whole tile: 0 .. C_in / 64; stream-K segments,
// conv_sk.hip.h: any sub-range) -> the two accumulators of this wave.  false: the tile lies in an item's padding (block-uniform).
template <int JT>
__device__ __forceinline__ bool conv_sp_tile(const ConvParams& P, const ConvGroup& G, float* lds, int mt, int nt, int b, int s_begin, int s_end,
                                             f32x16 (&acc)[2]) {
  constexpr int N_T = 64, M_T = 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int ROW = P.row_len;
  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }
  if (P.rag) {
    const int rl = P.rag[b], rc = P.rag[P.B];
    if (n0 >= conv_rag_limit(rl, rc, P.rag_out_mul, P.rag_out_add, P.rag_out_cap_add)) return false;
    const int il = conv_rag_limit(rl, rc, P.rag_in_mul, P.rag_in_add);
    t_lim = il < t_lim ? il : t_lim;
  }
  if (P.skip_len && n0 >= P.len[b]) return false;

  // ---- staging: wave w owns stage rows w, w + 4, ..., w + 60; lanes stride over the ROW columns.  With one wave on a SIMD nothing hides
  // a staging pass (it was 30 % of such a workgroup's time, profiles/r5_conv_sp.txt), so the per-element work is cut to what the tile
  // needs (block-uniform): no activation arithmetic where the input activation is the identity (every encoder / flow conv), no validity
  // select in the interior of a sequence.
  float stg[16][JT];
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float in_scale = P.in_scale, in_slope = P.in_slope;
  const int t_base = n0 - G.pad_l;
  CONV_STAGE_COLS(JT)
  unsigned tob[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) tob[j] = (unsigned)toff[j] * 4u;
  const bool plain = in_scale == 1.f && in_slope == 1.f;                                   // block-uniform
  const bool interior = t_base >= 0 && t_base + ROW <= (t_lim < P.Tin ? t_lim : P.Tin);     // block-uniform: every staged column is real data
  const __amdgpu_buffer_rsrc_t rx = bt_rsrc(xb);
  const int row_step = 4 * P.x_ch_sign * P.Tin_stride * 4;  // bytes between the rows a wave owns
  auto load_stage = [&](int s) {
    unsigned roff = (unsigned)((long long)(P.x_ch_off + (s * SP_STAGE_CH + wave) * P.x_ch_sign) * P.Tin_stride * 4);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
#pragma unroll
      for (int j = 0; j < JT; ++j) stg[rr][j] = bt_ld(rx, tob[j], roff);
      roff += (unsigned)row_step;
    }
  };
  const int chunk_f = ROW * SP_PITCH;          // floats per staged chunk
  const int buf_f = 4 * chunk_f;               // floats per stage buffer
  // LDS byte address of (chunk, channel r = wave + 4 (rr & 3), column) = chunk * chunk_f * 4 + column * 80 + perm(r) * 4 with
  // perm(r) = 8 (r & 1) + (r >> 1) = [8 (wave & 1) + (wave >> 1)] + 2 (rr & 3): one base per (chunk, column group), the rest is an immediate
  unsigned st_base[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) st_base[j] = (unsigned)((lane + 64 * j) * SP_PITCH + 8 * (wave & 1) + (wave >> 1)) * 4u;
  auto store_stage = [&](int buf) {
    char* base = reinterpret_cast<char*>(lds + buf * buf_f);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      char* cb = base + (size_t)(rr >> 2) * chunk_f * 4 + 8 * (rr & 3);
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        float v = stg[rr][j];
        if (!plain) v = conv_act_in(v, in_scale, in_slope);
        if (!interior) v = tok[j] ? v : 0.f;
        if (j < JT - 1 || col < ROW) *reinterpret_cast<float*>(cb + st_base[j]) = v;
      }
    }
  };

#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }

  const int n_mblocks = P.M >> 5;
  int mb = (m0 >> 5) + wm;
  if (mb >= n_mblocks) mb = 0;  // padded tile: compute on valid memory, never stored
  const __amdgpu_buffer_rsrc_t wp = bt_rsrc(reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int n_sg = G.n_sg;

  // ---- weight ring: slot u holds tap (4 g + u); taps 0..2 requested here, tap q + 3 at tap q
  f32x4 a[4][2];
  int sg = s_begin * 8 * K;  // step-group of the next tap to request (2 per tap, 4 K taps per stage)
  auto request = [&](auto SLOT) {
    constexpr int slot = decltype(SLOT)::value;
    const int sgc = sg < n_sg ? sg : n_sg - 2;  // clamped: a fixed number of loads per tap keeps the waits counted
    a[slot][0] = bt_ld4(wp, lane16, (unsigned)sgc * 1024u);
    a[slot][1] = bt_ld4(wp, lane16, (unsigned)sgc * 1024u + 1024u);
    sg += 2;
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>; using S3 = std::integral_constant<int, 3>;
  request(S0{}); request(S1{}); request(S2{});
  load_stage(s_begin);
  store_stage(s_begin & 1);
  __syncthreads();

  f32x4 bv[2][2];
  const float* lb = nullptr;  // this lane's eight B values of column (tile column + tap shift 0) of the stage's first chunk
  int kk = 0, jrow = 0;       // tap inside the chunk / float offset of the chunk inside the stage
  // B fragments of the tap at (jrow, kk) -- or, for the tap after a stage's last one (`valid` false, wave-uniform), a harmless re-read of
  // the stage's first element: no branch, so a tap stays ONE scheduling region
  auto read_b = [&](auto SET, bool valid) {
    constexpr int set = decltype(SET)::value;
    const f32x4* lk = reinterpret_cast<const f32x4*>(lb + (valid ? jrow + kk * dil * SP_PITCH : 0));
    bv[set][0] = lk[0];
    bv[set][1] = lk[1];
  };
  auto advance = [&]() {
    ++kk;
    if (kk == K) { kk = 0; jrow += chunk_f; }
  };
  // One tap on weight slot U and B set (U & 1).  It also requests tap + 3's weight fragments into slot (U + 3) & 3, reads the NEXT tap's
  // B fragments and (between(): first tap of a stage) issues the next stage's activation loads -- all of it INTERLEAVED with the tap's
  // eight MFMAs by sched_group_barrier: with one wave on a SIMD nothing else fills the matrix pipe while those ~40 scalar / vector /
  // LDS / memory instructions issue.  VM = memory reads to place per MFMA gap (the two weight loads, or those + the 16 activation loads).
  auto tap = [&](auto U, auto VM, bool more_in_stage, auto&& between) {
    constexpr int u = decltype(U)::value;
    constexpr int vm = decltype(VM)::value;
    request(std::integral_constant<int, (u + 3) & 3>{});
    advance();
    read_b(std::integral_constant<int, (u + 1) & 1>{}, more_in_stage);
    between();
#pragma unroll
    for (int p = 0; p < 8; ++p)
      acc[p & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][p >> 2][p & 3], bv[u & 1][p >> 2][p & 3], acc[p & 1], 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);   // scalar address / step-group arithmetic
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // LDS address
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // B-fragment reads of the next tap (two per tap)
      __builtin_amdgcn_sched_group_barrier(0x020, vm, 0);  // weight / activation loads
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto nothing = []() {};
  using VW = std::integral_constant<int, 1>;                 // 2 weight loads over 8 gaps
  using VS = std::integral_constant<int, (16 * JT + 2 + 7) / 8>;  // + the next stage's 16 JT activation loads

  const int ngroups = K;  // 4 K taps per stage, in groups of four
  for (int s = s_begin; s < s_end; ++s) {
    lb = lds + (s & 1) * buf_f + (wn * 32 + l31) * SP_PITCH + h * 8;
    kk = 0; jrow = 0;
    read_b(S0{}, true);
    // group 0, peeled: the next stage's activation loads go out with its first tap (straight-line code: exactly counted waits)
    const bool next = s + 1 < s_end;
    if (next) tap(S0{}, VS{}, true, [&]() { load_stage(s + 1); });
    else tap(S0{}, VW{}, true, nothing);
    tap(S1{}, VW{}, true, nothing);
    tap(S2{}, VW{}, true, nothing);
    tap(S3{}, VW{}, ngroups > 1, nothing);
#pragma unroll 1
    for (int g = 1; g < ngroups; ++g) {
      tap(S0{}, VW{}, true, nothing);
      tap(S1{}, VW{}, true, nothing);
      tap(S2{}, VW{}, true, nothing);
      tap(S3{}, VW{}, g + 1 < ngroups, nothing);
    }
    if (next) store_stage((s + 1) & 1);
    __syncthreads();
  }

  return true;
}

// the tile's epilogue on the summed accumulator (shared with conv_mfma_kernel)
template <int EPI>
__device__ __forceinline__ void conv_sp_epilogue(const ConvParams& P, const ConvGroup& G, int mt, int nt, int b, f32x16 (&accs)[1][1]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int n0 = nt * 64, m0 = mt * 64;
  const int lenb = (P.out_mask || EPI == EPI_RESSKIP || EPI == EPI_COUPLE) ? P.len[b] : 0x7fffffff;
  if (EPI == EPI_STORE && conv_epilogue_store_fast_ok(P, G)) {
    conv_epilogue_store_fragments<1, 1>(P, G, b, lenb, m0 + wm * 32, n0 + wn * 32, h, l31, accs);
  } else {
#pragma unroll
    for (int e0 = 0; e0 < 16; e0 += 4) {
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = accs[0][0][e0 + i];
      conv_epilogue_frag<EPI, 4>(P, G, b, lenb, m0 + wm * 32 + 4 * h, e0, n0 + wn * 32 + l31, v);
    }
  }
}

template <int EPI, int JT>
__global__ void __launch_bounds__(256, 3) conv_sp_kernel(const ConvParams P) {
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  const int tid = threadIdx.x;
  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  mt = __builtin_amdgcn_readfirstlane(mt); grp = __builtin_amdgcn_readfirstlane(grp);
  nt = __builtin_amdgcn_readfirstlane(nt); b = __builtin_amdgcn_readfirstlane(b);
  const ConvGroup& G = P.g[grp];
  CONV_DBG_DO(if (P.dbg && tid == 0 && blockIdx.x < 4000) {
    P.dbg[128 + blockIdx.x * 4 + 0] = wall_clock64();
    P.dbg[128 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492);
    P.dbg[128 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20);
  })
  f32x16 acc[2];
  if (!conv_sp_tile<JT>(P, G, lds, mt, nt, b, 0, P.Cin / SP_STAGE_CH, acc)) return;
  f32x16 accs[1][1];
#pragma unroll
  for (int e = 0; e < 16; ++e) accs[0][0][e] = acc[0][e] + acc[1][e];
  conv_sp_epilogue<EPI>(P, G, mt, nt, b, accs);
  CONV_DBG_DO(if (P.dbg && tid == 0 && blockIdx.x < 4000) P.dbg[128 + blockIdx.x * 4 + 1] = wall_clock64();)
  (void)tid;
}
