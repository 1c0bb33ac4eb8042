// conv_w1.hip.h — the batch-size ResBlock / STORE conv as INDEPENDENT WAVES (round 6): one wave = one 64 x 64 output tile, its own
// staged activation window in its own 8 KB of LDS, no workgroup barrier anywhere.
//
// Why (profiles/r6_bt_dom.txt): conv_mfma_kernel<2,2,2,2> gives four waves one 128 x 128 tile and one shared LDS window per 16-channel
// chunk, so every chunk ends in `store_chunk; s_barrier`.  A block trace of a dense 32-item launch (768 workgroups, three per CU, all
// resident at once) shows each wave spending 20 % of its main loop between the last tap of a chunk and the first of the next, and the
// three workgroups of a CU started together and stay in phase: while they stage / wait at their barriers the SIMD's matrix pipe has
// nobody to issue for (MFMA busy 0.80-0.83 of the launch; round 3 measured 0.88 inside the tap loops and named "a producer-consumer
// restructuring" as the remaining lever).  The wave tile (MI = NI = 2: 64 accumulator registers, weights two dwordx4 per m-block and tap,
// one tap ahead in two static slots) is unchanged; what changes is who a wave waits for: nobody.
//   * a 64-thread workgroup IS the wave: it stages ITS 16 channels x (64 + halo) columns into its own LDS rows (fixed pitch W1_PITCH, so
//     every B-fragment read is base + immediate), LDS operations of one wave execute in order, so the stores of chunk c + 1 simply follow
//     the reads of chunk c -- no barrier, no double buffer;
//   * the chunk's global loads go out behind tap 0 (as in the big-tile kernel) and ride in registers through the tap loop;
//   * up to twelve such waves share a CU (three per SIMD), each at its own phase: a wave's staging pass, prologue and epilogue are
//     covered by the other two waves of its SIMD as long as they have MFMAs to issue -- and nothing synchronises them.
// Cost: the window a four-wave workgroup staged once (128 + halo columns) is staged by four waves as 4 x (64 + halo) columns (2.6 x the
// L2 -> LDS activation traffic at halo 50; weights: unchanged, every wave streamed its own 64 rows already).
// Eligibility (launch_conv): EPI_STORE, one input tensor, no polyphase / reflection / channel split, M a multiple of 64, 64 + halo <=
// W1_PITCH.  Everything else (grouped launches, ragged tile maps, masks, Flip-folded channel order, the epilogue's operands) comes from
// ConvParams exactly as for the kernel it stands in for.
// Reference ops served: modules.py:190-223 (ResBlock1 convs), attentions.py:292-320 (FFN convs), modules.py:148-176 at batch size.
#pragma once
#include "conv_mfma.hip.h"

#define W1_PITCH 128  // floats per staged channel row (64 columns + halo <= 64)

template <int EPI, int JT>
__global__ void __launch_bounds__(64, 3) conv_w1_kernel(const ConvParams P) {
  constexpr int MI = 2, NI = 2, M_T = 64, N_T = 64;
  extern __shared__ float lds[];  // [CONV_CI_T][W1_PITCH]
  kernarg_warm<sizeof(ConvParams)>();
  const int lane = threadIdx.x, h = lane >> 5, l31 = lane & 31;

  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  mt = __builtin_amdgcn_readfirstlane(mt); grp = __builtin_amdgcn_readfirstlane(grp);
  nt = __builtin_amdgcn_readfirstlane(nt); b = __builtin_amdgcn_readfirstlane(b);
  const ConvGroup& G = P.g[grp];
  CONV_DBG_DO(if (P.dbg && lane == 0 && blockIdx.x < 4000) {
    P.dbg[128 + blockIdx.x * 4 + 0] = wall_clock64();
    P.dbg[128 + blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(63492);
    P.dbg[128 + blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 20);
  })

  const int ROW = P.row_len;  // 64 + halo
  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  const int nchunks = P.Cin / CONV_CI_T;
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }
  if (P.rag) {
    const int rl = P.rag[b], rc = P.rag[P.B];
    if (n0 >= conv_rag_limit(rl, rc, P.rag_out_mul, P.rag_out_add, P.rag_out_cap_add)) return;
    const int il = conv_rag_limit(rl, rc, P.rag_in_mul, P.rag_in_add);
    t_lim = il < t_lim ? il : t_lim;
  }
  if (P.skip_len && n0 >= P.len[b]) return;

  // ---- staging: the wave owns all 16 rows of a chunk; lanes stride over the ROW columns
  float stg[CONV_CI_T][JT];
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float in_scale = P.in_scale, in_slope = P.in_slope;
  const int t_base = n0 - G.pad_l;
  CONV_STAGE_COLS(JT)
  unsigned tob[JT];
#pragma unroll
  for (int j = 0; j < JT; ++j) tob[j] = (unsigned)toff[j] * 4u;
  const bool plain = in_scale == 1.f && in_slope == 1.f;                                 // block-uniform
  const bool interior = t_base >= 0 && t_base + ROW <= (t_lim < P.Tin ? t_lim : P.Tin);   // block-uniform: every staged column is real data
  const __amdgpu_buffer_rsrc_t rx = bt_rsrc(xb);
  const int row_step = P.x_ch_sign * P.Tin_stride * 4;  // bytes between consecutive channels
  auto load_chunk = [&](int c) {
    unsigned roff = (unsigned)((long long)(P.x_ch_off + c * CONV_CI_T * P.x_ch_sign) * P.Tin_stride * 4);
#pragma unroll
    for (int rr = 0; rr < CONV_CI_T; ++rr) {
#pragma unroll
      for (int j = 0; j < JT; ++j) stg[rr][j] = bt_ld(rx, tob[j], roff);
      roff += (unsigned)row_step;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int rr = 0; rr < CONV_CI_T; ++rr) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int col = lane + 64 * j;
        float v = stg[rr][j];
        if (!plain) v = conv_act_in(v, in_scale, in_slope);
        if (!interior) v = tok[j] ? v : 0.f;
        if (j < JT - 1 || col < ROW) lds[rr * W1_PITCH + col] = v;
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
  __amdgpu_buffer_rsrc_t wp[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + mi;
    if (mb >= n_mblocks) mb = 0;  // padded tile: compute on valid memory, never stored
    wp[mi] = bt_rsrc(reinterpret_cast<const f32x4*>(G.w) + (size_t)mb * G.n_sg * 64);
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  const int n_sg = G.n_sg;

  // weight fragments: two static slots, one tap ahead, taps in pairs (see conv_mfma_kernel)
  f32x4 a[2][MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    a[0][mi][0] = bt_ld4(wp[mi], lane16, 0);
    a[0][mi][1] = bt_ld4(wp[mi], lane16, 1024);
  }
  int sg = 2;
  load_chunk(0);
  store_chunk();

  const float* lb = lds + h * W1_PITCH + l31;
  auto tap = [&](auto CUR, int kk) {
    constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
    {
      const int sgc = sg < n_sg ? sg : n_sg - 2;  // clamped: a fixed number of loads per tap keeps the waits counted
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        a[nxt][mi][0] = bt_ld4(wp[mi], lane16, (unsigned)sgc * 1024u);
        a[nxt][mi][1] = bt_ld4(wp[mi], lane16, (unsigned)sgc * 1024u + 1024u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    sg += 2;
    const float* lk = lb + kk * dil;
    float bv[8][NI];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bv[p][ni] = lk[2 * p * W1_PITCH + ni * 32];
    __builtin_amdgcn_sched_barrier(0);
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
    tap(slot0{}, 0);
    if (c + 1 < nchunks) load_chunk(c + 1);  // behind tap 0: the next wait that covers these loads is the one before tap 1
    __builtin_amdgcn_sched_barrier(0);
    int kk = 1;
#pragma unroll 1
    for (; kk + 1 < K; kk += 2) {
      tap(slot1{}, kk);
      tap(slot0{}, kk + 1);
    }
    if (kk < K) {
      tap(slot1{}, kk);
    } else {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        a[0][mi][0] = a[1][mi][0];
        a[0][mi][1] = a[1][mi][1];
      }
    }
    // the wave's own LDS operations execute in order: these stores follow every read of chunk c, the reads of chunk c + 1 follow them
    if (c + 1 < nchunks) store_chunk();
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue (shared with conv_mfma_kernel)
  const int lenb = (P.out_mask || EPI == EPI_RESSKIP || EPI == EPI_COUPLE) ? P.len[b] : 0x7fffffff;
  if (EPI == EPI_STORE && conv_epilogue_store_fast_ok(P, G)) {
    conv_epilogue_store_fragments<MI, NI>(P, G, b, lenb, m0, n0, h, l31, acc);
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e0 = 0; e0 < 16; e0 += 4) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][e0 + i];
          conv_epilogue_frag<EPI, 4>(P, G, b, lenb, m0 + mi * 32 + 4 * h, e0, n0 + ni * 32 + l31, v);
        }
  }
  CONV_DBG_DO(if (P.dbg && lane == 0 && blockIdx.x < 4000) P.dbg[128 + blockIdx.x * 4 + 1] = wall_clock64();)
}
