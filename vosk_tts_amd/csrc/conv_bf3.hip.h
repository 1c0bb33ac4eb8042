// Split-bf16 ("bf16x3") variant of the big-tile conv kernel: BASELINE configs[2] allows a reduced-precision acoustic path; this is
// the form that keeps fp32-class accuracy.  Every fp32 operand is split once into two bf16 pieces, x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi), and a product is evaluated as  hi*hi + hi*lo + lo*hi  on the bf16 matrix core with fp32
// accumulation (the dropped lo*lo term is 2^-16 relative; measured on one tile, K = 2304: 3.6e-6 of max|C| against fp64 vs 1.25e-6
// for the fp32 MFMA, tools/bf16x3probe.hip) -- 3 v_mfma_f32_32x32x16_bf16 (16 channels x 1 tap each) instead of 8
// v_mfma_f32_32x32x2_f32, ~7x the issue rate.
//
// Same implicit GEMM as conv_mfma_kernel<2,2,2,2> (128 x 128 tile, 4 waves of 64 x 64, 16-channel chunks staged once per chunk in
// LDS and re-read for every tap at a shifted column, weights streamed from L2 in fragment order one tap ahead), with two changes:
//   * weights are split and packed at load time: [m-block of 32][chunk][tap][piece][lane][8 bf16]  (pack_conv_weights_bf3);
//   * the activation split is paid ONCE per element in the staging pass (not per fragment per wave: that costs more VALU than the
//     MFMAs it feeds): the chunk is written channel-fastest, [piece][column][16 channels as bf16, pitch 48 B], so a lane's B
//     fragment (8 consecutive channels of one column) is one ds_read_b128, bank-conflict free at any tap shift.
// The k index inside a chunk is permuted so that the staging thread that holds channels {w, w+4, w+8, w+12} of a column (wave w: the
// coalesced load pattern of the fp32 kernel) writes them as 4 consecutive bf16: position p = 4w + r  <->  channel w + 4r.
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define BF3_PITCH 24  // bf16 elements per staged column (16 used): 48 B, 16 consecutive columns hit 16 distinct 16-byte bank groups

static inline uint16_t bf3_rne(float x) {  // round-to-nearest-even fp32 -> bf16 bits (finite inputs)
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf3_f32(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// dst: Mpad/32 * Cin/16 * K * 2 * 64 * 8 bf16 (= Mpad * Cin * K * 4 bytes, the size of the fp32 packing)
template <typename F>
static void pack_conv_weights_bf3(uint16_t* dst, int Mpad, int Cin, int K, F src /* float(int row,int ci,int kk) */) {
  const int nch = Cin / CONV_CI_T;
  for (int mb = 0; mb < Mpad / 32; ++mb)
    for (int c = 0; c < nch; ++c)
      for (int kk = 0; kk < K; ++kk)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int p = 8 * (lane >> 5) + e;
            const int ci = c * CONV_CI_T + (p >> 2) + 4 * (p & 3);
            const float w = src(mb * 32 + (lane & 31), ci, kk);
            const uint16_t hi = bf3_rne(w), lo = bf3_rne(w - bf3_f32(hi));
            const size_t base = (((size_t)mb * nch + c) * K + kk) * 2;
            dst[((base + 0) * 64 + lane) * 8 + e] = hi;
            dst[((base + 1) * 64 + lane) * 8 + e] = lo;
          }
}

template <int V> struct bf3_int { static constexpr int value = V; };
// At 32 cycles per MFMA a SIMD has ~8 issue slots per MFMA for ALL of its waves (MI355X_MICROARCH.md): a first version with the
// fp32 kernel's rolled tap loop (16 v_mov per tap to rotate the weight registers, clamped 64-bit weight addresses, loop
// bookkeeping) left the matrix pipe 47 % busy even with every load removed.  Here taps are processed in PAIRS so that the two
// weight-fragment slots are named statically (slot = tap parity; an odd tap count costs one 16-register move per CHUNK instead of
// one per tap), one pointer per m-block advances by a constant, and the B reads use
// immediate offsets.  The weight stream is read one step past its end (add_bf3_packing pads the array).
// MI = 32-row blocks per wave: 2 -> 128 x 128 tiles, 1 -> 64 x 128 tiles (64-channel stages).  EPI: EPI_STORE, or EPI_GATE with MI == 2
// (a wave's two row blocks are the [tanh 32 | sigmoid 32] pre-activations of the same 32 channels, commons.py:100-107)
// PC (producer / consumer split): the workgroup has 6 waves.  Waves 4 and 5 only stage -- each owns 8 of a chunk's 16 channels
// (one 16-byte half of every staged column), loads them two chunks ahead, splits and stores them; waves 0..3 only stream weights,
// read LDS and issue MFMAs.  vmcnt retires in order: in the 4-wave form every weight wait that follows the activation loads of
// chunk c + 2 also waits for those (an HBM round trip once per chunk, measured as 26 % of the ResBlock launches' time with the
// staging compiled out, profiles/r3_bf3_ab.txt); here the consumers' counter only ever holds weight fragments.
// NS = weight-fragment slots: 2 (a step is requested one tap ahead) or 3 (two taps ahead, slot = (phase + tap) mod 3 with the chunk
// body instantiated per phase; the activation loads are then requested ONE chunk ahead straight into the single staging register
// set, at the top of the chunk, where the weight fragments of the chunk's first two taps are already older than them).
template <int MI, int EPI, bool PC = false, int NS = 2>
static __device__ __forceinline__ void conv_bf3_body(const ConvParams& P, const ConvGroup& G, float* lds, int mt, int nt, int b) {
  constexpr int N_T = 128, M_T = 64 * MI;
  constexpr int JT = (N_T + CONV_MAX_HALO + 63) / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row / weight-block offsets stay scalar
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int h = lane >> 5, l31 = lane & 31;
  const int ROW = P.row_len;  // staged columns: 128 + the launch's largest halo
  const int n0 = nt * N_T, m0 = mt * M_T;
  const int K = G.K, dil = G.dil;
  const int nchunks = P.Cin / CONV_CI_T;
  const int tap_base = P.ups_u ? P.ups_shift[m0 / P.ups_cout] : 0;  // polyphase ConvTranspose1d rows: the tile's phase shifts its taps
  int t_lim = P.Tin;
  if (P.in_mask) { int lb = P.len[b]; t_lim = lb < t_lim ? lb : t_lim; }
  if (P.rag) {
    const int rl = P.rag[b], rc = P.rag[P.B];  // rag[B]: where the padded batch tensor ends (frames): no limit reaches beyond it
    if (n0 >= conv_rag_limit(rl, rc, P.rag_out_mul, P.rag_out_add, P.rag_out_cap_add)) return;  // whole tile is padding of this item (block-uniform)
    const int il = conv_rag_limit(rl, rc, P.rag_in_mul, P.rag_in_add);
    t_lim = il < t_lim ? il : t_lim;
  }
  if (P.skip_len && n0 >= P.len[b]) return;

  const int piece_bytes = ROW * (BF3_PITCH * 2);  // one piece (hi or lo) of one chunk buffer
  if constexpr (PC) {
    if (wave >= 4) {
      // ---- producer wave pw: positions 8 pw .. 8 pw + 7 of every staged column = channels (2 pw + (i >> 2)) + 4 (i & 3), i = 0..7
      const int pw = wave - 4;
      float stg[8][JT], stn[8][JT];
      const float* xb = G.x + (long long)b * P.x_bstride;
      // leaky ReLU of the scaled input as max(s x, (s slope) x): in_scale >= 0 and 0 <= in_slope <= 1 (launcher: bf3_ok), one compare
      // and one select less per element than conv_act_in
      const float in_scale = P.in_scale, in_ss = P.in_scale * P.in_slope;
      const int t_base = n0 - G.pad_l;
      CONV_STAGE_COLS(JT)
      unsigned tob[JT];
#pragma unroll
      for (int j = 0; j < JT; ++j) tob[j] = (unsigned)toff[j] * 4u;
      const __amdgpu_buffer_rsrc_t rx = bt_rsrc(xb);
      auto load_chunk = [&](int c, float (&dst)[8][JT]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned ro = (unsigned)((long long)(c * CONV_CI_T + 2 * pw + (i >> 2) + 4 * (i & 3)) * P.Tin_stride * 4);
#pragma unroll
          for (int j = 0; j < JT; ++j) dst[i][j] = bt_ld(rx, tob[j], ro);
        }
      };
      auto store_chunk = [&](int buf) {
        char* dst = reinterpret_cast<char*>(lds) + buf * (2 * piece_bytes) + 16 * pw;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const int col = lane + 64 * j;
          bf16x8 hi, lo;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float v = tok[j] ? fmaxf(stg[i][j] * in_scale, stg[i][j] * in_ss) : 0.f;  // select: stale padding may hold NaN
            hi[i] = (__bf16)v;
            lo[i] = (__bf16)(v - (float)hi[i]);
          }
          if (j < JT - 1 || col < ROW) {
            *reinterpret_cast<bf16x8*>(dst + col * (BF3_PITCH * 2)) = hi;
            *reinterpret_cast<bf16x8*>(dst + piece_bytes + col * (BF3_PITCH * 2)) = lo;
          }
        }
      };
      load_chunk(0, stg);
      store_chunk(0);
      if (nchunks > 1) load_chunk(1, stg);
      __syncthreads();
#pragma unroll 1
      for (int c = 0; c < nchunks; ++c) {
        if (c + 2 < nchunks) load_chunk(c + 2, stn);
        if (c + 1 < nchunks) store_chunk((c + 1) & 1);  // (waits for the loads of chunk c + 1 only: the newer ones stay in flight)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < JT; ++j) stg[i][j] = stn[i][j];
      }
      return;
    }
  }

  // ---- staging: wave w owns chunk rows w, w+4, w+8, w+12 (coalesced along time); lanes stride over columns
  float stg[4][JT], stn[4][JT];  // staged values of chunk c + 1 (loaded one chunk earlier) and in-flight loads of chunk c + 2
  const float* xb = G.x + (long long)b * P.x_bstride;
  const float in_scale = P.in_scale, in_ss = P.in_scale * P.in_slope;  // (see the producer path)
  const int t_base = n0 - G.pad_l;
  CONV_STAGE_COLS(JT)
  unsigned tob[JT];  // byte offsets of the staging columns (buffer addressing: descriptor + scalar row offset + tob, conv_mfma.hip.h bt_ld)
#pragma unroll
  for (int j = 0; j < JT; ++j) tob[j] = (unsigned)toff[j] * 4u;
  const __amdgpu_buffer_rsrc_t rx = bt_rsrc(xb);
  auto load_chunk = [&](int c, float (&dst)[4][JT]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const unsigned ro = (unsigned)((long long)(c * CONV_CI_T + wave + 4 * rr) * P.Tin_stride * 4);
#pragma unroll
      for (int j = 0; j < JT; ++j) dst[rr][j] = bt_ld(rx, tob[j], ro);
    }
    // (nothing here may consume the loaded values: they ride through the tap loop and are split only in store_chunk)
  };
  auto store_chunk = [&](int buf) {
    char* dst = reinterpret_cast<char*>(lds) + buf * (2 * piece_bytes) + 8 * wave;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int col = lane + 64 * j;
      bf16x4 hi, lo;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const float v = tok[j] ? fmaxf(stg[rr][j] * in_scale, stg[rr][j] * in_ss) : 0.f;  // select: stale padding may hold NaN
        hi[rr] = (__bf16)v;
        lo[rr] = (__bf16)(v - (float)hi[rr]);
      }
      if (j < JT - 1 || col < ROW) {  // 64 (JT - 1) <= N_T <= ROW: only the last column group needs the test
        *reinterpret_cast<bf16x4*>(dst + col * (BF3_PITCH * 2)) = hi;
        *reinterpret_cast<bf16x4*>(dst + piece_bytes + col * (BF3_PITCH * 2)) = lo;
      }
    }
  };

  f32x16 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // weight stream: step s = chunk * K + tap; per step and m-block: [hi | lo] x 64 lanes x 16 B
  const int n_mblocks = P.M >> 5;
  const int nsteps = nchunks * K;
  __amdgpu_buffer_rsrc_t wq[MI];  // descriptors of this wave's weight m-blocks: a fragment load is wq + step offset (scalar) + lane * 16
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    int mb = (m0 >> 5) + wm * MI + mi;
    if (mb >= n_mblocks) mb = 0;  // padded tile: compute on valid memory, never stored
    wq[mi] = bt_rsrc(reinterpret_cast<const bf16x8*>(G.wb) + (size_t)mb * nsteps * 128);
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  auto wload = [&](__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, lane16, off, 0)); };

  if constexpr (!PC) {
    load_chunk(0, stg);
    store_chunk(0);
    if (nchunks > 1) load_chunk(1, stg);
  }
  __syncthreads();

  bf16x8 a[NS][MI][2];  // [slot][mi][piece]
#pragma unroll
  for (int sl = 0; sl < NS - 1; ++sl)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      a[sl][mi][0] = wload(wq[mi], sl * 2048);
      a[sl][mi][1] = wload(wq[mi], sl * 2048 + 1024);
    }
  unsigned ws = (NS - 1) * 2048;  // byte offset of the next step to fetch (2 KB per step and m-block; the packing is padded by two steps for the last prefetches)
  // one tap: prefetch the next step into the other slot, read this tap's B fragments, 12 MFMAs
  auto tap = [&](auto slot_, const char* lk0, const char* lk1) {
    constexpr int S = decltype(slot_)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      a[(S + NS - 1) % NS][mi][0] = wload(wq[mi], ws);
      a[(S + NS - 1) % NS][mi][1] = wload(wq[mi], ws + 1024);
    }
    ws += 2048;
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      bh[ni] = *reinterpret_cast<const bf16x8*>(lk0 + ni * 32 * (BF3_PITCH * 2));
      bl[ni] = *reinterpret_cast<const bf16x8*>(lk1 + ni * 32 * (BF3_PITCH * 2));
    }
    __builtin_amdgcn_sched_barrier(0);  // loads of the next step and this tap's B reads in flight before the first MFMA
    // lo*hi + hi*lo + hi*hi, the four accumulators interleaved (no back-to-back dependent MFMAs)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][mi][1], bh[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][mi][0], bl[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[S][mi][0], bh[ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (NS == 3) {
    static_assert(!PC, "three weight slots: 4-wave form only");
    const int dstep = dil * (BF3_PITCH * 2);
    const int kmod = K % 3;
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      // slot = tap mod 3 inside a chunk (static names); a chunk whose tap count is not a multiple of 3 ends with its successor's
      // first two steps in slots (K mod 3) and (K mod 3) + 1: rotated back to slots 0 and 1 below (32 moves per chunk, 7- and 11-tap
      // groups only)
      const char* lk0 = reinterpret_cast<const char*>(lds) + (c & 1) * (2 * piece_bytes) + (wn * 64 + l31 + tap_base) * (BF3_PITCH * 2) + 16 * h;
      const char* lk1 = lk0 + piece_bytes;
      tap(bf3_int<0>{}, lk0, lk1);
      lk0 += dstep; lk1 += dstep;
      int kk = 1;
#pragma unroll 1
      for (; kk + 2 < K; kk += 3) {
        tap(bf3_int<1>{}, lk0, lk1);
        tap(bf3_int<2>{}, lk0 + dstep, lk1 + dstep);
        tap(bf3_int<0>{}, lk0 + 2 * dstep, lk1 + 2 * dstep);
        lk0 += 3 * dstep; lk1 += 3 * dstep;
      }
      if (kk < K) {
        tap(bf3_int<1>{}, lk0, lk1);
        if (kk + 1 < K) tap(bf3_int<2>{}, lk0 + dstep, lk1 + dstep);
      }
      if (kmod == 1) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) { a[0][mi][pc] = a[1][mi][pc]; a[1][mi][pc] = a[2][mi][pc]; }
      } else if (kmod == 2) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int pc = 0; pc < 2; ++pc) { a[1][mi][pc] = a[0][mi][pc]; a[0][mi][pc] = a[2][mi][pc]; }
      }
      if (c + 1 < nchunks) store_chunk((c + 1) & 1);
      __syncthreads();
      if (c + 2 < nchunks) load_chunk(c + 2, stg);  // one chunk ahead, into the registers the store just freed
      __builtin_amdgcn_sched_barrier(0);
    }
  } else
#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    const int dstep = dil * (BF3_PITCH * 2);
    const char* lk0 = reinterpret_cast<const char*>(lds) + (c & 1) * (2 * piece_bytes) + (wn * 64 + l31 + tap_base) * (BF3_PITCH * 2) + 16 * h;
    const char* lk1 = lk0 + piece_bytes;
    // Tap 0 first, THEN the activation loads: vmcnt retires in order, so a wait for weight fragments requested behind the
    // activation loads waits for those too (conv_mfma.hip.h, same rule).  They are requested TWO chunks ahead: a chunk's taps
    // take 1.2 k (3 taps) .. 4.2 k (11 taps) MFMA cycles per wave, less than an HBM round trip under load.
    tap(bf3_int<0>{}, lk0, lk1);
    if constexpr (!PC) { if (c + 2 < nchunks) load_chunk(c + 2, stn); }
    __builtin_amdgcn_sched_barrier(0);
    lk0 += dstep;
    lk1 += dstep;
    int kk = 1;
#pragma unroll 1
    for (; kk + 1 < K; kk += 2) {
      tap(bf3_int<1>{}, lk0, lk1);
      tap(bf3_int<0>{}, lk0 + dstep, lk1 + dstep);
      lk0 += 2 * dstep;
      lk1 += 2 * dstep;
    }
    if (kk < K) {
      tap(bf3_int<1>{}, lk0, lk1);  // even tap count: the next chunk's first fragments are in slot 0 already
    } else {                        // odd tap count: they are in slot 1
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) a[0][mi][pc] = a[1][mi][pc];
    }
    if constexpr (!PC) { if (c + 1 < nchunks) store_chunk((c + 1) & 1); }
    __syncthreads();
    if constexpr (!PC) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int j = 0; j < JT; ++j) stg[rr][j] = stn[rr][j];
    }
  }

  // ---- epilogue (shared with the fp32 kernels).  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  if (EPI == EPI_GATE) {
    const int j = (m0 >> 6) + wm;  // channel block of 32
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {
        float at[4], as[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { at[i] = acc[0][ni][e0 + i]; as[i] = acc[MI - 1][ni][e0 + i]; }
        conv_epilogue_gate<4>(P, G, b, j * 32 + 4 * h, e0, n0 + wn * 64 + ni * 32 + l31, at, as);
      }
    return;
  }
  const int lenb = P.out_mask ? P.len[b] : 0x7fffffff;
  if (EPI == EPI_STORE && conv_epilogue_store_fast_ok(P, G)) {  // block-uniform: the fragment-wide epilogue of the fp32 kernel
    conv_epilogue_store_fragments<MI, 2>(P, G, b, lenb, m0 + wm * MI * 32, n0 + wn * 64, h, l31, acc);
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e0 = 0; e0 < 16; e0 += 4) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][e0 + i];
        conv_epilogue_frag<EPI == EPI_GATE ? EPI_STORE : EPI, 4>(P, G, b, lenb, m0 + (wm * MI + mi) * 32 + 4 * h, e0, n0 + wn * 64 + ni * 32 + l31, v);
      }
}

template <int MI, int EPI>
__global__ void __launch_bounds__(384, 2) conv_bf3pc_kernel(const ConvParams P) {
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  mt = __builtin_amdgcn_readfirstlane(mt); grp = __builtin_amdgcn_readfirstlane(grp);
  nt = __builtin_amdgcn_readfirstlane(nt); b = __builtin_amdgcn_readfirstlane(b);
  const ConvGroup& G = P.g[grp];
  conv_bf3_body<MI, EPI, true>(P, G, lds, mt, nt, b);
}

template <int MI, int EPI, int NS = 2>
__global__ void __launch_bounds__(256, 3) conv_bf3_kernel(const ConvParams P) {
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  int mt, grp, nt, b;
  if (!conv_decode_block(P, mt, grp, nt, b)) return;
  mt = __builtin_amdgcn_readfirstlane(mt); grp = __builtin_amdgcn_readfirstlane(grp);  // block-uniform (see conv_mfma_kernel)
  nt = __builtin_amdgcn_readfirstlane(nt); b = __builtin_amdgcn_readfirstlane(b);
  const ConvGroup& G = P.g[grp];
  conv_bf3_body<MI, EPI, false, NS>(P, G, lds, mt, nt, b);
}
