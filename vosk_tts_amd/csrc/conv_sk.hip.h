// conv_sk.hip.h — STREAM-K schedule for the 64 x 64 software-pipelined tile (conv_sp.hip.h): a measured prototype (round 6, the round-5
// review's item 1a), opt-in (VITS_SK=1).
//
// What it is for: launches whose tiles are all resident at once, one to three per CU, last as long as the CU with the most of them
// (288 equal tiles on 256 CUs: 39 us against 28 for a lone tile, profiles/r5_bt_64x64.txt).  Here the grid is G persistent workgroups
// (<= 2 per CU: co-resident), and the launch's work -- every (tile, 64-channel stage) of every group, weighted by the group's taps -- is
// cut into G CONTIGUOUS ranges of equal cost.  A range starts in the middle of a tile, covers whole tiles, and ends in the middle of one:
//   * a workgroup whose first segment continues a tile begun by its predecessor(s) PUBLISHES its partial accumulator (64 x 64 floats as
//     {value, epoch} cells, persist.hip.h) -- at the very start of its life;
//   * the workgroup that owns a tile's stage 0 computes its share LAST in its range, then collects the partials of the workgroups that
//     follow it, adds them in workgroup order (fixed order: bit-reproducible) and runs the epilogue.
// A waiter only ever waits for workgroups with HIGHER ids doing the FIRST thing they do, and all G are resident: no deadlock; the polls are
// bounded all the same (SkCtl::timeouts).  The epoch is the launch's: every workgroup reads ctl->epoch + 1 at its start, the last one to
// finish publishes it (cells are never reset).
// Eligibility: what conv_sp_kernel<EPI_STORE> takes (any number of groups, ragged tile maps).
// RESULT (profiles/r6_sk_ab.txt): parity green on the ragged / full-size / mid-size tests, and SLOWER -- s8 3.11 -> 3.65 ms, s16 5.15 -> 5.55,
// c3 19.98 -> 20.55 with G = 512; G = 256: s8 4.12.  The forwards' launches are GROUPED (three kernel sizes, heaviest first): 864 tiles of
// mixed cost on 256 CUs already end within one light tile of the mean, so there was little imbalance to remove -- and a persistent grid
// keeps at most two workgroups on a CU where the plain launch keeps three to four: it is the resident waves that hide a tile's
// staging and issue cost at these sizes, not the schedule.  Kept as an opt-in prototype (VITS_SK / vits_debug_conv_sk), default off.
#pragma once
#include "conv_sp.hip.h"
#include "persist.hip.h"

struct SkCtl { unsigned epoch, done, timeouts, pad; };
struct SkArgs { SkCtl* ctl; ll_t* ws; int spin_limit; };
#define SK_CELLS 4096  // a tile's partial accumulator: 4 waves x 16 elements x 64 lanes

template <int JT>
__global__ void __launch_bounds__(256, 3) conv_sk_kernel(const ConvParams P, const SkArgs A) {
  extern __shared__ float lds[];
  kernarg_warm<sizeof(ConvParams)>();
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x, Gn = gridDim.x;
  unsigned epoch;
  {
    const unsigned e = __hip_atomic_load(&A.ctl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    epoch = __builtin_amdgcn_readfirstlane(e ? e : 1u);
  }
  const int nstages = P.Cin / SP_STAGE_CH;
  const int ncol = P.tile_start ? P.tile_start[P.B] : P.ntiles_n * P.B;  // column tiles with work (all items)
  const long long S = (long long)P.ntiles_m * ncol * nstages;            // stage units of ONE group
  long long Ktot = 0;
  for (int q = 0; q < P.n_groups; ++q) Ktot += P.g[q].K;
  const long long C = S * Ktot;                                            // the launch's cost in (stage, tap) units
  // cost position -> global stage-unit index (groups laid one after the other, heaviest first: launch_conv sorted them)
  auto unit_of = [&](long long x) -> long long {
    long long base = 0;
    for (int q = 0; q < P.n_groups; ++q) {
      const long long span = S * P.g[q].K;
      if (x < base + span || q == P.n_groups - 1) { long long r = (x - base) / P.g[q].K; return q * S + (r < S ? r : S); }
      base += span;
    }
    return 0;
  };
  auto range_of = [&](int w, long long& a, long long& b) { a = unit_of(C * w / Gn); b = w == Gn - 1 ? S * P.n_groups : unit_of(C * (w + 1) / Gn); };
  long long u, u1;
  range_of(g, u, u1);
  PS_G ll_t* ws = (PS_G ll_t*)A.ws;
  const unsigned cell0 = (unsigned)(wave * 16 * 64 + lane) * 8u;  // this thread's element e lives at cell0 + e * 512 bytes of a slot
  while (u < u1) {
    const int grp = (int)(u / S);
    const long long r = u - (long long)grp * S;
    const long long v = r / nstages;
    const int sb = (int)(r - v * nstages);
    int se = nstages;
    if (u + (se - sb) > u1) se = sb + (int)(u1 - u);
    u += se - sb;
    // virtual block v of the group -> (M tile, item, column tile)
    const int mt = (int)(v % P.ntiles_m);
    const int q = (int)(v / P.ntiles_m);
    int b, nt;
    if (P.tile_start) {
      int lo = 0, hi = P.B - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (P.tile_start[mid] <= q) lo = mid; else hi = mid - 1; }
      b = lo; nt = q - P.tile_start[b];
    } else { nt = q % P.ntiles_n; b = q / P.ntiles_n; }
    const int mtu = __builtin_amdgcn_readfirstlane(mt), ntu = __builtin_amdgcn_readfirstlane(nt), bu = __builtin_amdgcn_readfirstlane(b);
    const ConvGroup& G = P.g[grp];
    f32x16 acc[2];
    if (!conv_sp_tile<JT>(P, G, lds, mtu, ntu, bu, sb, se, acc)) continue;  // padding tile: every segment of it sees the same (nobody publishes, nobody waits)
    f32x16 accs[1][1];
#pragma unroll
    for (int e = 0; e < 16; ++e) accs[0][0][e] = acc[0][e] + acc[1][e];
    if (sb > 0) {  // a tile somebody before this workgroup began: hand the partial over (the first -- and only such -- segment of this range)
      PS_G ll_t* slot = ws + (size_t)g * SK_CELLS;
#pragma unroll
      for (int e = 0; e < 16; ++e) ll_store_off(slot, cell0 + (unsigned)e * 512u, accs[0][0][e], epoch);
      continue;
    }
    if (se < nstages) {  // this workgroup began the tile: collect the rest, in workgroup order
      int got = se, w = g + 1;
      while (got < nstages && w < Gn) {
        long long wa, wb;
        range_of(w, wa, wb);
        const int len = (int)((wb - wa) < (long long)(nstages - got) ? (wb - wa) : (long long)(nstages - got));
        if (len > 0) {
          const PS_G ll_t* slot = ws + (size_t)w * SK_CELLS;
          ll_t qv[16];
          int spins = 0;
          bool pending;
          do {
            unsigned bad = 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) qv[e] = ll_load_off(slot, cell0 + (unsigned)e * 512u);
#pragma unroll
            for (int e = 0; e < 16; ++e) bad |= ll_bad(qv[e], epoch);
            pending = PS_PENDING(bad);
            if (pending && ++spins > 8) __builtin_amdgcn_s_sleep(4);
          } while (pending && spins < A.spin_limit);
          if (pending && tid == 0) atomicAdd(&A.ctl->timeouts, 1u);
#pragma unroll
          for (int e = 0; e < 16; ++e) accs[0][0][e] += ll_val(qv[e]);
          got += len;
        }
        ++w;
      }
    }
    conv_sp_epilogue<EPI_STORE>(P, G, mtu, ntu, bu, accs);
  }
  // the last workgroup to finish publishes the epoch
  __syncthreads();
  if (tid == 0) {
    const unsigned old = atomicAdd(&A.ctl->done, 1u);
    if (old == (unsigned)Gn - 1u) {
      __hip_atomic_store(&A.ctl->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&A.ctl->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
