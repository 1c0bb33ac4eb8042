// llprobe: what does a dependent "layer step" cost inside a persistent kernel when the exchanged tensor carries its own arrival
// flags?  (round 3; follows tools/xcdbarrierprobe.hip, which priced a step with a separate barrier at 3.9 / 6.7 us.)
//
// Protocol under test ("LL", as in the low-latency protocol of collective libraries): every exchanged element is an 8-byte cell
// {float value, u32 epoch} written with ONE 8-byte store; a consumer polls the cells it needs with L1-bypassing (sc1) 8-byte loads
// until every epoch matches the round.  No drain of the stores, no flag, no barrier: producer store -> L2 -> consumer load.
// Workers are the workgroups that landed on ONE XCD (the L2 of that XCD is the coherence point of its CUs); a chip-wide variant
// checks whether sc1 stores + sc1 loads are also coherent ACROSS XCDs (data errors are counted, every spin loop is bounded).
//   hipcc --offload-arch=gfx950 -O3 -o tools/llprobe tools/llprobe.hip && tools/llprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define MY_XCC_ID() (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf)  // HW_REG_XCC_ID[3:0]
#define NT 512
#define SPIN_LIMIT (1 << 16)  // polls before a loop gives up (sets Ctl::timeouts): the probe can never hang the box

typedef unsigned long long u64;

struct Ctl {
  unsigned arrive[16];
  unsigned total;
  unsigned rank[16];
  unsigned errors;
  unsigned timeouts;
  unsigned participants;
  unsigned retries;       // extra poll rounds (sum over waves of rank 0)
  unsigned flags[64];
};

__device__ __forceinline__ u64 ll_pack(float v, unsigned e) { return ((u64)e << 32) | (u64)__float_as_uint(v); }
__device__ __forceinline__ u64 ll_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int SC1>
__device__ __forceinline__ void ll_st(u64* p, u64 v) {
  if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

__device__ __forceinline__ float val_of(int r, int i) { return (float)(r * 7 + (i & 1023)); }

// set-up: which workgroups work, and their rank.  CHIP = 0: the workgroups of XCD 0; CHIP = 1: one per CU everywhere.
template <int CHIP>
__device__ __forceinline__ bool setup(Ctl* c, unsigned& rank, unsigned& P) {
  __shared__ unsigned s_rank, s_P;
  const unsigned xcc = MY_XCC_ID();
  if (threadIdx.x == 0) {
    atomicAdd(&c->arrive[xcc], 1u);
    __threadfence();
    atomicAdd(&c->total, 1u);
    int spins = 0;
    while (atomicAdd(&c->total, 0u) < gridDim.x && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
    if (spins >= SPIN_LIMIT) atomicAdd(&c->timeouts, 1u);
    const bool worker = CHIP ? true : xcc == 0;
    s_P = CHIP ? gridDim.x : atomicAdd(&c->arrive[0], 0u);
    s_rank = worker ? atomicAdd(&c->rank[0], 1u) : 0xffffffffu;
    if (worker && s_rank == 0) c->participants = s_P;
  }
  __syncthreads();
  rank = s_rank; P = s_P;
  return rank != 0xffffffffu;
}

// ---- MODE 0: LL cells, every worker reads the WHOLE tensor (the pattern of xcdbarrierprobe, for a like-for-like figure)
// ---- MODE 1: LL cells, every worker reads a WINDOW: all C rows x W columns (what a conv tile's prologue needs)
// ---- MODE 2: plain floats + drained stores + flag words + sc1 loads of the same window (xcdbarrierprobe MODE 4 with the real pattern)
// tensor: [C][T] (row-major, T contiguous); worker r produces rows [r*C/P, (r+1)*C/P).
template <int MODE, int CHIP, int SC1ST>
__global__ void __launch_bounds__(NT) probe(Ctl* c, u64* buf, float* fbuf, int C, int T, int W, int rounds, long long* cycles) {
  unsigned rank, P;
  if (!setup<CHIP>(c, rank, P)) return;
  const int tid = threadIdx.x;
  const int n = C * T;
  const int rows_per = (C + (int)P - 1) / (int)P;
  const int r_lo = (int)rank * rows_per, r_hi = r_lo + rows_per < C ? r_lo + rows_per : C;
  const int ntile = (T + 15) / 16;
  const int w0 = ((int)rank % ntile) * 16 - (W - 16) / 2;  // window start column (clamped per element)
  long long t0 = 0;
  float acc = 0.f;
  unsigned my_retries = 0;
  __shared__ int s_gave_up;
  if (tid == 0) s_gave_up = 0;
  __syncthreads();
  for (int r = 0; r < rounds; ++r) {
    if (r == 1) t0 = __builtin_readcyclecounter();
    const unsigned epoch = (unsigned)(r + 1);
    u64* cur = buf + (size_t)(r % 3) * n;
    float* fcur = fbuf + (size_t)(r % 3) * n;
    // produce this worker's rows
    const int cnt = (r_hi - r_lo) * T;
    for (int i = tid; i < cnt; i += NT) {
      const int idx = r_lo * T + i;
      if (MODE == 2) fcur[idx] = val_of(r, idx);
      else ll_st<SC1ST>(cur + idx, ll_pack(val_of(r, idx), epoch));
    }
    if (MODE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        unsigned* fp = c->flags + rank;
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(fp), "v"(epoch) : "memory");
      }
      if (tid < 64) {
        const unsigned* fp = c->flags + (tid < (int)P ? tid : 0);
        unsigned seen;
        int spins = 0;
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(fp) : "memory");
        } while (__builtin_amdgcn_read_exec() != __builtin_amdgcn_ballot_w64(seen >= epoch) && ++spins < SPIN_LIMIT);
        if (spins >= SPIN_LIMIT && tid == 0) atomicAdd(&c->timeouts, 1u);
      }
      __syncthreads();
    }
    // consume
    const int total = MODE == 0 ? n : C * W;
    for (int i0 = tid; i0 < total; i0 += NT * 8) {
      int idx[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int i = i0 + NT * k;
        if (i >= total) i = total - 1;
        if (MODE == 0) idx[k] = i;
        else {
          const int row = i / W, col = w0 + (i - row * W);
          idx[k] = row * T + (col < 0 ? 0 : (col >= T ? T - 1 : col));
        }
      }
      float v[8];
      if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __hip_atomic_load(fcur + idx[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        int spins = 0;
        bool ok;
        do {
          u64 q[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) q[k] = ll_ld(cur + idx[k]);
          ok = true;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            ok = ok && (unsigned)(q[k] >> 32) == epoch;
            v[k] = __uint_as_float((unsigned)q[k]);
          }
          ok = __builtin_amdgcn_ballot_w64(ok) == __builtin_amdgcn_read_exec();
          if (!ok) ++my_retries;
        } while (!ok && ++spins < SPIN_LIMIT);
        if (spins >= SPIN_LIMIT && (tid & 63) == 0) { atomicAdd(&c->timeouts, 1u); s_gave_up = 1; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (v[k] != val_of(r, idx[k])) atomicAdd(&c->errors, 1u);
        acc += v[k];
      }
    }
    __syncthreads();  // (a real step has at least one workgroup barrier: LDS staging)
    if (s_gave_up) break;  // incoherent variant: one timeout per workgroup, then out
  }
  if (rank == 0 && tid == 0) cycles[0] = __builtin_readcyclecounter() - t0;
  if (rank == 0 && (tid & 63) == 0) atomicAdd(&c->retries, my_retries);
  if (acc == 12345.678f) fbuf[0] = acc;
}

// ---- one-way latency: two workers bounce ONE cell.  CHIP = 0: both on XCD 0; CHIP = 1: rank 0 on XCD 0, the partner on XCD 1.
template <int CHIP, int SC1ST>
__global__ void __launch_bounds__(64) pingpong(Ctl* c, u64* cells, int rounds, long long* cycles) {
  const unsigned xcc = MY_XCC_ID();
  __shared__ unsigned s_role;
  if (threadIdx.x == 0) {
    unsigned role = 0xffffffffu;
    if (xcc == 0) { if (atomicAdd(&c->rank[0], 1u) == 0) role = 0; else if (!CHIP && atomicAdd(&c->rank[1], 1u) == 0) role = 1; }
    else if (CHIP && xcc == 1) { if (atomicAdd(&c->rank[1], 1u) == 0) role = 1; }
    s_role = role;
  }
  __syncthreads();
  const unsigned role = s_role;
  if (role == 0xffffffffu || threadIdx.x != 0) return;
  u64* mine = cells + role * 32;         // separate 256-byte lines
  u64* theirs = cells + (1 - role) * 32;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 1; r <= rounds; ++r) {
    if (role == 0) ll_st<SC1ST>(mine, ll_pack(1.f, (unsigned)r));
    int spins = 0;
    while ((unsigned)(ll_ld(theirs) >> 32) != (unsigned)r && ++spins < SPIN_LIMIT) {}
    if (spins >= SPIN_LIMIT) { atomicAdd(&c->timeouts, 1u); break; }
    if (role == 1) ll_st<SC1ST>(mine, ll_pack(1.f, (unsigned)r));
  }
  if (role == 0) cycles[0] = __builtin_readcyclecounter() - t0;
}

template <int MODE, int CHIP, int SC1ST>
static void run(const char* what, int C, int T, int W, int rounds) {
  Ctl* c; u64* buf; float* fbuf; long long* cyc;
  const size_t n = (size_t)C * T;
  hipMalloc((void**)&c, sizeof(Ctl)); hipMemset(c, 0, sizeof(Ctl));
  hipMalloc((void**)&buf, 8 * n * 3); hipMemset(buf, 0, 8 * n * 3);
  hipMalloc((void**)&fbuf, 4 * n * 3); hipMemset(fbuf, 0, 4 * n * 3);
  hipMalloc((void**)&cyc, 8); hipMemset(cyc, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<MODE, CHIP, SC1ST>), dim3(256), dim3(NT), 0, 0, c, buf, fbuf, C, T, W, rounds, cyc);
  hipEventRecord(e1, 0);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  Ctl h; long long hc = 0;
  hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("  %-66s %s workers %3u  %6.2f us/round (event)  %6.0f cyc/round  errors %u timeouts %u retries/round %.1f\n", what,
         e == hipSuccess ? "ok " : hipGetErrorString(e), h.participants, ms * 1e3 / rounds, (double)hc / (rounds - 1), h.errors, h.timeouts,
         (double)h.retries / rounds);
  fflush(stdout);
  hipFree(c); hipFree(buf); hipFree(fbuf); hipFree(cyc);
}

template <int CHIP, int SC1ST>
static void run_pp(const char* what, int rounds) {
  Ctl* c; u64* cells; long long* cyc;
  hipMalloc((void**)&c, sizeof(Ctl)); hipMemset(c, 0, sizeof(Ctl));
  hipMalloc((void**)&cells, 4096); hipMemset(cells, 0, 4096);
  hipMalloc((void**)&cyc, 8); hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((pingpong<CHIP, SC1ST>), dim3(256), dim3(64), 0, 0, c, cells, rounds, cyc);
  hipError_t e = hipDeviceSynchronize();
  Ctl h; long long hc = 0;
  hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("  %-66s %s one-way %6.0f cycles  timeouts %u\n", what, e == hipSuccess ? "ok " : hipGetErrorString(e), (double)hc / rounds / 2, h.timeouts);
  fflush(stdout);
  hipFree(c); hipFree(cells); hipFree(cyc);
}

int main() {
  const int rounds = 200;
  printf("ping-pong of one 8-byte cell (cycles at the shader clock, ~2.4 GHz):\n");
  run_pp<0, 0>("same XCD, plain store + sc1 load", 2000);
  run_pp<0, 1>("same XCD, sc1 store + sc1 load", 2000);
  run_pp<1, 1>("XCD 0 <-> XCD 1, sc1 store + sc1 load", 2000);
  run_pp<1, 0>("XCD 0 <-> XCD 1, plain store + sc1 load (expected to time out)", 50);
  const int shapes[][2] = {{192, 50}, {256, 64}, {192, 160}, {768, 160}};
  for (auto& s : shapes) {
    const int C = s[0], T = s[1];
    printf("tensor [%d x %d] (%d KB as floats), %d rounds:\n", C, T, C * T * 4 / 1024, rounds);
    run<0, 0, 0>("one XCD, LL cells (plain 8 B stores), read ALL", C, T, 16, rounds);
    run<0, 0, 1>("one XCD, LL cells (sc1 8 B stores), read ALL", C, T, 16, rounds);
    run<1, 0, 0>("one XCD, LL cells (plain stores), read a 34-column window", C, T, 34, rounds);
    run<1, 0, 1>("one XCD, LL cells (sc1 stores), read a 34-column window", C, T, 34, rounds);
    run<2, 0, 0>("one XCD, floats + drained stores + flag words, same window", C, T, 34, rounds);
    run<1, 1, 1>("WHOLE CHIP, LL cells (sc1 stores), 34-column window", C, T, 34, rounds);
    run<0, 1, 1>("WHOLE CHIP, LL cells (sc1 stores), read ALL", C, T, 16, rounds);
  }
  return 0;
}
