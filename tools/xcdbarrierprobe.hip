// xcdbarrierprobe: what does one "layer step" cost inside a persistent kernel?  
// A layer of the single-utterance encoder / duration predictor / flow is: every workgroup produces its slice of a small activation
// tensor (192 channels x 50..150 columns = 38..115 KB), all workgroups synchronise, every workgroup reads the whole tensor.
// Today that is one kernel launch per layer (>= 4.2 us).  The probe runs R such rounds inside ONE kernel and reports the time
// per round, with the working set of workgroups (a) restricted to ONE XCD, exchange through that XCD's L2, or (b) spread over the
// whole chip (8 XCDs), exchange through the memory side -- each with the fence / load flavour it needs, and checks the data.
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcdbarrierprobe tools/xcdbarrierprobe.hip && tools/xcdbarrierprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define MY_XCC_ID() (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf)  // HW_REG_XCC_ID[3:0]

struct Ctl {
  unsigned arrive[16];   // workgroups seen per XCD (set-up barrier)
  unsigned total;        // all workgroups arrived
  unsigned rank[16];     // rank dispenser per XCD
  unsigned bar;          // round barrier counter (monotonic)
  unsigned errors;
  unsigned participants;
  unsigned flags[64];    // MODE 4: one arrival word per worker
};

// MODE 0: workers = workgroups of XCD 0, agent-scope release/acquire fences (what __threadfence() gives), plain loads
// MODE 1: workers = workgroups of XCD 0, stores drained (write-through L1) + agent-scope relaxed loads (L1 bypass, L2 hit): no L2 write-back / invalidate
// MODE 2: workers = one workgroup per CU on all XCDs, agent-scope fences (the chip-wide variant)
// MODE 3: MODE 1 with the barrier counter updated by workgroup-scope atomics (hangs: not run)
// MODE 4: MODE 1 with a flag barrier: no atomics, one arrival word per workgroup, polled with sc1 loads (P <= 64).  Same cost as the
//         atomic counter: the price is the ~0.4 us per dependent agent-scope access, not the atomics.  (With sc0 = workgroup-scope
//         loads / stores instead of sc1 the pollers read stale L1 lines forever: hangs.)
template <int MODE>
__global__ void __launch_bounds__(256) probe(Ctl* c, float* buf, int n_floats, int rounds, long long* cycles) {
  const int tid = threadIdx.x;
  __shared__ unsigned s_rank, s_P;
  const unsigned xcc = MY_XCC_ID();
  if (tid == 0) {
    atomicAdd(&c->arrive[xcc], 1u);
    __threadfence();
    atomicAdd(&c->total, 1u);
    while (atomicAdd(&c->total, 0u) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    const bool worker = MODE == 2 ? true : xcc == 0;  // (modes 0, 1, 3: the workgroups that landed on XCD 0)
    unsigned P = 0;
    if (MODE == 2) P = gridDim.x; else P = atomicAdd(&c->arrive[0], 0u);
    s_P = P;
    s_rank = worker ? atomicAdd(&c->rank[MODE == 2 ? 0 : xcc], 1u) : 0xffffffffu;
    if (worker && s_rank == 0) c->participants = P;
  }
  __syncthreads();
  const unsigned rank = s_rank, P = s_P;
  if (rank == 0xffffffffu) return;
  const int per = (n_floats + (int)P - 1) / (int)P;
  const int lo = (int)rank * per, hi = lo + per < n_floats ? lo + per : n_floats;
  long long t0 = 0;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (r == 1) t0 = __builtin_readcyclecounter();  // round 0 warms caches / code
    float* cur = buf + (size_t)(r & 1) * n_floats;  // ping-pong: one barrier per round is enough
    // produce this workgroup's slice
    for (int i = lo + tid; i < hi; i += 256) cur[i] = (float)(r * 7 + (i & 1023));
    // ---- barrier over the P workers
    if (MODE == 1 || MODE == 3 || MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have reached L2 (the L1 is write-through)
    else __threadfence();
    __syncthreads();
    if (MODE == 4) {
      // flag barrier without atomics: the XCD's L2 is the coherence point of its CUs.  Every workgroup publishes the round number in
      // its own word (plain store, written through the L1), then wave 0 polls all P words with L1-bypassing loads (L2 hits).
      if (tid == 0) {
        unsigned val = (unsigned)(r + 1);
        unsigned* fp = c->flags + rank;
        asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" :: "v"(fp), "v"(val) : "memory");
      }
      if (tid < 64) {
        const unsigned* fp = c->flags + (tid < (int)P ? tid : 0);
        unsigned seen;
        do {
          asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(fp) : "memory");
        } while (__builtin_amdgcn_read_exec() != __builtin_amdgcn_ballot_w64(seen >= (unsigned)(r + 1)));
      }
    } else
    if (tid == 0) {
      const unsigned target = (unsigned)(r + 1) * P;
      if (MODE == 3) {  // workgroup-scope atomics execute in the XCD's own L2: coherent among the CUs of ONE XCD, no trip to the memory side
        __hip_atomic_fetch_add(&c->bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_fetch_add(&c->bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {}
      } else {
        atomicAdd(&c->bar, 1u);
        while (atomicAdd(&c->bar, 0u) < target) {}
      }
    }
    __syncthreads();
    if (MODE != 1 && MODE != 3 && MODE != 4) __threadfence();
    // consume the whole tensor (MODE 1: sc1 loads = L1 bypass, L2 hit; 8 x dwordx4 in flight per thread)
    if (MODE == 1 || MODE == 3 || MODE == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const int nv = n_floats >> 2;
      for (int i0 = tid; i0 < nv; i0 += 256 * 8) {
        f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = i0 + 256 * k < nv ? i0 + 256 * k : nv - 1;
          const float* p = cur + 4 * (size_t)i;
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[k]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = i0 + 256 * k < nv ? i0 + 256 * k : nv - 1;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (v[k][q] != (float)(r * 7 + ((4 * i + q) & 1023))) atomicAdd(&c->errors, 1u);
            acc += v[k][q];
          }
        }
      }
    } else {
      for (int i = tid; i < n_floats; i += 256) {
        const float v = cur[i];
        if (v != (float)(r * 7 + (i & 1023))) atomicAdd(&c->errors, 1u);
        acc += v;
      }
    }
  }
  if (rank == 0 && tid == 0) cycles[0] = __builtin_readcyclecounter() - t0;
  if (acc == 12345.678f) buf[0] = acc;
}

template <int MODE>
static void run(const char* what, int n_floats, int rounds) {
  Ctl* c; float* buf; long long* cyc;
  hipMalloc((void**)&c, sizeof(Ctl)); hipMemset(c, 0, sizeof(Ctl));
  hipMalloc((void**)&buf, sizeof(float) * n_floats * 2);
  hipMalloc((void**)&cyc, 8); hipMemset(cyc, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, c, buf, n_floats, rounds, cyc);
  hipEventRecord(e1, 0);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  Ctl h; long long hc = 0;
  hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %s  workers %3u  %6.2f us/round (event, incl. set-up)  %7.0f cycles/round  data errors %u  [per XCD:", what,
         e == hipSuccess ? "ok " : hipGetErrorString(e), h.participants, ms * 1e3 / rounds, (double)hc / (rounds - 1), h.errors);
  for (int i = 0; i < 8; ++i) printf(" %u", h.arrive[i]);
  printf("]\n");
  hipFree(c); hipFree(buf); hipFree(cyc);
}

int main() {
  const int rounds = 200;
  for (int n : {256, 192 * 50, 192 * 150, 768 * 150}) {
    printf("tensor of %d floats (%d KB), %d rounds of produce-slice / barrier / read-all:\n", n, n * 4 / 1024, rounds);
    run<2>("  whole chip, agent-scope fences", n, rounds);
    run<0>("  one XCD, agent-scope fences", n, rounds);
    run<1>("  one XCD, stores drained + L1-bypassing loads (no L2 maintenance)", n, rounds);
    run<4>("  one XCD, as above + flag barrier (plain stores, L1-bypassing polls)", n, rounds);
    // run<3>: the same with the barrier counter on workgroup-scope atomics (hoping for XCD-local L2 atomics) never sees the other
    // CUs' arrivals -- it hangs; not run
  }
  return 0;
}
