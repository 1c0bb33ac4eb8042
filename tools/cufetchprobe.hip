// cufetchprobe: how many bytes per cycle can ONE compute unit pull with global_load_dwordx4, as a function of where the data
// lives (fresh from HBM/MALL vs resident in the XCD's L2 vs L1-sized), the number of waves issuing (4 / 8 / 16 per workgroup,
// one workgroup per CU) and the number of CUs that stream at the same time (32 / 128 / 256 workgroups).  This number bounds
// every few-workgroup kernel of the single-utterance regime (weight slab per workgroup / this rate).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// each workgroup streams `bytes_per_wg` starting at base + (shared ? 0 : blockIdx.x * bytes_per_wg), `reps` times
template <int UNROLL>
__global__ void stream_kernel(const f32x4* __restrict__ base, size_t vec_per_wg, int shared, int reps, float* sink, long long* cyc) {
  const f32x4* p = base + (shared ? 0 : (size_t)blockIdx.x * vec_per_wg);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i + (size_t)(UNROLL - 1) * blockDim.x < vec_per_wg; i += (size_t)UNROLL * blockDim.x) {
      f32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) v[u] = p[i + (size_t)u * blockDim.x];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
  const long long t1 = __builtin_readcyclecounter();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const size_t total = (size_t)1 << 30;
  f32x4* d; hipMalloc(&d, total); hipMemset(d, 0, total);
  float* sink; hipMalloc(&sink, 4);
  long long* cyc; hipMalloc(&cyc, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-10s %6s %6s %9s | %10s %12s %12s\n", "data", "wgs", "waves", "KB/wg", "us", "GB/s per CU", "B/clk per CU");
  for (int shared = 0; shared < 2; ++shared)
    for (size_t kb : {64, 360, 2048})
      for (int wgs : {32, 128, 256})
        for (int waves : {4, 16}) {
          if (!shared && kb * 1024 * wgs > total) continue;
          const size_t vec = kb * 1024 / 16;
          const int reps = shared ? 8 : 1;
          float best = 1e9f; double bclk = 0;
          for (int it = 0; it < 3; ++it) {
            if (!shared) hipMemset(d, 0, 64 << 20);  // evict
            hipEventRecord(e0);
            hipLaunchKernelGGL((stream_kernel<8>), dim3(wgs), dim3(waves * 64), 0, 0, d, vec, shared, reps, sink, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(wgs); hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
            long long mx = 0; for (auto c : h) mx = c > mx ? c : mx;
            if (ms < best) { best = ms; bclk = (double)kb * 1024 * reps / (double)mx; }
          }
          printf("%-10s %6d %6d %9zu | %10.2f %12.1f %12.2f\n", shared ? "L2-shared" : "distinct", wgs, waves, kb, best * 1e3,
                 (double)kb * 1024 * reps / (best * 1e-3) / 1e9, bclk);
        }
  return 0;
}
