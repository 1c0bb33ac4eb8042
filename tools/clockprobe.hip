// clockprobe: shader clock (s_memtime cycles vs 100 MHz wall clock) and dependent global-load
// latency, for short kernels launched back to back (the B=1 regime) and for one long kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void spin(long long* out, int iters) {
  long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}
__global__ void chase(const int* p, int n, long long* out) {
  long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  int j = 0;
  for (int i = 0; i < n; ++i) j = p[j];
  long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[0] = c1 - c0; out[1] = w1 - w0; out[2] = j;
}
int main() {
  long long* d; hipMalloc(&d, 64);
  long long h[3];
  int wfreq = 0; hipDeviceGetAttribute(&wfreq, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate kHz: %d\n", wfreq);
  for (int rep = 0; rep < 3; ++rep) {
    for (int iters : {2000, 200000, 20000000}) {
      hipLaunchKernelGGL(spin, dim3(256 * 4), dim3(256), 0, 0, d, iters);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("spin iters=%9d cycles=%12lld wall_ticks=%10lld -> %.0f MHz (%.1f us)\n", iters, h[0], h[1], (double)h[0] / h[1] * wfreq / 1e3, h[1] * 1e3 / wfreq);
    }
  }
  // pointer chase over 256 MB (beyond MALL) with a random permutation: true miss latency
  for (size_t mb : {1, 64, 1024}) {
    size_t n = mb * 1024 * 1024 / 4 / 16;  // one int per 64 B line
    std::vector<int> perm(n * 16, 0);
    std::vector<int> idx(n); for (size_t i = 0; i < n; ++i) idx[i] = i;
    unsigned s = 12345; for (size_t i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; size_t j = s % (i + 1); std::swap(idx[i], idx[j]); }
    for (size_t i = 0; i < n; ++i) perm[idx[i] * 16] = idx[(i + 1) % n] * 16;
    int* dp; hipMalloc(&dp, perm.size() * 4); hipMemcpy(dp, perm.data(), perm.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, dp, 2000, d);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("chase %5zu MB: %.0f cycles/load, %.0f ns/load\n", mb, (double)h[0] / 2000, h[1] * 1e6 / wfreq / 2000);
    }
    hipFree(dp);
  }
  return 0;
}
