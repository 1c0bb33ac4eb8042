// mfmaclock: shader clock while EVERY SIMD issues back-to-back v_mfma_f32_32x32x2f32 (the sustained-load case of the
// big-tile conv kernel), from s_memtime cycles vs the 100 MHz wall clock, for kernels of ~0.1 .. 5 ms; and the fp32
// MFMA rate that clock allows (256 CUs x 4 SIMDs x 4096 FLOP / 64 cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) burn(float* out, long long* t, int iters) {
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  const float a = (float)(threadIdx.x & 63), b = 1.0f / (float)(1 + (threadIdx.x & 7));
  long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
  }
  long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  float* out; long long* d; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&d, 64);
  int wfreq = 0; hipDeviceGetAttribute(&wfreq, hipDeviceAttributeWallClockRate, 0);
  for (int blocks : {256, 768}) for (int iters : {500, 5000, 50000}) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(burn, dim3(blocks), dim3(256), 0, 0, out, d, iters);
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double us = (double)h[1] / wfreq * 1e3, mhz = (double)h[0] / us;
    const double mf = (double)blocks * 4 * 8 * iters;   // wave-level MFMAs
    printf("%4d workgroups, %6d iters: %8.1f us, cycle counter %.0f MHz, %.1f cycles per MFMA per SIMD, %.1f TFLOP/s\n", blocks, iters, us, mhz,
           h[0] / (8.0 * iters) / (blocks / 256), mf * 4096 / us / 1e6);
  }
  return 0;
}
