// mfmapower: does the sustained fp32 MFMA rate depend on the DATA?  Every SIMD issues back-to-back v_mfma_f32_32x32x2f32 for ~1-10 ms
// (768 four-wave workgroups = 3 waves per SIMD, like the big-tile conv kernel) with (0) one constant operand pair, (1) pseudo-random
// per-lane operands that change every MFMA (what a convolution feeds the matrix cores), (2) all-zero operands.  Reports the cycle
// counter rate (s_memtime vs the 100 MHz wall clock) and the achieved TFLOP/s against the 157.3 nominal (2.4 GHz) peak.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfmapower tools/mfmapower.hip && tools/mfmapower
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(256) burn(float* out, const float* in, long long* t, int iters) {
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  float a[8], b[8];
  for (int k = 0; k < 8; ++k) {
    if (MODE == 1) { a[k] = in[(threadIdx.x * 8 + k) & 4095]; b[k] = in[(threadIdx.x * 8 + k + 2048 + blockIdx.x) & 4095]; }
    else if (MODE == 2) { a[k] = 0.f; b[k] = 0.f; }
    else { a[k] = (float)(threadIdx.x & 63); b[k] = 1.0f / (float)(1 + (threadIdx.x & 7)); }
  }
  long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[k], acc[k & 3], 0, 0, 0);
    if (MODE == 1) {  // rotate the operands so consecutive MFMAs of a chain see different bits (2 VALU ops per 8 MFMAs)
      const float t0 = a[0]; a[0] = a[7] * -1.0009765625f; a[7] = a[3]; a[3] = t0;
      const float t1 = b[0]; b[0] = b[5] * -0.9990234375f; b[5] = b[2]; b[2] = t1;
    }
  }
  long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  float* out; float* in; long long* d; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&d, 64);
  float h[4096]; unsigned s = 12345u;
  for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  int wfreq = 0; hipDeviceGetAttribute(&wfreq, hipDeviceAttributeWallClockRate, 0);
  const char* names[3] = {"constant operands", "random operands ", "zero operands    "};
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 3; ++mode) for (int iters : {5000, 50000}) {
    const int blocks = 768;
    for (int r = 0; r < 3; ++r) {
      if (mode == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, 0, out, in, d, iters);
      if (mode == 1) hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(256), 0, 0, out, in, d, iters);
      if (mode == 2) hipLaunchKernelGGL(burn<2>, dim3(blocks), dim3(256), 0, 0, out, in, d, iters);
    }
    long long t[2]; hipMemcpy(t, d, 16, hipMemcpyDeviceToHost);
    const double us = (double)t[1] / wfreq * 1e3, mhz = (double)t[0] / us;
    const double mf = (double)blocks * 4 * 8 * iters;
    printf("%s %6d iters: %8.1f us, cycle counter %.0f MHz, %.1f counter cycles per MFMA per SIMD, %.1f TFLOP/s = %.3f of 157.3\n", names[mode], iters, us, mhz,
           t[0] / (8.0 * iters) / (blocks / 256), mf * 4096 / us / 1e6, mf * 4096 / us / 1e6 / 157.3);
  }
  return 0;
}
