// mfmaprobe: what one wave per SIMD can sustain around v_mfma_f32_32x32x2f32 (64 pipe cycles) when the same wave
// also issues the loads / address math / activation of the K-split conv loop.  Cycles per MFMA, 1 wave per SIMD
// (256 blocks x 256 threads), variants cumulative:
//   0 mfma only (one accumulator)             1 + one L1-hit global_load_dword per MFMA (32-bit offset addressing)
//   2 same with 64-bit per-lane addresses      3 + leaky-relu/select VALU (4 ops)
//   4 variant 1 with 4 independent accumulators  5 two MFMAs per load (register-tiled)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ x, float* out, long long* cyc, int iters, float slope, long long stride) {
  f32x16 acc[4];
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = (float)lane;
  const float* xp = x + lane;
  unsigned off = lane;
  float b[8];
  for (int p = 0; p < 8; ++p) b[p] = x[lane + p];
  long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float nb;
      if (V == 0) nb = b[p];
      else if (V == 2 || V == 3) nb = xp[(long long)(p + (i & 3)) * stride];          // 64-bit address math per load
      else nb = x[off + (unsigned)((p + (i & 3)) * 64)];                                 // SGPR base + 32-bit VGPR offset
      float bv = b[p];
      if (V == 3) bv = (lane < 60) ? fmaxf(bv, bv * slope) : 0.f;
      if (V == 4) acc[p & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[p & 3], 0, 0, 0);
      else if (V == 5) { acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + 1.f, bv, acc[1], 0, 0, 0); }
      else acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[0], 0, 0, 0);
      b[p] = nb;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
}
template <int V> void run(const float* x, float* out, long long* d, int blocks) {
  const int iters = 400;
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, x, out, d, iters, 0.1f, 64LL);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, x, out, d, iters, 0.1f, 64LL);
  long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  const int mf = (V == 5 ? 2 : 1) * 8 * iters;
  printf("variant %d, %4d blocks: %.1f cycles per MFMA\n", V, blocks, (double)h / mf);
}
int main() {
  float *x, *out; long long* d;
  hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&d, 64);
  for (int blocks : {256, 512}) {
    run<0>(x, out, d, blocks); run<1>(x, out, d, blocks); run<2>(x, out, d, blocks); run<3>(x, out, d, blocks); run<4>(x, out, d, blocks); run<5>(x, out, d, blocks);
  }
  return 0;
}
