// mfmaprobe: what ONE wave per SIMD sustains around v_mfma_f32_32x32x2f32 (64 pipe cycles) when the same wave also
// issues the loads / address math / activation of the K-split conv loop.  Software-pipelined like the real loop:
// fragments loaded in step-group g are consumed in group g+1 (two register buffers, no copies), all loads hit L1.
// Cycles per MFMA for 1, 2 and 3 workgroups per CU (= waves per SIMD):
//   0 MFMA only                                  1 + one global_load_dword per MFMA (SGPR base + 32-bit lane offset)
//   2 as 1 with 64-bit per-lane addresses          3 as 1 + leaky-relu/select VALU (3 ops) on the operand
//   4 as 3 + one global_load_dwordx4 per 4 MFMAs   5 as 4 with 2 MFMAs per operand load (64x32 register tile)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__device__ __forceinline__ void group(const float* __restrict__ x, const float* xp, unsigned off, int i, long long stride, float slope, int lane,
                                      float a, const float (&cur)[8], float (&nxt)[8], const float4 (&acur)[2], float4 (&anxt)[2], f32x16 (&acc)[2]) {
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    if (V >= 4 && (p & 3) == 0) anxt[p >> 2] = reinterpret_cast<const float4*>(x)[lane + 64 * ((p >> 2) + (i & 3))];
    if (V == 2) nxt[p] = xp[(long long)(p + (i & 3)) * stride];
    else if (V >= 1) nxt[p] = x[off + (unsigned)((p + (i & 3)) * 64)];
    float bv = cur[p];
    if (V >= 3) bv = (lane < 60) ? fmaxf(bv, bv * slope) : 0.f;
    const float av = V >= 4 ? (&acur[p >> 2].x)[p & 3] + a : a;
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0], 0, 0, 0);
    if (V == 5) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av + 1.f, bv, acc[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int V>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ x, float* out, long long* cyc, int iters, float slope, long long stride) {
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  const int lane = threadIdx.x & 63;
  const float a = (float)lane;
  const float* xp = x + lane;
  const unsigned off = lane;
  float b0[8], b1[8];
  float4 a0[2], a1[2];
  for (int p = 0; p < 8; ++p) { b0[p] = x[lane + p]; b1[p] = 0.f; }
  a0[0] = a0[1] = a1[0] = a1[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; i += 2) {
    group<V>(x, xp, off, i, stride, slope, lane, a, b0, b1, a0, a1, acc);
    group<V>(x, xp, off, i + 1, stride, slope, lane, a, b1, b0, a1, a0, acc);
  }
  long long c1 = __builtin_readcyclecounter();
  float s = a0[0].x + a1[1].w;
  for (int k = 0; k < 2; ++k) for (int e = 0; e < 16; ++e) s += acc[k][e];
  for (int p = 0; p < 8; ++p) s += b0[p] + b1[p];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}
template <int V> void run(const float* x, float* out, long long* d, int blocks) {
  const int iters = 400;
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, x, out, d, iters, 0.1f, 64LL);
  hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(256), 0, 0, x, out, d, iters, 0.1f, 64LL);
  static long long h[4096];
  hipMemcpy(h, d, 8 * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  const int mf = (V == 5 ? 2 : 1) * 8 * iters;
  printf("variant %d, %d workgroup(s)/CU: %.1f cycles per MFMA per wave = %.1f per SIMD\n", V, blocks / 256, s / blocks / mf, s / blocks / mf / (blocks / 256));
}
int main() {
  float *x, *out; long long* d;
  hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&d, 8 * 4096);
  for (int blocks : {256, 512, 768}) {
    run<0>(x, out, d, blocks); run<1>(x, out, d, blocks); run<2>(x, out, d, blocks); run<3>(x, out, d, blocks); run<4>(x, out, d, blocks); run<5>(x, out, d, blocks);
  }
  return 0;
}
