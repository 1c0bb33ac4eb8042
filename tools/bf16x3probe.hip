// bf16x3probe: the "split-bf16" lever named in DESIGN.md section 6, measured before anyone builds a conv kernel on it.
// One wave computes a 32 x 32 tile C = A[32 x K] * B[K x 32] four ways and the host checks every result against fp64:
//   0  v_mfma_f32_32x32x2f32                                   (what the conv kernels use today)
//   1  one v_mfma_f32_32x32x16_bf16 per 16 k                   (plain bf16: rate ceiling, error floor)
//   2  3 bf16 MFMAs: hi*hi + hi*lo + lo*hi       (2-piece split, ~2^-16), operands split ahead of time
//   3  6 bf16 MFMAs: 3-piece split up to 2^-16 cross terms (~2^-24), operands split ahead of time
//   4  as 3, but B arrives as fp32 and is split in the loop (v_cvt_pk_bf16_f32 + subtract): the activation side of a conv
// Reports cycles per 16 k-steps of one tile per wave (fp32: 8 MFMAs x 64 = 512 pipe cycles) with 1 and 2 waves per SIMD,
// and max |error| / max |C| against the fp64 reference.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_round(float x) {  // fp32 value of RNE(x -> bf16)
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ void split3(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float h = bf16_round(x[e]), r1 = x[e] - h, m = bf16_round(r1), l = bf16_round(r1 - m);
    hi[e] = (__bf16)h; mid[e] = (__bf16)m; lo[e] = (__bf16)l;
  }
}
// A fp32 [32][K] row-major, B fp32 [K][32]; fragments: 32x32x2: lane -> (row l&31, k = l>>5); 32x32x16: lane -> (row l&31, k = 8*(l>>5) .. +8)
template <int V>
__global__ void __launch_bounds__(256) probe(const float* __restrict__ A, const float* __restrict__ B, float* C, long long* cyc, int K, int reps) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  long long c0 = 0, c1 = 0;
  for (int rep = 0; rep < reps; ++rep) {
    if (rep == reps - 1) { for (int e = 0; e < 16; ++e) acc[e] = 0.f; c0 = __builtin_readcyclecounter(); }
#pragma unroll 2
    for (int k0 = 0; k0 < K; k0 += 16) {
      if (V == 0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k0 + 2 * p + h], B[(k0 + 2 * p + h) * 32 + l31], acc, 0, 0, 0);
      } else {
        float a[8], b[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] = A[l31 * K + k0 + 8 * h + e]; b[e] = B[(k0 + 8 * h + e) * 32 + l31]; }
        bf16x8 ah, am, al, bh, bm, bl;
        split3(a, ah, am, al);
        split3(b, bh, bm, bl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        if (V >= 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
        }
        if (V >= 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        }
      }
    }
    if (rep == reps - 1) c1 = __builtin_readcyclecounter();
  }
  if (blockIdx.x == 0 && threadIdx.x < 64)
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + l31] = acc[e];
  if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

// rate only: operands already split and resident in registers (variants 1..3), or B split in the loop from fp32 registers (4)
template <int V>
__global__ void __launch_bounds__(256) rate(const float* __restrict__ X, float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[2];
  for (int k = 0; k < 2; ++k) for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
  float x[8];
  for (int e = 0; e < 8; ++e) x[e] = X[lane * 8 + e];
  bf16x8 ah, am, al, bh, bm, bl;
  split3(x, ah, am, al);
  split3(x, bh, bm, bl);
  const float fa = x[0], fb = x[1];
  const long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    if (V == 0) {
#pragma unroll
      for (int p = 0; p < 8; ++p) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
    } else {
      if (V == 4) {  // fresh fp32 B fragment every step (kept live through acc), split here
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += acc[1][e];
        split3(x, bh, bm, bl);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[0], 0, 0, 0);
      if (V >= 2) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[0], 0, 0, 0);
      }
      if (V >= 3) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[0], 0, 0, 0);
      }
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += acc[0][e] + acc[1][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int V> static void run_rate(const float* X, float* out, long long* d, int wgs_per_cu) {
  const int blocks = 256 * wgs_per_cu, iters = 2000;
  hipLaunchKernelGGL(rate<V>, dim3(blocks), dim3(256), 0, 0, X, out, d, iters);
  hipLaunchKernelGGL(rate<V>, dim3(blocks), dim3(256), 0, 0, X, out, d, iters);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (long long v : h) s += v;
  const double per = s / blocks / iters;  // cycles per 16-k step per wave
  printf("  rate  variant %d, %d wave(s)/SIMD: %7.1f cycles per 16-k tile step per wave (%.1f per SIMD; fp32 MFMA needs 512)\n", V, wgs_per_cu, per, per / wgs_per_cu);
}
template <int V> static void run_err(const float* dA, const float* dB, float* dC, long long* d, const std::vector<double>& ref, int K) {
  hipLaunchKernelGGL(probe<V>, dim3(1), dim3(64), 0, 0, dA, dB, dC, d, K, 1);
  std::vector<float> c(1024);
  hipMemcpy(c.data(), dC, 4096, hipMemcpyDeviceToHost);
  double err = 0, mx = 0;
  for (int i = 0; i < 1024; ++i) { err = fmax(err, fabs(c[i] - ref[i])); mx = fmax(mx, fabs(ref[i])); }
  printf("  error variant %d: max |C - C_fp64| / max |C| = %.3e   (K = %d)\n", V, err / mx, K);
}
int main() {
  const int K = 2304;  // enc.ffn2-sized contraction
  std::vector<float> A(32 * K), B(K * 32);
  srand(7);
  for (float& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
  for (float& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 4.0f;
  std::vector<double> ref(1024, 0.0);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC, *out; long long* d;
  hipMalloc((void**)&dA, A.size() * 4); hipMalloc((void**)&dB, B.size() * 4); hipMalloc((void**)&dC, 4096);
  hipMalloc((void**)&out, 4 * 256 * 1024); hipMalloc((void**)&d, 8 * 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  run_err<0>(dA, dB, dC, d, ref, K); run_err<1>(dA, dB, dC, d, ref, K); run_err<2>(dA, dB, dC, d, ref, K); run_err<3>(dA, dB, dC, d, ref, K);
  for (int w = 1; w <= 2; ++w) { run_rate<0>(dA, out, d, w); run_rate<1>(dA, out, d, w); run_rate<2>(dA, out, d, w); run_rate<3>(dA, out, d, w); run_rate<4>(dA, out, d, w); }
  return 0;
}
