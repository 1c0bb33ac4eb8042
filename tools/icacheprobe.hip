// icacheprobe: cost of executing cold straight-line code once per launch.
// big<ID>: N dependent FMAs fully unrolled (8 B each -> N*8 bytes of code), distinct instantiations thrash
// the instruction cache like the ~15 different kernels of one forward do.  small<ID>: the same FMAs in a loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ID, int N>
__global__ void big(float* p) {
  float a = p[threadIdx.x] + ID;
#pragma unroll
  for (int i = 0; i < N; ++i) a = a * 1.0001f + 0.5f;
  p[threadIdx.x] = a;
}
template <int ID, int N>
__global__ void small(float* p) {
  float a = p[threadIdx.x] + ID;
#pragma unroll 1
  for (int i = 0; i < N; ++i) a = a * 1.0001f + 0.5f;
  p[threadIdx.x] = a;
}
template <int N, bool BIG>
static float chain(hipStream_t st, float* d, int reps, int blocks) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int r = 0; r < reps; ++r) {
    if (BIG) {
      hipLaunchKernelGGL((big<0, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((big<1, N>), dim3(blocks), dim3(256), 0, st, d);
      hipLaunchKernelGGL((big<2, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((big<3, N>), dim3(blocks), dim3(256), 0, st, d);
      hipLaunchKernelGGL((big<4, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((big<5, N>), dim3(blocks), dim3(256), 0, st, d);
      hipLaunchKernelGGL((big<6, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((big<7, N>), dim3(blocks), dim3(256), 0, st, d);
    } else {
      hipLaunchKernelGGL((small<0, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((small<1, N>), dim3(blocks), dim3(256), 0, st, d);
      hipLaunchKernelGGL((small<2, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((small<3, N>), dim3(blocks), dim3(256), 0, st, d);
      hipLaunchKernelGGL((small<4, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((small<5, N>), dim3(blocks), dim3(256), 0, st, d);
      hipLaunchKernelGGL((small<6, N>), dim3(blocks), dim3(256), 0, st, d); hipLaunchKernelGGL((small<7, N>), dim3(blocks), dim3(256), 0, st, d);
    }
  }
  hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / (reps * 8);
}
int main() {
  float* d; hipMalloc(&d, 1 << 20); hipMemset(d, 0, 1 << 20);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int blocks : {12, 256}) {
    printf("blocks=%d\n", blocks);
    printf("  N=500  (4 KB code):  unrolled %.2f us/kernel   rolled %.2f us/kernel\n", chain<500, true>(st, d, 20, blocks), chain<500, false>(st, d, 20, blocks));
    printf("  N=2000 (16 KB code): unrolled %.2f us/kernel   rolled %.2f us/kernel\n", chain<2000, true>(st, d, 20, blocks), chain<2000, false>(st, d, 20, blocks));
    printf("  N=5000 (40 KB code): unrolled %.2f us/kernel   rolled %.2f us/kernel\n", chain<5000, true>(st, d, 20, blocks), chain<5000, false>(st, d, 20, blocks));
  }
  return 0;
}
