// launchprobe: cost of a dependent kernel boundary (eager stream vs hipGraph replay) for trivial
// kernels of a few grid sizes, and for a kernel with a large dynamic-LDS request.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny_lds(float* p) { extern __shared__ float s[]; s[threadIdx.x] = p[0]; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = s[1] + 1.f; }
static float run(hipStream_t st, float* d, int n, int blocks, size_t lds, bool graph) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipGraphExec_t ge = nullptr;
  if (graph) {
    hipGraph_t g; hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) { if (lds) hipLaunchKernelGGL(tiny_lds, dim3(blocks), dim3(256), lds, st, d); else hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, d); }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); hipGraphDestroy(g);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
  }
  hipEventRecord(e0, st);
  if (graph) hipGraphLaunch(ge, st);
  else for (int i = 0; i < n; ++i) { if (lds) hipLaunchKernelGGL(tiny_lds, dim3(blocks), dim3(256), lds, st, d); else hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st, d); }
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}
int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipFuncSetAttribute((const void*)tiny_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep)
    for (int blocks : {1, 36, 256, 2048}) {
      printf("blocks=%5d  eager %.2f us/kernel   graph %.2f us/kernel   graph+49KB-LDS %.2f   graph+8KB-LDS %.2f\n", blocks,
             run(st, d, 500, blocks, 0, false), run(st, d, 500, blocks, 0, true), run(st, d, 500, blocks, 49152, true), run(st, d, 500, blocks, 8192, true));
    }
  return 0;
}
