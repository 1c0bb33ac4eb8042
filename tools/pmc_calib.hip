// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the conv kernels use
// (MI355X_MICROARCH.md "HBM": only the 16 B/lane streaming read is calibrated there, "calibrate on a known byte
// count in your own access pattern").  Streams 1 GiB (> the 256 MiB Infinity Cache) per kernel:
//   calib_read_dw / calib_read_x4   : N floats read, 4 B or 16 B per lane, one float per block written
//   calib_write_dw / calib_write_x4 : N floats written, nothing read
// build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read_dw(const float* __restrict__ x, float* __restrict__ out, size_t n) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
  if (s == 123.456f) out[blockIdx.x] = s;
}
__global__ void calib_read_x4(const float4* __restrict__ x, float* __restrict__ out, size_t n4) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[blockIdx.x] = s;
}
__global__ void calib_write_dw(float* __restrict__ y, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = 1.f;
}
__global__ void calib_write_x4(float4* __restrict__ y, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) y[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main() {
  const size_t n = (size_t)1 << 28;  // 2^28 floats = 1 GiB
  float *x, *y, *o;
  if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&y, n * 4) != hipSuccess || hipMalloc(&o, 1 << 16) != hipSuccess) return 1;
  hipMemset(x, 0, n * 4);
  hipMemset(y, 0, n * 4);
  hipDeviceSynchronize();
  for (int r = 0; r < 2; ++r) {
    calib_read_dw<<<4096, 256>>>(x, o, n);
    calib_read_x4<<<4096, 256>>>((const float4*)x, o, n / 4);
    calib_write_dw<<<4096, 256>>>(y, n);
    calib_write_x4<<<4096, 256>>>((float4*)y, n / 4);
  }
  hipDeviceSynchronize();
  printf("calibration bytes per launch: %zu\n", n * 4);
  return 0;
}
