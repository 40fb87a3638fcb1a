// Calibration of FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ on gfx950 (MI355X_MICROARCH.md, HBM): a plain 16 B/lane streaming copy of the
// O320 spectral array (421 080 x 36 x 36 floats = 2.18 GB), grid-stride, launched three times.  Known bytes per launch: 2 182 878 720
// read + as many written.   hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_copy tools/calib_copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_calib_copy16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// the same bytes with 4 B/lane accesses
__global__ void k_calib_copy4(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t nfl = (size_t)421080 * 36 * 36;
  float *a, *b;
  if (hipMalloc(&a, nfl * 4) != hipSuccess || hipMalloc(&b, nfl * 4) != hipSuccess) return 1;
  (void)hipMemset(a, 1, nfl * 4);
  (void)hipMemset(b, 0, nfl * 4);
  for (int it = 0; it < 3; it++) {
    hipLaunchKernelGGL(k_calib_copy16, dim3(256 * 32), dim3(256), 0, 0, (const float4*)a, (float4*)b, nfl / 4);
    hipLaunchKernelGGL(k_calib_copy4, dim3(256 * 32), dim3(256), 0, 0, (const float*)a, b, nfl);
  }
  (void)hipDeviceSynchronize();
  printf("bytes per launch: read %zu, written %zu\n", nfl * 4, nfl * 4);
  return 0;
}
