// LDS access-pattern microbenchmark for the IMPLSCH sweep (gfx950): cycles of the CU's LDS pipe per wave-instruction for the
// accesses the rotated reads of a staged row can be made of -- 8-byte reads at 8-byte and at 4-byte aligned addresses, 4-byte reads
// at a stride of one and of two words, ds_read2_b32, 12-byte reads, 16-byte reads, and the writes.  Four SIMDs of a CU issue (blocks of
// 256 threads), W blocks per CU; the figure printed is ns and cycles per wave-instruction per CU.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds tools/ubench_lds.hip && ./ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define N_INNER 64
#define N_OUTER 1000

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n_outer, long long* clk) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 512];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int i = t; i < 4 * 512; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int base = wv * 2048;   // bytes: 512 words per wave
  float a[8]; f2 p[8]; f3 q[8]; f4 r[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = (float)i; p[i] = f2{a[i], 1.f}; q[i] = f3{a[i], 1.f, 2.f}; r[i] = f4{a[i], 1.f, 2.f, 3.f}; }
  const int a8 = base + lane * 8, a8u = base + lane * 8 + 4, a4 = base + lane * 4, a16 = base + lane * 16;
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int o = 0; o < n_outer; o++) {
#pragma unroll
    for (int i = 0; i < N_INNER; i++) {
      const int s = i & 7;
      if (KIND == 0) asm volatile("ds_read_b64 %0, %1" : "=v"(p[s]) : "v"(a8));
      if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(p[s]) : "v"(a8u));
      if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(a[s]) : "v"(a4));
      if (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(a[s]) : "v"(a8));
      if (KIND == 4) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(p[s]) : "v"(a8u));
      if (KIND == 5) asm volatile("ds_read_b96 %0, %1" : "=v"(q[s]) : "v"(a8));
      if (KIND == 6) asm volatile("ds_read_b96 %0, %1" : "=v"(q[s]) : "v"(a8u));
      if (KIND == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(r[s]) : "v"(a16));
      if (KIND == 8) asm volatile("ds_write_b64 %0, %1" : : "v"(a8), "v"(p[s]));
      if (KIND == 9) asm volatile("ds_write_b64 %0, %1" : : "v"(a8u), "v"(p[s]));
      if (KIND == 10) asm volatile("ds_write_b32 %0, %1" : : "v"(a4), "v"(a[s]));
      if (KIND == 11) asm volatile("ds_read_b128 %0, %1" : "=v"(r[s]) : "v"(a8));      // 16 bytes at an 8-byte stride (overlapping, 8-byte aligned)
      if (KIND == 12) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1" : "=v"(r[s]) : "v"(a8));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + q[i].x + q[i].z + r[i].x + r[i].w;
  out[blockIdx.x * 256 + t] = s + lds[t];
  if (t == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, float* out, long long* clk) {
  for (int w = 1; w <= 2; w++) {
    const int blocks = 256 * w;   // 256 CUs x w blocks of 4 waves = w waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, N_OUTER, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[2];
    hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // s_memrealtime ticks at 100 MHz
    const double inst_per_cu = 4.0 * w * N_OUTER * N_INNER;
    const double ns = ms * 1e6 / inst_per_cu;
    printf("%-34s waves/SIMD %d  %.3f ms  %.2f ns per wave-instruction per CU  clock %.2f GHz -> %.2f cycles of the CU's LDS pipe\n", name, w, ms, ns, ghz, ns * ghz);
  }
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&clk, 16);
  run<0>("ds_read_b64 8-byte aligned", out, clk);
  run<1>("ds_read_b64 4-byte aligned", out, clk);
  run<2>("ds_read_b32 stride 1 word", out, clk);
  run<3>("ds_read_b32 stride 2 words", out, clk);
  run<4>("ds_read2_b32 adjacent, odd word", out, clk);
  run<5>("ds_read_b96 stride 8 B, aligned 8", out, clk);
  run<6>("ds_read_b96 stride 8 B, aligned 4", out, clk);
  run<7>("ds_read_b128 stride 16 B", out, clk);
  run<11>("ds_read_b128 stride 8 B", out, clk);
  run<12>("ds_read2_b64 adjacent stride 8 B", out, clk);
  run<8>("ds_write_b64 8-byte aligned", out, clk);
  run<9>("ds_write_b64 4-byte aligned", out, clk);
  run<10>("ds_write_b32 stride 1 word", out, clk);
  return 0;
}
