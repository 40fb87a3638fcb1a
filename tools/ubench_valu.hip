// Issue-rate microbenchmark for the IMPLSCH cost model (gfx950): cycles per wave-instruction of v_fma_f32, v_pk_fma_f32,
// v_exp_f32, v_cndmask, ds_bpermute_b32, ds_read_b32/b64 and DPP moves at 1..8 waves per SIMD.  Every CU runs the same
// number of waves; time = max over the chip.  Output: one line per (instruction, waves/SIMD) with ns per instruction per SIMD
// and, from the measured shader clock (s_memtime / s_memrealtime), cycles.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define N_INNER 64
#define N_OUTER 2000

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n_outer, long long* clk) {
  __shared__ float lds[256 * 4];
  const int t = threadIdx.x;
  lds[t] = (float)t;
  lds[t + 256] = 1.f;
  __syncthreads();
  float a[8];
  f2 p[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = 1.0f + 1e-3f * (float)(t + i); p[i] = f2{a[i], a[i] * 0.5f}; }
  const float c = 0.999f, d = 1e-4f;
  const f2 c2 = {c, c}, d2 = {d, d};
  int addr = ((t + 7) & 63) * 4, addr2 = ((t + 3) & 63) * 8;
  int sg[4] = {0, 0, 0, 0};
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int o = 0; o < n_outer; o++) {
#pragma unroll
    for (int i = 0; i < N_INNER; i++) {
      const int s = i & 7;
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[s]) : "v"(c), "v"(d));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[s]) : "v"(c2), "v"(d2));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[s]));
      if (KIND == 3) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %1, %2, vcc" : "+v"(a[s]) : "v"(c), "v"(d) : "vcc");
      if (KIND == 4) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(a[s]) : "v"(addr));
      if (KIND == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(a[s]) : "v"(addr));
      if (KIND == 6) asm volatile("ds_read_b64 %0, %1" : "=v"(p[s]) : "v"(addr2));
      if (KIND == 7) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[s]));
      if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[s]));
      if (KIND == 9) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[s]) : "v"(c));
      if (KIND == 10) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[s]) : "v"(c2));
      if (KIND == 11) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[s]));
      if (KIND == 12) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[s]));
      if (KIND == 13) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sg[s & 3]) : "v"(a[s]));
      if (KIND == 14) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[s]), "+v"(a[(s + 1) & 7]));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  s += (float)(sg[0] + sg[1] + sg[2] + sg[3]);
  out[blockIdx.x * 256 + t] = s;
  if (t == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, float* out, long long* clk) {
  for (int wps = 1; wps <= 8; wps *= 2) {
    for (int extra = 0; extra < (wps == 2 ? 2 : 1); extra++) {
      const int w = extra ? 3 : wps;
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
      const double inst_per_simd = (double)w * N_OUTER * N_INNER;
      const double ns = ms * 1e6 / inst_per_simd;
      printf("%-14s waves/SIMD %d  %.3f ms  %.2f ns/inst/SIMD  clock %.2f GHz  -> %.2f cycles per wave-instruction per SIMD\n", name, w, ms, ns, ghz,
             ns * ghz);
    }
  }
}

int main() {
  float* out;
  long long* clk;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float) * 4);
  hipMalloc(&clk, 16);
  run<0>("v_fma_f32", out, clk);
  run<1>("v_pk_fma_f32", out, clk);
  run<9>("v_mul_f32", out, clk);
  run<10>("v_pk_mul_f32", out, clk);
  run<2>("v_exp_f32", out, clk);
  run<8>("v_rcp_f32", out, clk);
  run<3>("cmp+cndmask", out, clk);
  run<4>("ds_bpermute", out, clk);
  run<5>("ds_read_b32+add", out, clk);
  run<6>("ds_read_b64+pkadd", out, clk);
  run<7>("dpp quad+add", out, clk);
  run<11>("dpp wave_shr", out, clk);
  run<12>("dpp row_shr+add", out, clk);
  run<13>("v_readlane", out, clk);
  run<14>("permlane32_swap", out, clk);
  return 0;
}
