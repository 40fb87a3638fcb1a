// LDS bank-conflict patterns (gfx950): ds_read_b32 / ds_read_b64 / ds_write_b64 with a per-lane address table given by the host, two waves per
// SIMD; prints the cycles of the CU's LDS pipe per wave-instruction.  Used to find the staging-row layout of the IMPLSCH sweep whose rotated
// reads do not collide.   hipcc --offload-arch=gfx950 -O3 -w -o ubench_lds_pat tools/ubench_lds_pat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define N_INNER 64
#define N_OUTER 1000
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n_outer, long long* clk, const int* tab) {
  __shared__ __attribute__((aligned(16))) float lds[4 * 1024];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  for (int i = t; i < 4 * 1024; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int addr = wv * 4096 + tab[lane];
  float a[8]; f2 p[8]; f4 r4[8];
  for (int i = 0; i < 8; i++) { a[i] = (float)i; p[i] = f2{a[i], 1.f}; r4[i] = f4{a[i], 1.f, 2.f, 3.f}; }
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int o = 0; o < n_outer; o++) {
#pragma unroll
    for (int i = 0; i < N_INNER; i++) {
      const int s = i & 7;
      if (KIND == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(a[s]) : "v"(addr));
      if (KIND == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(p[s]) : "v"(addr));
      if (KIND == 2) asm volatile("ds_write_b64 %0, %1" : : "v"(addr), "v"(p[s]));
      if (KIND == 3) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(a[s]));
      if (KIND == 4) asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:54" : "=v"(r4[s]) : "v"(addr));     // two rows (row stride 108 words)
      if (KIND == 5) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:108" : "=v"(p[s]) : "v"(addr));
      if (KIND == 6) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:54" : : "v"(addr), "v"(p[s]), "v"(p[(s + 1) & 7]));
      if (KIND == 7) asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(r4[s]));
      if (KIND == 8) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(p[s]) : "v"(addr));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + r4[i].x + r4[i].w;
  out[blockIdx.x * 256 + t] = s + lds[t];
  if (t == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
static float* out; static long long* clk; static int* dtab;
template <int KIND>
double run(const std::vector<int>& tab) {
  hipMemcpy(dtab, tab.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
  const int blocks = 512;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, clk, dtab);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, N_OUTER, clk, dtab);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);
  return ms * 1e6 / (8.0 * N_OUTER * N_INNER) * ghz;
}
// the kernel's lane map at 36 directions: lanes 0..47 pairs 0..15 of points 0..2, lanes 48..53 pairs 16, 17, lanes 54..63 shadow lane 32;
// word of element e of point p in a row: base[p] + e
static std::vector<int> v4tab(const int base[3], int shift, int half, int elem_bytes = 4) {
  std::vector<int> t(64);
  for (int l = 0; l < 64; l++) {
    int p, j;
    if (l < 48) { p = l >> 4; j = l & 15; } else if (l < 54) { p = (l - 48) >> 1; j = 16 + ((l - 48) & 1); } else { p = 2; j = 0; }
    int e = ((2 * j + shift) % 36 + 36) % 36 + half;
    if (e >= 36) e -= 36;
    t[l] = (base[p] + e) * elem_bytes;
  }
  return t;
}
int main() {
  hipMalloc(&out, 512 * 256 * sizeof(float)); hipMalloc(&clk, 16); hipMalloc(&dtab, 256);
  std::vector<int> t(64);
  auto pr = [&](const char* n, double c) { printf("%-86s %.2f cycles\n", n, c); };
  for (int l = 0; l < 64; l++) t[l] = 4 * l;              pr("b32 stride 1 word", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 8 * l;              pr("b32 stride 2 words", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 4 * (l < 32 ? 2 * l : 2 * (l - 32) + 1);   pr("b32 stride 2 words, upper half-wave on the odd words", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 4 * (l < 32 ? 2 * l : 2 * l + 1);         pr("b32 stride 2 words, upper half-wave shifted by one word (words 65, 67, ...)", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 4 * (2 * l + ((l >> 4) & 1));             pr("b32 stride 2 words, odd 16-lane rows shifted by one word", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 16 * l;             pr("b32 stride 4 words", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 4 * (l & 31);       pr("b32 two half-waves on the same 32 words (broadcast pairs)", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 4 * (l & 15);       pr("b32 four rows on the same 16 words", run<0>(t));
  for (int l = 0; l < 64; l++) t[l] = 8 * l;              pr("b64 stride 2 words (aligned)", run<1>(t));
  for (int l = 0; l < 64; l++) t[l] = 16 * l;             pr("b64 stride 4 words", run<1>(t));
  for (int l = 0; l < 64; l++) t[l] = 8 * l;              pr("write b64 stride 2 words", run<2>(t));
  for (int l = 0; l < 64; l++) t[l] = 4 * l;              pr("write b32 stride 1 word", run<3>(t));
  for (int l = 0; l < 64; l++) t[l] = 8 * l;              pr("write b32 stride 2 words", run<3>(t));

  // ---- which lanes share a pass with lane 0, and how many banks: all lanes read word 0 (a broadcast) except lane X, which reads word W
  for (int kind = 0; kind < 3; kind++) {
    printf("two-lane probe, %s: rows = lane X, columns = word offset of lane X {1|2, 16, 32, 64, 128}; entries = cycles\n", kind == 0 ? "ds_read_b32" : kind == 1 ? "ds_read_b64" : "ds_write_b64");
    for (int X : {1, 7, 8, 15, 16, 24, 31, 32, 40, 47, 48, 56, 63}) {
      printf("  X=%2d:", X);
      for (int W : {2, 16, 32, 64, 128}) {
        for (int l = 0; l < 64; l++) t[l] = 0;
        t[X] = 4 * W;
        const double c = kind == 0 ? run<0>(t) : kind == 1 ? run<1>(t) : run<2>(t);
        printf(" %5.2f", c);
      }
      printf("\n");
    }
  }

  {
    const int bb[3] = {0, 36, 72};
    printf("two rows per instruction, kernel map (second row 108 words further):\n");
    for (int sh : {-4, -2, 0, 2, 4}) { char nm[160]; snprintf(nm, sizeof nm, "  ds_read2_b64 even rotation %+d of two rows", sh); pr(nm, run<4>(v4tab(bb, sh, 0))); }
    for (int sh : {-3, -1, 1, 3}) { char nm[160]; snprintf(nm, sizeof nm, "  ds_read2_b32 first halves of odd rotation %+d of two rows", sh); pr(nm, run<5>(v4tab(bb, sh, 0))); }
    for (int sh : {-3, -1, 1, 3}) { char nm[160]; snprintf(nm, sizeof nm, "  ds_read2_b32 both halves (adjacent words) of odd rotation %+d", sh); pr(nm, run<8>(v4tab(bb, sh, 0))); }
    pr("  ds_write2_b64 own pair into two rows", run<6>(v4tab(bb, 0, 0)));
    for (int l = 0; l < 64; l++) t[l] = 16 * l;
    pr("  ds_write_b128 stride 16 B", run<7>(t));
    for (int l = 0; l < 64; l++) t[l] = 8 * l;
    pr("  ds_write2_b64 stride 8 B into two rows", run<6>(t));
    pr("  ds_read2_b64 stride 8 B of two rows", run<4>(t));
  }
  const int b0[3] = {0, 36, 72}, b1[3] = {0, 37, 74}, b2[3] = {0, 36, 73}, b3[3] = {0, 38, 76}, b4[3] = {0, 37, 72}, b5[3] = {0, 40, 80}, b6[3] = {0, 41, 82};
  struct { const char* n; const int* b; } lay[] = {{"bases 0,36,72 (now)", b0}, {"bases 0,37,74", b1}, {"bases 0,36,73", b2}, {"bases 0,38,76", b3}, {"bases 0,37,72", b4}, {"bases 0,40,80", b5}, {"bases 0,41,82", b6}};
  for (auto& L : lay) {
    char nm[200];
    for (int sh : {-3, -1, 1, 3}) {
      snprintf(nm, sizeof nm, "kernel map, %s: b32 odd rotation %+d, first half", L.n, sh); pr(nm, run<0>(v4tab(L.b, sh, 0)));
      snprintf(nm, sizeof nm, "kernel map, %s: b32 odd rotation %+d, second half", L.n, sh); pr(nm, run<0>(v4tab(L.b, sh, 1)));
    }
    for (int sh : {-4, -2, 0, 2, 4}) {
      if (L.b[1] % 2 == 0 && L.b[2] % 2 == 0) { snprintf(nm, sizeof nm, "kernel map, %s: b64 even rotation %+d", L.n, sh); pr(nm, run<1>(v4tab(L.b, sh, 0))); }
    }
    if (L.b[1] % 2 == 0 && L.b[2] % 2 == 0) { snprintf(nm, sizeof nm, "kernel map, %s: write b64 own pair", L.n); pr(nm, run<2>(v4tab(L.b, 0, 0))); }
  }
  return 0;
}
