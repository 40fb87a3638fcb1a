// Launcher of the IMPLSCH kernel generations 1-3 (implsch_v1.h: lane = direction, three tiles; implsch_v2.h: two tiles, fused sweep;
// implsch_v3.h: three points per wavefront).  The fourth generation lives in implsch4.hip.
#include "implsch_v1.h"
#include "implsch_v2.h"
#include "implsch_v3.h"

template <typename T>
int launch_implsch(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                   double* w2n, void* dbg, int NANG, int NFRE, int variant, hipStream_t s) {
  const int n = kijl - kijs;
  if (n <= 0) return 0;
  const bool norma = (variant & 16) != 0;  // LLNORMAGAM, packed by capi.hip
  const bool v3ok = (variant & 64) != 0;   // the configuration fits the three-points-per-wavefront kernel (capi.hip)
  const bool rare = (variant & 32) != 0;   // any of LLGCBZ0 / LCIWA2 / LCIWA3 / LCISCAL / LWNEMOCOU: the build that carries those branches
  variant &= 15;
  const bool variant2 = (variant == 2);
  if constexpr (sizeof(T) == 4) {
    // third kernel generation (three points per wavefront, implsch_v3.h): flag set A without the optional branches, 36 directions.
    // ECWAM_HIP_IMPLSCH_V3=0 falls back to k_implsch2 (diagnostics, and the parity test that keeps both generations checked)
    const char* e3 = getenv("ECWAM_HIP_IMPLSCH_V3");
    if (!(e3 && atoi(e3) == 0) && v3ok && variant2 && !norma && !rare && NANG == 2 * V3G && !w2n && !dbg) {
      const size_t per3 = (size_t)((V3P * (NANG * NFRE + V3_NFAC * NFRE + 4 * NFRE) + V3P * NSC + 3) & ~3) * sizeof(float);
      int wpb3 = 1;
      { const char* ew = getenv("ECWAM_HIP_V3_WPB"); if (ew) wpb3 = atoi(ew); }   // diagnostics
#define LAUNCH3(W)                                                                                                             \
  do {                                                                                                                         \
    const size_t shmem = per3 * W;                                                                                             \
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_implsch3<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL((k_implsch3<W>), dim3((n + V3P * W - 1) / (V3P * W)), dim3(64 * W), shmem, s, (const DevTab<float>*)tab, kijs, \
                       kijl, (float*)fl1, (const float*)wvprpt, (float*)ff, (float*)intf, mij, (float*)xllws);                 \
  } while (0)
      if (wpb3 == 4) LAUNCH3(4);
      else if (wpb3 == 2) LAUNCH3(2);
      else LAUNCH3(1);
#undef LAUNCH3
      return 0;
    }
  }
  const int ntile = (variant == 2) ? 2 : 3;
  const int NAP = variant2 ? NANG : (NANG | 1);
  const int nscr = variant2 ? 0 : 64;
  const size_t per_wave = (size_t)(ntile * NFRE * NAP + nscr) * sizeof(T) + (variant == 2 ? NSC * sizeof(T) : 0);
  // waves (= points) per block: the choice that fits the most waves into the 160 KiB of LDS of a CU; ties go to the larger
  // block, which amortises the lane-per-point scalar stages of variant 2 over more points
  static const int cands3[] = {4, 2, 1}, cands2[] = {3, 1, 1};
  const int* cands = (variant == 2) ? cands2 : cands3;
  const int ncand = 3;
  // register-limited residency: variant 2 is compiled for 3 (sp) / 2 (dp) waves per SIMD, variant 1 uses 151 / 256 VGPRs
  const int capw = 4 * ((sizeof(T) == 4) ? (variant == 2 ? 4 : 3) : 2);
  int wpb = 1, best = 0;
  for (int i = 0; i < ncand; i++) {
    const int cand = cands[i];
    int nb = (int)((160 * 1024) / (per_wave * cand));
    if (nb * cand > capw) nb = capw / cand;
    const int waves = nb * cand;
    if (waves > best) { best = waves; wpb = cand; }
  }
  if (best == 0) return 1;
  { const char* e_ = getenv("ECWAM_HIP_IMPLSCH_WPB"); if (e_ && variant == 2) { const int w = atoi(e_); if (w == 3 || w == 1) wpb = w; } }
  size_t shmem = per_wave * wpb;
  { const char* e_ = getenv("ECWAM_HIP_IMPLSCH_PADLDS"); if (e_) shmem += (size_t)atoi(e_); }  // diagnostics: lower the residency
  const int blocks = (n + wpb - 1) / wpb;
#define LAUNCHK(KFN)                                                                                                         \
  do {                                                                                                                       \
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)KFN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL(KFN, dim3(blocks), dim3(64 * wpb), shmem, s, (const DevTab<T>*)tab, kijs, kijl, (T*)fl1,              \
                       (const T*)wvprpt, (T*)ff, (T*)intf, mij, (T*)xllws, w2n, (T*)dbg);                                         \
  } while (0)
#define LAUNCH(K, W) LAUNCHK((K<T, W>))
  if (variant2) {
#define LAUNCH2(W)                                                                                                           \
  do {                                                                                                                       \
    if (norma && rare) LAUNCHK((k_implsch2<T, W, true, true>));                                                              \
    else if (norma) LAUNCHK((k_implsch2<T, W, true, false>));                                                                \
    else if (rare) LAUNCHK((k_implsch2<T, W, false, true>));                                                                 \
    else LAUNCHK((k_implsch2<T, W, false, false>));                                                                          \
  } while (0)
    if (wpb == 3) LAUNCH2(3);
    else LAUNCH2(1);
#undef LAUNCH2
  } else {
    if (wpb == 4) LAUNCH(k_implsch, 4);
    else if (wpb == 2) LAUNCH(k_implsch, 2);
    else LAUNCH(k_implsch, 1);
  }
#undef LAUNCH
#undef LAUNCHK
  return 0;
}
template int launch_implsch<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, double*, void*, int, int, int, hipStream_t);
template int launch_implsch<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, double*, void*, int, int, int, hipStream_t);
