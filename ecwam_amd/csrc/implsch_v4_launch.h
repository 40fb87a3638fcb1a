// Launch glue of k_implsch4 (implsch_v4.h), shared by the translation units that instantiate it: implsch4.hip (flag sets A and B) and
// implsch4x.hip (IPHYS = 0, ISNONLIN = 1).
#pragma once
#include "implsch_common.h"
#include "implsch_point.h"
#include "implsch_v4.h"

#ifndef V4_DP36_PP
#define V4_DP36_PP 3
#endif

template <typename T, int NANG, int PP>
static constexpr size_t v4_lds_bytes() {
  return (size_t)((V4_NFRE + V4_NSTG) * PP * NANG + PP * V4_NFRE * 4 + 2 * V4_PLN(PP, NANG) + PP * NSC) * sizeof(T);
}

#ifndef V4_SPLIT_ALL
#define V4_SPLIT_ALL 0   // 1: every build of the translation unit runs as the two-kernel split (build variant "split": the round-5 prototype)
#endif

// SPLIT: PART 1 and PART 2 of the kernel one after the other instead of the one kernel (needs the context's wi rows, n x NANG x NFRE)
template <typename T, int NANG, int PP, int R1, int R2, int NH, bool EXT, bool JAN = false, bool ENHMC = false, bool RARE = false, bool SPLIT = (V4_SPLIT_ALL != 0)>
static int launch4(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws, void* fin,
                   double* w2n, void* gfast, int gk, void* wi, hipStream_t s) {
  const int n = kijl - kijs;
  constexpr size_t shmem0 = v4_lds_bytes<T, NANG, PP>();
  static_assert(shmem0 <= 160 * 1024, "LDS");
  // diagnostics (tools/occupancy_sweep4.py): ECWAM_HIP_IMPLSCH_PADLDS=<bytes> pads the LDS request, i.e. lowers the resident waves per CU
  static const int pad = [] { const char* e = getenv("ECWAM_HIP_IMPLSCH_PADLDS"); return e ? atoi(e) : 0; }();
  const size_t shmem = (pad > 0 && shmem0 + (size_t)pad <= 160 * 1024) ? shmem0 + (size_t)pad : shmem0;
  if (SPLIT && !wi) return -1;
  // the scalar start of the step (first TAUT_Z0), one point per lane, into the rows of fin
  hipLaunchKernelGGL((k_implsch4_pre<T, EXT, RARE>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)ff, (T*)fin);
#define V4_KARGS (const DevTab<T>*)tab, kijs, kijl, (T*)fl1, (const T*)wvprpt, (T*)ff, (T*)intf, mij, (T*)xllws, (T*)fin, (T*)gfast, gk, (T*)wi, V4Adv<T>{}
  if constexpr (SPLIT) {
    auto k1 = k_implsch4<T, NANG, PP, R1, R2, NH, EXT, JAN, ENHMC, RARE, 1>;
    auto k2 = k_implsch4<T, NANG, PP, R1, R2, NH, EXT, JAN, ENHMC, RARE, 2>;
    if (shmem > 64 * 1024) {
      (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    }
    hipLaunchKernelGGL(k1, dim3((n + PP - 1) / PP), dim3(64), shmem, s, V4_KARGS);
    hipLaunchKernelGGL(k2, dim3((n + PP - 1) / PP), dim3(64), shmem, s, V4_KARGS);
  } else {
    auto kfn = k_implsch4<T, NANG, PP, R1, R2, NH, EXT, JAN, ENHMC, RARE, 0>;
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(kfn, dim3((n + PP - 1) / PP), dim3(64), shmem, s, V4_KARGS);
  }
#undef V4_KARGS
  // the scalar end of the step (second STRESSO, WNFLUXES), one point per lane, from the rows the kernel above left in fin
  hipLaunchKernelGGL((k_implsch4_fin<T, EXT>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)fin, (T*)ff,
                     (T*)intf, w2n);
  return 0;
}

// The one-kernel WAMINTGR step (ADV builds of k_implsch4: the tile load is PROPAGS2 of the wave's points from the rows of adv.f_in; fl1 = the
// rows the new spectrum is stored to, another buffer).  The grid is a whole number of rounds of the 8 XCDs (adv.xcd_walk).
template <typename T, int NANG, int PP, int R1, int R2, int NH, bool EXT, int ADV>
static int launch4_adv(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws, void* fin,
                       double* w2n, void* gfast, int gk, const V4Adv<T>& adv, hipStream_t s) {
  const int n = kijl - kijs;
  constexpr size_t shmem = v4_lds_bytes<T, NANG, PP>();
  static_assert(shmem <= 160 * 1024, "LDS");
  hipLaunchKernelGGL((k_implsch4_pre<T, EXT, false>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)ff, (T*)fin);
  auto kfn = k_implsch4<T, NANG, PP, R1, R2, NH, EXT, false, false, false, 0, ADV>;
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  int nb = (n + PP - 1) / PP;
  if (adv.xcd_walk == 1) nb = (nb + 7) & ~7;
  else if (adv.xcd_walk > 1) nb = (nb + 8 * adv.xcd_walk - 1) / (8 * adv.xcd_walk) * (8 * adv.xcd_walk);
  hipLaunchKernelGGL(kfn, dim3(nb), dim3(64), shmem, s, (const DevTab<T>*)tab, kijs, kijl, (T*)fl1, (const T*)wvprpt, (T*)ff, (T*)intf, mij, (T*)xllws,
                     (T*)fin, (T*)gfast, gk, (T*)nullptr, adv);
  hipLaunchKernelGGL((k_implsch4_fin<T, EXT>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)fin, (T*)ff,
                     (T*)intf, w2n);
  return 0;
}
