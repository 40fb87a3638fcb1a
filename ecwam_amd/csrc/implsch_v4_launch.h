// Launch glue of k_implsch4 (implsch_v4.h), shared by the translation units that instantiate it: implsch4.hip (flag sets A and B) and
// implsch4x.hip (IPHYS = 0, ISNONLIN = 1).
#pragma once
#include "implsch_common.h"
#include "implsch_v2.h"
#include "implsch_v4.h"

#ifndef V4_DP36_PP
#define V4_DP36_PP 3
#endif

template <typename T, int NANG, int PP>
static constexpr size_t v4_lds_bytes() {
  return (size_t)((V4_NFRE + V4_NSTG) * PP * NANG + PP * V4_NFRE * V4_NFAC + PP * NSC) * sizeof(T);
}

template <typename T, int NANG, int PP, int R1, int R2, int NH, bool EXT, bool JAN = false, bool ENHMC = false, bool RARE = false>
static int launch4(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws, void* fin,
                   double* w2n, void* gfast, int gk, hipStream_t s) {
  const int n = kijl - kijs;
  constexpr size_t shmem = v4_lds_bytes<T, NANG, PP>();
  static_assert(shmem <= 160 * 1024, "LDS");
  auto kfn = k_implsch4<T, NANG, PP, R1, R2, NH, EXT, JAN, ENHMC, RARE>;
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  // the scalar start of the step (first TAUT_Z0), one point per lane, into the rows of fin
  hipLaunchKernelGGL((k_implsch4_pre<T, EXT, RARE>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)ff, (T*)fin);
  hipLaunchKernelGGL(kfn, dim3((n + PP - 1) / PP), dim3(64), shmem, s, (const DevTab<T>*)tab, kijs, kijl, (T*)fl1, (const T*)wvprpt, (T*)ff,
                     (T*)intf, mij, (T*)xllws, (T*)fin, (T*)gfast, gk);
  // the scalar end of the step (second STRESSO, WNFLUXES), one point per lane, from the rows the kernel above left in fin
  hipLaunchKernelGGL((k_implsch4_fin<T, EXT>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)fin, (T*)ff,
                     (T*)intf, w2n);
  return 0;
}

