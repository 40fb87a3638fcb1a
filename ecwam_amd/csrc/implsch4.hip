// Launcher of the fourth IMPLSCH kernel generation (implsch_v4.h): PP sea points per wavefront on adjacent direction pairs, every
// rotation in K through LDS rows.  implsch.F90:10-468 on flag set A (LLGCBZ0 = F, LLNORMAGAM = F) and, in the EXT build of the
// kernel, flag set B (either or both of them T: cy49r1 / cy50r1); sea-ice damping LCIWA1 / LCIWA3 / LCISCAL (no LCIWA2), the NEMO coupling outputs of LWNEMOCOU (without ice stress, strain, break-up), IPHYS = 1, ISNONLIN = 0,
// ICODE = 3, NFRE = 36, NANG = 36 / 24 / 12, single and double precision.  Everything else runs k_implsch2 (implsch.hip).
#include "implsch_common.h"
#include "implsch_v2.h"
#include "implsch_v4.h"

template <typename T, int NANG, int PP>
static constexpr size_t v4_lds_bytes() {
  return (size_t)((V4_NFRE + V4_NSTG) * PP * NANG + PP * V4_NFRE * V4_NFAC + PP * NSC) * sizeof(T);
}

template <typename T, int NANG, int PP, int R1, int R2, int NH, bool EXT>
static int launch4(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws, void* fin,
                   double* w2n, hipStream_t s) {
  const int n = kijl - kijs;
  constexpr size_t shmem = v4_lds_bytes<T, NANG, PP>();
  static_assert(shmem <= 160 * 1024, "LDS");
  auto kfn = k_implsch4<T, NANG, PP, R1, R2, NH, EXT>;
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  // the scalar start of the step (first TAUT_Z0), one point per lane, into the rows of fin
  hipLaunchKernelGGL((k_implsch4_pre<T, EXT>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)ff, (T*)fin);
  hipLaunchKernelGGL(kfn, dim3((n + PP - 1) / PP), dim3(64), shmem, s, (const DevTab<T>*)tab, kijs, kijl, (T*)fl1, (const T*)wvprpt, (T*)ff,
                     (T*)intf, mij, (T*)xllws, (T*)fin);
  // the scalar end of the step (second STRESSO, WNFLUXES), one point per lane, from the rows the kernel above left in fin
  hipLaunchKernelGGL((k_implsch4_fin<T, EXT>), dim3((n + 63) / 64), dim3(64), 0, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)fin, (T*)ff,
                     (T*)intf, w2n);
  return 0;
}

// returns 0 when launched, -1 when no instantiation covers (NANG, r1, r2, nh): the caller falls back to k_implsch2
template <typename T>
int launch_implsch4(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                    void* fin, double* w2n, int NANG, int NFRE, int r1, int r2, int nh, int ext, hipStream_t s) {
  if (kijl - kijs <= 0) return 0;
  if (NFRE != V4_NFRE) return -1;
  constexpr bool SP = sizeof(T) == 4;
#ifndef V4_DP36_PP
#define V4_DP36_PP 3
#endif
#define V4_ARGS tab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, fin, w2n, s
  if (NANG == 36 && r1 == 1 && r2 == 3 && nh == 8)
    return ext ? launch4<T, 36, SP ? 3 : V4_DP36_PP, 1, 3, 8, true>(V4_ARGS) : launch4<T, 36, SP ? 3 : V4_DP36_PP, 1, 3, 8, false>(V4_ARGS);
  if (NANG == 24 && r1 == 0 && r2 == 2 && nh == 5)
    return ext ? launch4<T, 24, SP ? 5 : 4, 0, 2, 5, true>(V4_ARGS) : launch4<T, 24, SP ? 5 : 4, 0, 2, 5, false>(V4_ARGS);
  if (NANG == 12 && r1 == 0 && r2 == 1 && nh == 3)
    return ext ? launch4<T, 12, SP ? 10 : 5, 0, 1, 3, true>(V4_ARGS) : launch4<T, 12, SP ? 10 : 5, 0, 1, 3, false>(V4_ARGS);
#undef V4_ARGS
  return -1;
}
// elements of the working precision per sea point the caller provides in fin (indexed by the absolute point number, like FL1)
int implsch4_fin_row() { return V4_NFIN; }
template int launch_implsch4<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, int, int, int, int, int, int, hipStream_t);
template int launch_implsch4<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, int, int, int, int, int, int, hipStream_t);
