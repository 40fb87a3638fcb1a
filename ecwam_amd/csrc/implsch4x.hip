// Launcher of the k_implsch4 builds for the alternate physics of SURVEY.md 8f rank 4 that the registered test configurations select:
// IPHYS = 0 (Janssen wind input + WAM cycle 4 dissipation: etopo1_oper_an_fc_O48_iphys_0) and ISNONLIN = 1 (the DIA scaled per
// interaction frequency), both on flag set A (LLGCBZ0 = F, LLNORMAGAM = F).  A translation unit of its own: the builds compile beside
// those of implsch4.hip.
#include "implsch_v4_launch.h"

// variant: 1 = IPHYS 0, 2 = ISNONLIN 1.  Returns 0 when launched, -1 when no instantiation covers the configuration (ecwam_hip_create refuses those).
template <typename T>
int launch_implsch4x(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                     void* fin, double* w2n, void* gfast, int gk, void* wi, int NANG, int NFRE, int r1, int r2, int nh, int variant, hipStream_t s) {
  if (kijl - kijs <= 0) return 0;
  if (NFRE != V4_NFRE || (variant != 1 && variant != 2)) return -1;
  constexpr bool SP = sizeof(T) == 4;
#define V4_ARGS tab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, fin, w2n, gfast, gk, wi, s
  if (NANG == 48 && r1 == 1 && r2 == 4 && nh == 11)
    return variant == 1 ? launch4<T, 48, 2, 1, 4, 11, false, true, false>(V4_ARGS) : launch4<T, 48, 2, 1, 4, 11, false, false, true>(V4_ARGS);
  if (NANG == 36 && r1 == 1 && r2 == 3 && nh == 8)
    return variant == 1 ? launch4<T, 36, SP ? 3 : V4_DP36_PP, 1, 3, 8, false, true, false>(V4_ARGS)
                        : launch4<T, 36, SP ? 3 : V4_DP36_PP, 1, 3, 8, false, false, true>(V4_ARGS);
  if (NANG == 24 && r1 == 0 && r2 == 2 && nh == 5)
    return variant == 1 ? launch4<T, 24, SP ? 5 : 4, 0, 2, 5, false, true, false>(V4_ARGS) : launch4<T, 24, SP ? 5 : 4, 0, 2, 5, false, false, true>(V4_ARGS);
  if (NANG == 12 && r1 == 0 && r2 == 1 && nh == 3)
    return variant == 1 ? launch4<T, 12, SP ? 10 : 5, 0, 1, 3, false, true, false>(V4_ARGS) : launch4<T, 12, SP ? 10 : 5, 0, 1, 3, false, false, true>(V4_ARGS);
#undef V4_ARGS
  return -1;
}
template int launch_implsch4x<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
template int launch_implsch4x<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
