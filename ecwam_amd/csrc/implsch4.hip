// Launcher of the common builds of k_implsch4 (implsch_v4.h: PP sea points per wavefront on adjacent direction pairs, every rotation in K
// through LDS rows): implsch.F90:10-468 on flag set A (LLGCBZ0 = F, LLNORMAGAM = F) and, in the EXT build, flag set B (either or both of
// them T: cy49r1 / cy50r1), with or without the sea-ice damping LCIWA1 / LCIWA3 / LCISCAL and the NEMO coupling outputs of LWNEMOCOU; IPHYS = 1,
// ISNONLIN = 0, ICODE = 3; NFRE = 36, NANG = 48 / 36 / 24 / 12, single and double precision.  (IPHYS 0 / ISNONLIN 1: implsch4x.hip; every
// other switch: implsch4r.hip.)
#include "implsch_v4_launch.h"

// returns 0 when launched, -1 when no instantiation covers (NANG, r1, r2, nh): ecwam_hip_create refuses those configurations
template <typename T>
int launch_implsch4(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                    void* fin, double* w2n, void* gfast, int gk, void* wi, int NANG, int NFRE, int r1, int r2, int nh, int ext, hipStream_t s) {
  if (kijl - kijs <= 0) return 0;
  if (NFRE != V4_NFRE) return -1;
  constexpr bool SP = sizeof(T) == 4;
#define V4_ARGS tab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, fin, w2n, gfast, gk, wi, s
  if (NANG == 48 && r1 == 1 && r2 == 4 && nh == 11)      // two points per wavefront, 24 lanes each
    return ext ? launch4<T, 48, 2, 1, 4, 11, true>(V4_ARGS) : launch4<T, 48, 2, 1, 4, 11, false>(V4_ARGS);
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
// 1 when every build of the library runs as the two-kernel split (build variant "split"): the context then owns wind-input rows
int implsch4_split_all() { return V4_SPLIT_ALL != 0 ? 1 : 0; }
template int launch_implsch4<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
template int launch_implsch4<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
