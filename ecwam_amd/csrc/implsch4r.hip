// Launcher of the RARE builds of k_implsch4 (implsch_v4.h): what no registered test configuration of the reference selects -- ISNONLIN = 2
// (TRANSF_SNL + PEAK_ANG), LCIWA2 (sdice2.F90), the NEMO ice stress and strain LWNEMOCOUWRS / LWNEMOCOUSTRN (wnfluxes.F90:178-196,
// cimsstrn.F90), friction-velocity forcing ICODE = 1, 2 (airsea.F90:100-117), LWVFLX_SNL = F, and ISNONLIN = 1 beside LLGCBZ0 / LLNORMAGAM or
// IPHYS = 0 -- decided at run time inside two builds per direction count: flag sets A and B with IPHYS = 1 (EXT + ENHMC), and IPHYS = 0 on
// flag set A (JAN + ENHMC).  A translation unit of its own: the builds compile beside those of implsch4.hip / implsch4x.hip.
#include "implsch_v4_launch.h"

// jan: 1 = IPHYS 0.  Returns 0 when launched, -1 when no instantiation covers the configuration (ecwam_hip_create refuses those).
// V4R_PREC selects the precision this object instantiates: 1 = single (implsch4r.o), 2 = double (implsch4rd.o).
// V4R_DP = 2 (the product's implsch4rd.o since round 6): the double precision builds as the two-kernel split (launch4 SPLIT; + 12 % time, the
// context owns wind-input rows for them).  As ONE function (V4R_DP = 1) the double precision builds end in a memory access fault or in wrong
// numbers at -O3 on their first launch, with every rare switch off at run time, while -O2, -O1, -O3 with index assertions on every table and
// row access and -O3 as the split run clean and bit-identical (profiles/r05_rare_dp_rootcause.txt: code generation of one 280 KB function
// with > 256 VGPRs + AGPR copies + > 100 SGPR spills + a call; no small reproducer).  Round 5 shipped the one function at -O2; the split does
// not depend on the optimisation level, so the next compiler cannot bring the fault back.
#ifndef V4R_PREC
#define V4R_PREC 3
#endif
#ifndef V4R_DP
#define V4R_DP 1
#endif
template <typename T>
int launch_implsch4r(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                     void* fin, double* w2n, void* gfast, int gk, void* wi, int NANG, int NFRE, int r1, int r2, int nh, int jan, hipStream_t s) {
  constexpr bool SP = sizeof(T) == 4;
  if constexpr (!SP && !V4R_DP) return -1;
  else {
    if (kijl - kijs <= 0) return 0;
    if (NFRE != V4_NFRE) return -1;
#define V4_ARGS tab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, fin, w2n, gfast, gk, wi, s
    // double precision: the two-kernel split (V4R_DP = 2); single precision: one kernel unless the whole library is built as the split
    constexpr bool SPL = (!SP && V4R_DP == 2) || V4_SPLIT_ALL != 0;
    // (the IPHYS = 0 build carries LLGCBZ0 / LLNORMAGAM too: EXT and JAN)
    if (NANG == 48 && r1 == 1 && r2 == 4 && nh == 11)
      return jan ? launch4<T, 48, 2, 1, 4, 11, true, true, true, true, SPL>(V4_ARGS) : launch4<T, 48, 2, 1, 4, 11, true, false, true, true, SPL>(V4_ARGS);
    if (NANG == 36 && r1 == 1 && r2 == 3 && nh == 8)
      return jan ? launch4<T, 36, 3, 1, 3, 8, true, true, true, true, SPL>(V4_ARGS) : launch4<T, 36, 3, 1, 3, 8, true, false, true, true, SPL>(V4_ARGS);
    if (NANG == 24 && r1 == 0 && r2 == 2 && nh == 5)
      return jan ? launch4<T, 24, SP ? 5 : 4, 0, 2, 5, true, true, true, true, SPL>(V4_ARGS) : launch4<T, 24, SP ? 5 : 4, 0, 2, 5, true, false, true, true, SPL>(V4_ARGS);
    if (NANG == 12 && r1 == 0 && r2 == 1 && nh == 3)
      return jan ? launch4<T, 12, SP ? 10 : 5, 0, 1, 3, true, true, true, true, SPL>(V4_ARGS) : launch4<T, 12, SP ? 10 : 5, 0, 1, 3, true, false, true, true, SPL>(V4_ARGS);
#undef V4_ARGS
    return -1;
  }
}
#define V4R_SIG(T) template int launch_implsch4r<T>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t)
#if V4R_PREC & 1
V4R_SIG(float);
#endif
#if V4R_PREC & 2
V4R_SIG(double);
// 1 when the double precision builds of this unit need the context's wind-input rows (the two-kernel split)
int implsch4r_dp_split() { return V4R_DP == 2 ? 1 : 0; }
#endif
