// Launcher of the ADV builds of k_implsch4 (implsch_v4.h): the one-kernel WAMINTGR step of round 6 -- PROPAGS2 (IREFRA = 0, one time step for
// every frequency, no obstructions) fused into IMPLSCH's tile load (wamintgr.F90:94-146, propag_wam.F90:247-251, propags2.F90:99-121).
// Flag sets A and B (EXT), IPHYS = 1, ISNONLIN = 0, ICODE = 3; NFRE = 36; NANG = 36, single and double precision; ADV = 1 the plain step, ADV = 3 the
// last advection step of the native O1280 cycle (fast waves M <= IFRELFMAX with their own time step, read from the compact rows of their sub-steps);
// ADV = 5 / 7 the same with the sub-grid obstructions of LSUBGRID (the reference's default on real bathymetry).  Everything else runs the
// two kernels (ecwam_hip_propags2_otf + ecwam_hip_implsch), which is also the A/B partner: the results are bit-identical.
#include "implsch_v4_launch.h"
#include "implsch_adv_args.h"

#ifndef V4_ADV_PROBE
#define V4_ADV_PROBE 0
#endif

// returns 0 when launched, -1 when no ADV build covers the configuration (the caller then runs the two kernels)
template <typename T>
int launch_implsch4_adv(const void* tab, int kijs, int kijl, void* fl_out, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws, void* fin,
                        double* w2n, const Implsch4AdvArgs* a, int NANG, int NFRE, int r1, int r2, int nh, int ext, hipStream_t s) {
  if (kijl - kijs <= 0) return 0;
  if (NFRE != V4_NFRE) return -1;
  V4Adv<T> adv;
  adv.f_in = (const T*)a->f_in; adv.klon = a->klon; adv.klat = a->klat; adv.kcor = a->kcor; adv.cg = (const T*)a->cg; adv.pt = (const T*)a->pt;
  adv.dirT = (const T*)a->dirT; adv.dirI = a->dirI; adv.xdella = (T)a->xdella; adv.delpro = (T)a->delpro; adv.m0 = a->m0; adv.m1 = a->m1;
  adv.xcd_walk = a->xcd_walk;
  adv.gin = (const T*)a->gin; adv.delpro_lf = (T)a->delpro_lf; adv.gin_k = a->gin_k; adv.mlf = a->mlf; adv.obs = (const T*)a->obs;
#define V4_ARGS tab, kijs, kijl, fl_out, wvprpt, ff, intf, mij, xllws, fin, w2n, a->gfast, a->gfast_k, adv, s
  constexpr int PP36 = sizeof(T) == 4 ? 3 : V4_DP36_PP;
  if (NANG == 36 && r1 == 1 && r2 == 3 && nh == 8) {
#if V4_ADV_PROBE
    if constexpr (sizeof(T) == 4)
      if (a->mode == 2) return ext ? -1 : launch4_adv<T, 36, 3, 1, 3, 8, false, 2>(V4_ARGS);
#endif
#if !ECWAM_HIP_CTU_STRICT      // (the strict build of the weights has the plain form only: the caller runs the two kernels for the rest)
    if (a->mode == 1 && a->obs) {      // LSUBGRID
      if (a->gin) return ext ? launch4_adv<T, 36, PP36, 1, 3, 8, true, 7>(V4_ARGS) : launch4_adv<T, 36, PP36, 1, 3, 8, false, 7>(V4_ARGS);
      if (a->mlf == 0) return ext ? launch4_adv<T, 36, PP36, 1, 3, 8, true, 5>(V4_ARGS) : launch4_adv<T, 36, PP36, 1, 3, 8, false, 5>(V4_ARGS);
      return -1;
    }
    if (a->mode == 1 && a->gin) return ext ? launch4_adv<T, 36, PP36, 1, 3, 8, true, 3>(V4_ARGS) : launch4_adv<T, 36, PP36, 1, 3, 8, false, 3>(V4_ARGS);
#endif
    if (a->mode == 1 && !a->gin && a->mlf == 0 && !a->obs)
      return ext ? launch4_adv<T, 36, PP36, 1, 3, 8, true, 1>(V4_ARGS) : launch4_adv<T, 36, PP36, 1, 3, 8, false, 1>(V4_ARGS);
  }
#undef V4_ARGS
  return -1;
}
template int launch_implsch4_adv<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, const Implsch4AdvArgs*, int, int, int, int, int, int, hipStream_t);
template int launch_implsch4_adv<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, const Implsch4AdvArgs*, int, int, int, int, int, int, hipStream_t);
