// Launcher of the ADV builds of k_implsch4 (implsch_v4.h): the one-kernel WAMINTGR step of round 6 -- PROPAGS2 (IREFRA = 0, one time step for
// every frequency, no obstructions) fused into IMPLSCH's tile load (wamintgr.F90:94-146, propag_wam.F90:247-251, propags2.F90:99-121).
// Flag sets A and B (EXT), IPHYS = 1, ISNONLIN = 0, ICODE = 3; NFRE = 36; NANG = 48 / 36 / 24 / 12 in single, 36 in double precision; ADV = 1 the plain step, ADV = 3 the
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
  // one direction count: the plain step (ADV 1), with fast waves (3), with obstructions (5), with both (7); each for flag sets A and B (EXT)
#define V4_ADV_CASE(NA, PPV, R1V, R2V, NHV, LFOK)                                                                                         \
  if (NANG == NA && r1 == R1V && r2 == R2V && nh == NHV) {                                                                              \
    if (a->mode != 1) return -1;                                                                                                        \
    const int form = (a->gin ? 2 : 0) | (a->obs ? 4 : 0);                                                                               \
    if (!a->gin && a->mlf != 0) return -1;                                                                                              \
    if (form == 0) return ext ? launch4_adv<T, NA, PPV, R1V, R2V, NHV, true, 1>(V4_ARGS) : launch4_adv<T, NA, PPV, R1V, R2V, NHV, false, 1>(V4_ARGS); \
    if constexpr (!ECWAM_HIP_CTU_STRICT) {      /* (the strict build of the weights has the plain form only: the caller runs the two kernels) */ \
      if (form == 4) return ext ? launch4_adv<T, NA, PPV, R1V, R2V, NHV, true, 5>(V4_ARGS) : launch4_adv<T, NA, PPV, R1V, R2V, NHV, false, 5>(V4_ARGS); \
      if constexpr (LFOK) {                                                                                                             \
        if (form == 2) return ext ? launch4_adv<T, NA, PPV, R1V, R2V, NHV, true, 3>(V4_ARGS) : launch4_adv<T, NA, PPV, R1V, R2V, NHV, false, 3>(V4_ARGS); \
        if (form == 6) return ext ? launch4_adv<T, NA, PPV, R1V, R2V, NHV, true, 7>(V4_ARGS) : launch4_adv<T, NA, PPV, R1V, R2V, NHV, false, 7>(V4_ARGS); \
      }                                                                                                                                 \
    }                                                                                                                                   \
    return -1;                                                                                                                          \
  }
  constexpr bool SP = sizeof(T) == 4;
#if V4_ADV_PROBE
  if constexpr (SP)
    if (NANG == 36 && r1 == 1 && r2 == 3 && nh == 8 && a->mode == 2) return ext ? -1 : launch4_adv<T, 36, 3, 1, 3, 8, false, 2>(V4_ARGS);
#endif
  V4_ADV_CASE(36, (SP ? 3 : V4_DP36_PP), 1, 3, 8, true)
  // The other direction counts in single precision only.  Their double precision builds were built and measured in round 6 and are NOT
  // shipped: at -O3 the 12-direction builds and the fast-wave builds at 24 directions give wrong numbers (34 000 .. 58 000 of the bins of a
  // 1 109-point grid differ from the two kernels, errors of order one from frequency 7 on), at -O1 the same source is bit-identical to the
  // two kernels in every case -- the code-generation problem of the large double precision functions (256 VGPRs + 70 .. 84 AGPR copies + up to
  // 100 SGPR spills) that the RARE builds met in round 4 (implsch4r.hip; profiles/r05_rare_dp_rootcause.txt, r06_fused_step_experiments.txt).
  // Double precision at 48 / 24 / 12 directions runs the two kernels; the 36-direction builds are held to the two kernels bit for bit by
  // tests/test_gpu_fused.py in every form.
  if constexpr (SP) {
    V4_ADV_CASE(24, 5, 0, 2, 5, true)
    V4_ADV_CASE(12, 10, 0, 1, 3, true)
    V4_ADV_CASE(48, 2, 1, 4, 11, false)      // (two points per wave: the LDS the tile load borrows has no room for the fast waves' second table)
  }
#undef V4_ADV_CASE
#undef V4_ARGS
  return -1;
}
template int launch_implsch4_adv<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, const Implsch4AdvArgs*, int, int, int, int, int, int, hipStream_t);
template int launch_implsch4_adv<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, const Implsch4AdvArgs*, int, int, int, int, int, int, hipStream_t);
// the forms of the one-kernel step this library holds for a direction count: bit 0 the plain step, bit 1 fast-wave sub-steps (compact input
// rows), bit 2 sub-grid obstructions (ecwam_hip_propags2_implsch_supported)
int implsch4_adv_forms(int NANG, int real_bytes) {
  if (NANG != 48 && NANG != 36 && NANG != 24 && NANG != 12) return 0;
  if (real_bytes == 8 && NANG != 36) return 0;      // (see launch_implsch4_adv)
  return 1 | (ECWAM_HIP_CTU_STRICT ? 0 : 4) | ((ECWAM_HIP_CTU_STRICT || NANG == 48) ? 0 : 2);
}
