/*
 * oracle/ora_propag.c -- TEST INFRASTRUCTURE ONLY. See ora.h header ("parity unpinned").
 * Plain-C restatement of the IPROPAGS=2 advection: CTU weight construction
 * (ctuwupdt.F90, ctuwini.F90, ctuw.F90; spherical grid, ICASE=1, IREFRA=0, LSUBGRID=F => OBS*=1)
 * and the PROPAGS2 stencil (propags2.F90:95-121).
 *
 * Point indices are 0-based; the land slot ("NSUP+1", propag_wam.F90:146) is index nland.
 *   KLON[ij][2]  KLAT[ij][2][2]  KCOR[ij][4][2]   (IC = 0 south/west, 1 north/east as in the reference's 1,2)
 *   WLAT[ij][2]  WCOR[ij][4]
 *   spectra F[(npts+1)][NANG][NFRE], weights W*[ij][k][m(NFRE_RED)][...]
 */
#include "ora.h"

/* LSUBGRID: OBS[ij][8][NFRE] = OBSLAT(IJ,M,1:2), OBSLON(IJ,M,1:2), OBSCOR(IJ,M,1:4) used by the next ora_ctuw* calls (NULL: none) */
static const real *g_obs = NULL;
void ora_set_obstructions(const real *obs) { g_obs = obs; }
/* ctuw.F90:703-733: the blocking coefficients scale the weights of the surrounding points (after the checks; SUMWN untouched) */
static void apply_obstructions(int n, int MSTART, int MEND, real *WLONN, real *WLATN, real *WCORN) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  if (!g_obs) return;
  for (int IJ = 0; IJ < n; IJ++)
    for (int K = 0; K < NANG; K++)
      for (int M = MSTART - 1; M < MEND; M++) {
        size_t b = ((size_t)IJ * NANG + K) * NR + M;
        const real *o = g_obs + (size_t)IJ * 8 * NFRE + M;
        for (int IC = 0; IC < 2; IC++) {
          for (int ICL = 0; ICL < 2; ICL++) WLATN[(b * 2 + IC) * 2 + ICL] = WLATN[(b * 2 + IC) * 2 + ICL] * o[IC * NFRE];
          WLONN[b * 2 + IC] = WLONN[b * 2 + IC] * o[(2 + IC) * NFRE];
        }
        for (int ICR = 0; ICR < 4; ICR++)
          for (int ICL = 0; ICL < 2; ICL++)
            WCORN[(b * 4 + ICR) * 2 + ICL] = WCORN[(b * 4 + ICR) * 2 + ICL] * o[(4 + (S.KCR[K][ICR] - 1)) * NFRE];
      }
}

/* ctuwini.F90:58-99 (mutates WLAT/WCOR near land) + :157-164 (DP) */
void ora_ctuwini(int n, int nland, int ngy, const int *KXLT, const real *COSPH, const real *COSPHM1_EXT,
                 const int *KLAT, const int *KCOR, real *WLAT, real *WCOR, real *WLATM1, real *WCORM1, real *DP) {
  for (int IC = 0; IC < 2; IC++)
    for (int IJ = 0; IJ < n; IJ++) {
      int k1 = KLAT[(IJ * 2 + IC) * 2 + 0], k2 = KLAT[(IJ * 2 + IC) * 2 + 1];
      real *w = &WLAT[IJ * 2 + IC];
      if (k1 < nland && k2 < nland) {
      } else if (k1 == nland) {
        if (*w <= C_(0.75)) *w = C_(0.0);
      } else {
        if (*w >= C_(0.5)) *w = C_(1.0);
      }
      WLATM1[IJ * 2 + IC] = C_(1.0) - *w;
    }
  for (int ICR = 0; ICR < 4; ICR++)
    for (int IJ = 0; IJ < n; IJ++) {
      int k1 = KCOR[(IJ * 4 + ICR) * 2 + 0], k2 = KCOR[(IJ * 4 + ICR) * 2 + 1];
      real *w = &WCOR[IJ * 4 + ICR];
      if (k1 < nland && k2 < nland) {
      } else if (k1 == nland) {
        if (*w <= C_(0.75)) *w = C_(0.0);
      } else {
        if (*w > C_(0.5)) *w = C_(1.0);
      }
      WCORM1[IJ * 4 + ICR] = C_(1.0) - *w;
    }
  for (int IC = 1; IC <= 2; IC++)
    for (int IJ = 0; IJ < n; IJ++) {
      int KY = KXLT[IJ] + 1; /* 1-based row */
      int KK = KY + 2 * IC - 3;
      int KKM = KK < 1 ? 1 : (KK > ngy ? ngy : KK);
      DP[IJ * 2 + (IC - 1)] = COSPH[KKM - 1] * COSPHM1_EXT[IJ];
    }
}

/* ctuw.F90:110-275 (space weights), :407-484 (direction weights, IREFRA=0), :536-608 (checks + SUMWN) */
int ora_ctuw(int n, int nland, real DELPRO, int MSTART, int MEND, const int *KXLT, const real *ZDELLO, real XDELLA,
             const real *COSPH, const real *SINPH, const int *KLON, const int *KLAT, const real *WLAT, const real *WCOR,
             const real *WLATM1, const real *WCORM1, const real *DP, const real *CGROUP_EXT, const real *COSPHM1_EXT,
             real *SUMWN, real *WLONN, real *WLATN, real *WCORN, real *WKPMN, int *LCFLFAIL) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  (void)nland;
  real CMTODEG = C_(360.0) / S.CIRC;
  for (int IJ = 0; IJ < n; IJ++) LCFLFAIL[IJ] = 0;
  /* (the reference loops M, K, IJ with IJ innermost over arrays that have IJ fastest; here IJ is the slowest index of the arrays:
   *  IJ outermost -- the elements are independent of each other, the interchange is exact) */
#pragma omp parallel for schedule(static)
  for (int IJ = 0; IJ < n; IJ++) {
    for (int K = 1; K <= NANG; K++) {
      const int *jx = S.JXO[K - 1], *jy = S.JYO[K - 1], *kc = S.KCR[K - 1];
      for (int M = MSTART; M <= MEND; M++) {
        real CGX[3], CGY[3], ADXP[3], ADYP[3], DXUP[3], DXDW[3], DYUP[3], DYDW[3], WEIGHT[5];
        real CG0 = CGROUP_EXT[(size_t)IJ * NFRE + (M - 1)];
        int KY = KXLT[IJ];
        for (int IC = 1; IC <= 2; IC++) {
          real CGL = CGROUP_EXT[(size_t)KLON[IJ * 2 + (IC - 1)] * NFRE + (M - 1)];
          CGX[IC] = C_(0.5) * (CG0 + CGL) * S.SINTH[K - 1] * COSPHM1_EXT[IJ];
          real CGYP = WLAT[IJ * 2 + (IC - 1)] * CGROUP_EXT[(size_t)KLAT[(IJ * 2 + (IC - 1)) * 2 + 0] * NFRE + (M - 1)] +
                      (C_(1.0) - WLAT[IJ * 2 + (IC - 1)]) * CGROUP_EXT[(size_t)KLAT[(IJ * 2 + (IC - 1)) * 2 + 1] * NFRE + (M - 1)];
          CGY[IC] = C_(0.5) * (CG0 + DP[IJ * 2 + (IC - 1)] * CGYP) * S.COSTH[K - 1];
          real UREL = CGX[IC], VREL = CGY[IC];
          int ISSU = 1, ISSV = 1;
          real DXP = -DELPRO * UREL * CMTODEG;
          real DYP = -DELPRO * VREL * CMTODEG;
          ADXP[IC] = FABS(DXP); ADYP[IC] = FABS(DYP);
          DXUP[IC] = ADXP[IC] * ISSU; DXDW[IC] = ADXP[IC] * (1 - ISSU);
          DYUP[IC] = ADYP[IC] * ISSV; DYDW[IC] = ADYP[IC] * (1 - ISSV);
        }
        real ZD = ZDELLO[KY];
        real DXX = ZD - DXUP[jx[1]] - DXDW[jx[0]];
        real DYY = XDELLA - DYUP[jy[1]] - DYDW[jy[0]];
        real GRIDAREAM1 = C_(1.0) / (ZD * XDELLA);
        size_t b = ((size_t)IJ * NANG + (K - 1)) * NR + (M - 1);
        WEIGHT[jy[0]] = DXX * DYUP[jy[0]] * GRIDAREAM1;
        WEIGHT[jy[1]] = DXX * DYDW[jy[1]] * GRIDAREAM1;
        WLATN[(b * 2 + 0) * 2 + 0] = WLAT[IJ * 2 + 0] * WEIGHT[1];
        WLATN[(b * 2 + 0) * 2 + 1] = WLATM1[IJ * 2 + 0] * WEIGHT[1];
        WLATN[(b * 2 + 1) * 2 + 0] = WLAT[IJ * 2 + 1] * WEIGHT[2];
        WLATN[(b * 2 + 1) * 2 + 1] = WLATM1[IJ * 2 + 1] * WEIGHT[2];
        WLONN[b * 2 + (jx[0] - 1)] = DYY * DXUP[jx[0]] * GRIDAREAM1;
        WLONN[b * 2 + (jx[1] - 1)] = DYY * DXDW[jx[1]] * GRIDAREAM1;
        WEIGHT[1] = DXUP[jx[0]] * DYUP[jy[0]] * GRIDAREAM1;
        WEIGHT[2] = DXDW[jx[1]] * DYUP[jy[0]] * GRIDAREAM1;
        WEIGHT[3] = DXUP[jx[0]] * DYDW[jy[1]] * GRIDAREAM1;
        WEIGHT[4] = DXDW[jx[1]] * DYDW[jy[1]] * GRIDAREAM1;
        for (int ICR = 1; ICR <= 4; ICR++) {
          WCORN[(b * 4 + (ICR - 1)) * 2 + 0] = WCOR[IJ * 4 + (kc[ICR - 1] - 1)] * WEIGHT[ICR];
          WCORN[(b * 4 + (ICR - 1)) * 2 + 1] = WCORM1[IJ * 4 + (kc[ICR - 1] - 1)] * WEIGHT[ICR];
        }
        SUMWN[b] = (ZD * (DYDW[jy[0]] + DYUP[jy[1]]) + XDELLA * (DXUP[jx[1]] + DXDW[jx[0]]) -
                    (DXDW[jx[0]] + DXUP[jx[1]]) * (DYDW[jy[0]] + DYUP[jy[1]])) * GRIDAREAM1;
        if (ADXP[1] > ZD || ADYP[1] > XDELLA || ADXP[2] > ZD || ADYP[2] > XDELLA) LCFLFAIL[IJ] = 1;
      }
    }
  }
  /* refraction terms, ctuw.F90:407-484 */
  real DELTH0 = (real)(0.25 * DELPRO) / S.DELTH; /* 0.25 is a default-kind literal, ctuw.F90:407 */
#pragma omp parallel for schedule(static)
  for (int IJ = 0; IJ < n; IJ++) {
    for (int K = 1; K <= NANG; K++) {
      int KP1 = K + 1; if (KP1 > NANG) KP1 = 1;
      int KM1 = K - 1; if (KM1 < 1) KM1 = NANG;
      real SP = DELTH0 * (S.SINTH[K - 1] + S.SINTH[KP1 - 1]) / S.R;
      real SM = DELTH0 * (S.SINTH[K - 1] + S.SINTH[KM1 - 1]) / S.R;
      int JH = KXLT[IJ];
      real TANPH = SINPH[JH] / COSPH[JH];
      real DRGP = TANPH * SP, DRGM = TANPH * SM;
      for (int M = MSTART; M <= MEND; M++) {
        size_t b = ((size_t)IJ * NANG + (K - 1)) * NR + (M - 1);
        real CG0 = CGROUP_EXT[(size_t)IJ * NFRE + (M - 1)];
        real DTHP = DRGP * CG0 + C_(0.0);
        real DTHM = DRGM * CG0 + C_(0.0);
        WKPMN[b * 3 + 1] = (DTHP + FABS(DTHP)) + (FABS(DTHM) - DTHM);
        WKPMN[b * 3 + 2] = -DTHP + FABS(DTHP);
        WKPMN[b * 3 + 0] = DTHM + FABS(DTHM);
      }
    }
  }
  /* checks + SUMWN accumulation, ctuw.F90:536-687 */
#pragma omp parallel for schedule(static)
  for (int IJ = 0; IJ < n; IJ++)
    for (int K = 1; K <= NANG; K++)
      for (int M = MSTART; M <= MEND; M++) {
        size_t b = ((size_t)IJ * NANG + (K - 1)) * NR + (M - 1);
        for (int i = 0; i < 4; i++) if (WLATN[b * 4 + i] > C_(1.0) || WLATN[b * 4 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        for (int i = 0; i < 2; i++) if (WLONN[b * 2 + i] > C_(1.0) || WLONN[b * 2 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        for (int i = 0; i < 8; i++) if (WCORN[b * 8 + i] > C_(1.0) || WCORN[b * 8 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        for (int i = 0; i < 3; i++) if (WKPMN[b * 3 + i] > C_(1.0) || WKPMN[b * 3 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        SUMWN[b] = SUMWN[b] + WKPMN[b * 3 + 1];
        if (SUMWN[b] > C_(1.0) || SUMWN[b] < C_(0.0)) LCFLFAIL[IJ] = 1;
      }
  int nfail = 0;
  for (int IJ = 0; IJ < n; IJ++) nfail += LCFLFAIL[IJ];
  if (!nfail) apply_obstructions(n, MSTART, MEND, WLONN, WLATN, WCORN); /* ctuw.F90:691-733: only reached without a failure */
  return nfail;
}

/* propags2.F90:101-120.  F1,F3: [(npts)][NANG][NFRE] with the land slot inside F1; ND3S..ND3E 1-based inclusive */
void ora_propags2(int KIJS, int KIJL, const real *F1, real *F3, const int *KLON, const int *KLAT, const int *KCOR,
                  const real *SUMWN, const real *WLONN, const real *WLATN, const real *WCORN, const real *WKPMN, int ND3S,
                  int ND3E) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
#define FF1(ij, k, m) F1[((size_t)(ij) * NANG + (k)) * NFRE + (m)]
#pragma omp parallel for schedule(static)
  for (int IJ = KIJS; IJ < KIJL; IJ++) {
    for (int K = 0; K < NANG; K++) {
      int jx = S.JXO[K][0] - 1, jy = S.JYO[K][0] - 1, kc = S.KCR[K][0] - 1;
      int km = S.KPM[K][0] - 1, kp = S.KPM[K][2] - 1;
      int ilon = KLON[IJ * 2 + jx];
      int ilat1 = KLAT[(IJ * 2 + jy) * 2 + 0], ilat2 = KLAT[(IJ * 2 + jy) * 2 + 1];
      int icor1 = KCOR[(IJ * 4 + kc) * 2 + 0], icor2 = KCOR[(IJ * 4 + kc) * 2 + 1];
      for (int M = ND3S - 1; M < ND3E; M++) {
        size_t b = ((size_t)IJ * NANG + K) * NR + M;
        F3[((size_t)IJ * NANG + K) * NFRE + M] =
            (C_(1.0) - SUMWN[b]) * FF1(IJ, K, M) + WLONN[b * 2 + jx] * FF1(ilon, K, M) +
            WLATN[(b * 2 + jy) * 2 + 0] * FF1(ilat1, K, M) + WLATN[(b * 2 + jy) * 2 + 1] * FF1(ilat2, K, M) +
            WCORN[(b * 4 + 0) * 2 + 0] * FF1(icor1, K, M) + WCORN[(b * 4 + 0) * 2 + 1] * FF1(icor2, K, M) +
            WKPMN[b * 3 + 0] * FF1(IJ, km, M) + WKPMN[b * 3 + 2] * FF1(IJ, kp, M);
      }
    }
  }
#undef FF1
}

/* propag_wam.F90:247-313, IPROPAGS = 2 without refraction: the PROPAG_WAM advection sequence on ONE domain (MPEXCHNG returns at once
 * when NPROC <= 1, mpexchng.F90:106-109).  The weights are the ones CTUWUPDT leaves behind (ctuwupdt.F90:220-256): with IFRELFMAX > 0
 * DELPRO_LF for M <= IFRELFMAX and IDELPRO above, i.e. two ora_ctuw calls into the same arrays.
 *   :247-254  PROPAGS2(FL1_EXT, FL3_EXT, ..., ND3SF1=1, ND3EF1=NFRE_RED, ND3S=1, ND3E=NFRE_RED)
 *   :257-259  NSTEP_LF = NINT(REAL(IDELPRO)/DELPRO_LF); ISUBST = 2
 *   :261-264  ND3SF1 = 1, ND3EF1 = IFRELFMAX+1, ND3S = 1, ND3E = IFRELFMAX
 *   :266-311  DO WHILE (ISUBST <= NSTEP_LF): FL1_EXT(IJ,K,ND3S:ND3E) = FL3_EXT(IJ,K,ND3S:ND3E) for the owned points; MPEXCHNG; PROPAGS2
 *   :373-386  FL1(IJ,K,M,ICHNK) = FL3_EXT(..) for M <= NFRE_RED (block -> chunk; M > NFRE_RED is not touched).
 * FL1_EXT: [(n+1)][NANG][NFRE] in/out, row n = the land slot (zero, propag_wam.F90:146); FL3_EXT: scratch of the same shape.
 * Returns NSTEP_LF (1 when the fast waves are not sub-stepped). */
int ora_propag_wam(int n, real *FL1_EXT, real *FL3_EXT, const int *KLON, const int *KLAT, const int *KCOR, const real *SUMWN,
                   const real *WLONN, const real *WLATN, const real *WCORN, const real *WKPMN, int IFRELFMAX, int IDELPRO,
                   real DELPRO_LF) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  int NSTEP_LF = 1;
  for (size_t i = 0; i < (size_t)NANG * NFRE; i++) FL1_EXT[(size_t)n * NANG * NFRE + i] = C_(0.0);
  ora_propags2(0, n, FL1_EXT, FL3_EXT, KLON, KLAT, KCOR, SUMWN, WLONN, WLATN, WCORN, WKPMN, 1, NR);
  if (IFRELFMAX > 0 && IFRELFMAX < NR) {
    NSTEP_LF = NINT((real)IDELPRO / DELPRO_LF);
    int ISUBST = 2;
    const int ND3S = 1, ND3E = IFRELFMAX;
    while (ISUBST <= NSTEP_LF) {
      for (int IJ = 0; IJ < n; IJ++)
        for (int K = 0; K < NANG; K++)
          for (int M = ND3S - 1; M < ND3E; M++)
            FL1_EXT[((size_t)IJ * NANG + K) * NFRE + M] = FL3_EXT[((size_t)IJ * NANG + K) * NFRE + M];
      ora_propags2(0, n, FL1_EXT, FL3_EXT, KLON, KLAT, KCOR, SUMWN, WLONN, WLATN, WCORN, WKPMN, ND3S, ND3E);
      ISUBST++;
    }
  }
  for (int IJ = 0; IJ < n; IJ++)
    for (int K = 0; K < NANG; K++)
      for (int M = 0; M < NR; M++) FL1_EXT[((size_t)IJ * NANG + K) * NFRE + M] = FL3_EXT[((size_t)IJ * NANG + K) * NFRE + M];
  return NSTEP_LF;
}

/* Timing variant of PROPAGS2 for bench.py's cpu_baseline: the reference keeps its weight arrays with IJ fastest, so the eight
 * weights the stencil reads are eight contiguous streams for its compiler; the [ij][k][m][..] layout of this restatement makes them
 * strided and drags the ten unused weights per element through the caches.  ora_pack_w8 packs the eight once as W8[ij][8][K][M]
 * (order: 1-SUMWN, WLONN(JXO), WLATN(JYO,1:2), WCORN(KCR,1:2), WKPMN(-1), WKPMN(+1)); ora_propags2_w8 is the same sum, same order of
 * the terms, with unit-stride vectorisable loops over M.  Same results as ora_propags2 bit for bit (tests/test_host.py). */
void ora_pack_w8(int n, const real *SUMWN, const real *WLONN, const real *WLATN, const real *WCORN, const real *WKPMN, real *W8) {
  const int NANG = S.NANG, NR = S.NFRE_RED;
#pragma omp parallel for schedule(static)
  for (int IJ = 0; IJ < n; IJ++)
    for (int K = 0; K < NANG; K++) {
      const int jx = S.JXO[K][0] - 1, jy = S.JYO[K][0] - 1, kc = S.KCR[K][0] - 1;
      for (int M = 0; M < NR; M++) {
        const size_t b = ((size_t)IJ * NANG + K) * NR + M;
        real *w = W8 + (((size_t)IJ * 8) * NANG + K) * NR + M;
        const size_t pl = (size_t)NANG * NR;
        w[0] = C_(1.0) - SUMWN[b];
        w[pl] = WLONN[b * 2 + jx];
        w[2 * pl] = WLATN[(b * 2 + jy) * 2 + 0]; w[3 * pl] = WLATN[(b * 2 + jy) * 2 + 1];
        w[4 * pl] = WCORN[(b * 4 + 0) * 2 + 0]; w[5 * pl] = WCORN[(b * 4 + 0) * 2 + 1];   /* WCORN(..,1,1:2): the corner KCR(K,1) */
        w[6 * pl] = WKPMN[b * 3 + 0]; w[7 * pl] = WKPMN[b * 3 + 2];
      }
    }
}
void ora_propags2_w8(int KIJS, int KIJL, const real *F1, real *F3, const int *KLON, const int *KLAT, const int *KCOR, const real *W8,
                     int ND3S, int ND3E) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  const size_t pl = (size_t)NANG * NR;
#pragma omp parallel for schedule(static)
  for (int IJ = KIJS; IJ < KIJL; IJ++) {
    for (int K = 0; K < NANG; K++) {
      const int jx = S.JXO[K][0] - 1, jy = S.JYO[K][0] - 1, kc = S.KCR[K][0] - 1;
      const int km = S.KPM[K][0] - 1, kp = S.KPM[K][2] - 1;
      const real *fo = F1 + ((size_t)IJ * NANG + K) * NFRE, *flon = F1 + ((size_t)KLON[IJ * 2 + jx] * NANG + K) * NFRE;
      const real *fl1 = F1 + ((size_t)KLAT[(IJ * 2 + jy) * 2 + 0] * NANG + K) * NFRE, *fl2 = F1 + ((size_t)KLAT[(IJ * 2 + jy) * 2 + 1] * NANG + K) * NFRE;
      const real *fc1 = F1 + ((size_t)KCOR[(IJ * 4 + kc) * 2 + 0] * NANG + K) * NFRE, *fc2 = F1 + ((size_t)KCOR[(IJ * 4 + kc) * 2 + 1] * NANG + K) * NFRE;
      const real *fkm = F1 + ((size_t)IJ * NANG + km) * NFRE, *fkp = F1 + ((size_t)IJ * NANG + kp) * NFRE;
      const real *w = W8 + (((size_t)IJ * 8) * NANG + K) * NR;
      real *o = F3 + ((size_t)IJ * NANG + K) * NFRE;
#pragma omp simd
      for (int M = ND3S - 1; M < ND3E; M++)
        o[M] = w[M] * fo[M] + w[pl + M] * flon[M] + w[2 * pl + M] * fl1[M] + w[3 * pl + M] * fl2[M] + w[4 * pl + M] * fc1[M] +
               w[5 * pl + M] * fc2[M] + w[6 * pl + M] * fkm[M] + w[7 * pl + M] * fkp[M];
    }
  }
}

/* ======================================================================================================================
 * IREFRA = 1 (depth refraction), 2 (current refraction), 3 (depth + current refraction)
 * ====================================================================================================================== */

/* gradi.F90:113-232 (depth and current gradients; DELPHI, DELLAM from readmdlconf.F90:136-154) followed by
 * propdot.F90:108-196 (THDD, THDC, SDOT).  *_EXT arrays: [npts+1] resp. [npts+1][NFRE], land slot = index nland with
 * the values proenvhalo.F90:99-106 assigns (DEPTH = BATHYMAX, U = V = COSPHM1 = 0).
 * Outputs THDC, THDD: [n][NANG]; SDOT: [n][NANG][NFRE_RED]. */
void ora_propdot(int n, int nland, int IREFRA, const int *KXLT, const int *KLON, const int *KLAT, const real *WLAT,
                 const real *ZDELLO, real XDELLA, const real *COSPH, const real *COSPHM1_EXT, const real *DEPTH_EXT,
                 const real *U_EXT, const real *V_EXT, const real *WAVNUM_EXT, const real *CGROUP_EXT,
                 const real *OMOSNH2KD_EXT, real *THDC, real *THDD, real *SDOT) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  const real CURRENT_GRADIENT_MAX = C_(0.00001); /* yowcurr.F90:19 */
  const real DELPHI = XDELLA * S.CIRC / C_(360.0);
  const real ONEO2DELPHI = C_(0.5) / DELPHI;
  for (int IJ = 0; IJ < n; IJ++) {
    real DDPHI = C_(0.0), DDLAM = C_(0.0), DUPHI = C_(0.0), DULAM = C_(0.0), DVPHI = C_(0.0), DVLAM = C_(0.0);
    const int KX = KXLT[IJ];
    const real DELLAM = ZDELLO[KX] * S.CIRC / C_(360.0);
    if (IREFRA == 1 || IREFRA == 3) { /* gradi.F90:120-158 */
      int IPP = KLAT[(IJ * 2 + 1) * 2 + 0], IPM = KLAT[(IJ * 2 + 0) * 2 + 0];
      int IPP2 = KLAT[(IJ * 2 + 1) * 2 + 1], IPM2 = KLAT[(IJ * 2 + 0) * 2 + 1];
      if (IPP != nland && IPM != nland && IPP2 != nland && IPM2 != nland) {
        real DPTP = WLAT[IJ * 2 + 1] * DEPTH_EXT[IPP] + (C_(1.0) - WLAT[IJ * 2 + 1]) * DEPTH_EXT[IPP2];
        real DPTM = WLAT[IJ * 2 + 0] * DEPTH_EXT[IPM] + (C_(1.0) - WLAT[IJ * 2 + 0]) * DEPTH_EXT[IPM2];
        DDPHI = (DPTP - DPTM) * ONEO2DELPHI;
      } else if (IPP != nland && IPM != nland) {
        DDPHI = (DEPTH_EXT[IPP] - DEPTH_EXT[IPM]) * ONEO2DELPHI;
      } else if (IPP2 != nland && IPM2 != nland) {
        DDPHI = (DEPTH_EXT[IPP2] - DEPTH_EXT[IPM2]) * ONEO2DELPHI;
      }
      int ILP = KLON[IJ * 2 + 1], ILM = KLON[IJ * 2 + 0];
      if (ILP != nland && ILM != nland) DDLAM = (DEPTH_EXT[ILP] - DEPTH_EXT[ILM]) / (C_(2.) * DELLAM);
    }
    if (IREFRA == 2 || IREFRA == 3) { /* gradi.F90:167-228 */
      int IPP = KLAT[(IJ * 2 + 1) * 2 + 0], IPM = KLAT[(IJ * 2 + 0) * 2 + 0];
      int IPP2 = KLAT[(IJ * 2 + 1) * 2 + 1], IPM2 = KLAT[(IJ * 2 + 0) * 2 + 1];
      /* exact 0 means that the current field was not defined: no gradient is extrapolated */
      if (U_EXT[IPP] == C_(0.0) && V_EXT[IPP] == C_(0.0)) IPP = nland;
      if (U_EXT[IPM] == C_(0.0) && V_EXT[IPM] == C_(0.0)) IPM = nland;
      if (U_EXT[IPP2] == C_(0.0) && V_EXT[IPP2] == C_(0.0)) IPP2 = nland;
      if (U_EXT[IPM2] == C_(0.0) && V_EXT[IPM2] == C_(0.0)) IPM2 = nland;
      if (IPP != nland && IPM != nland && IPP2 != nland && IPM2 != nland) {
        real UP = WLAT[IJ * 2 + 1] * U_EXT[IPP] + (C_(1.0) - WLAT[IJ * 2 + 1]) * U_EXT[IPP2];
        real VP = WLAT[IJ * 2 + 1] * V_EXT[IPP] + (C_(1.0) - WLAT[IJ * 2 + 1]) * V_EXT[IPP2];
        real UM = WLAT[IJ * 2 + 0] * U_EXT[IPM] + (C_(1.0) - WLAT[IJ * 2 + 0]) * U_EXT[IPM2];
        real VM = WLAT[IJ * 2 + 0] * V_EXT[IPM] + (C_(1.0) - WLAT[IJ * 2 + 0]) * V_EXT[IPM2];
        DUPHI = (UP - UM) * ONEO2DELPHI;
        DVPHI = (VP - VM) * ONEO2DELPHI;
      } else if (IPP != nland && IPM != nland) {
        DUPHI = (U_EXT[IPP] - U_EXT[IPM]) * ONEO2DELPHI;
        DVPHI = (V_EXT[IPP] - V_EXT[IPM]) * ONEO2DELPHI;
      }
      int ILP = KLON[IJ * 2 + 1], ILM = KLON[IJ * 2 + 0];
      if (U_EXT[ILP] == C_(0.0) && V_EXT[ILP] == C_(0.0)) ILP = nland;
      if (U_EXT[ILM] == C_(0.0) && V_EXT[ILM] == C_(0.0)) ILM = nland;
      if (ILP != nland && ILM != nland) {
        DULAM = (U_EXT[ILP] - U_EXT[ILM]) / (C_(2.0) * DELLAM);
        DVLAM = (V_EXT[ILP] - V_EXT[ILM]) / (C_(2.0) * DELLAM);
      }
      const real CGMAX = CURRENT_GRADIENT_MAX * COSPH[KX];
      DUPHI = COPYSIGN(RMIN(FABS(DUPHI), CGMAX), DUPHI);
      DVPHI = COPYSIGN(RMIN(FABS(DVPHI), CGMAX), DVPHI);
      DULAM = COPYSIGN(RMIN(FABS(DULAM), CGMAX), DULAM);
      DVLAM = COPYSIGN(RMIN(FABS(DVLAM), CGMAX), DVLAM);
    }
    /* propdot.F90:108-196 */
    const real DCO = COSPHM1_EXT[IJ];
    real OMDD = C_(0.0);
    if (IREFRA == 3) OMDD = V_EXT[IJ] * DDPHI + U_EXT[IJ] * DDLAM * DCO;
    for (int K = 0; K < NANG; K++) {
      const real SD = S.SINTH[K], CD = S.COSTH[K];
      THDD[IJ * NANG + K] = (IREFRA == 1 || IREFRA == 3) ? SD * DDPHI - CD * DDLAM * DCO : C_(0.0);
      THDC[IJ * NANG + K] = C_(0.0);
      if (IREFRA == 2 || IREFRA == 3) {
        const real SS = SD * SD, SC = SD * CD, CC = CD * CD;
        const real S0 = -SC * DUPHI - CC * DVPHI - (SS * DULAM + SC * DVLAM) * DCO;
        THDC[IJ * NANG + K] = SS * DUPHI + SC * DVPHI - (SC * DULAM + CC * DVLAM) * DCO;
        for (int M = 0; M < NR; M++)
          SDOT[((size_t)IJ * NANG + K) * NR + M] =
              (S0 * CGROUP_EXT[(size_t)IJ * NFRE + M] + OMDD * OMOSNH2KD_EXT[(size_t)IJ * NFRE + M]) * WAVNUM_EXT[(size_t)IJ * NFRE + M];
      } else {
        for (int M = 0; M < NR; M++) SDOT[((size_t)IJ * NANG + K) * NR + M] = C_(0.0);
      }
    }
  }
}

/* ctuw.F90:1-757 for one ICALL: all 18 space weights (currents can make the downwind terms non-zero), the direction
 * weights with depth (IREFRA=1) or current (IREFRA=2,3) refraction, the frequency-shift weights (IREFRA=2,3), the
 * range checks and SUMWN.  WLONN[b][2], WLATN[b][2][2], WCORN[b][4][2], WKPMN[b][3], WMPMN[b][3] with
 * b = (ij*NANG + k)*NFRE_RED + m. */
static int isamesign(real A, real B) { return COPYSIGN(C_(1.0), A) == COPYSIGN(C_(1.0), B); }
static void ctuw_gen_call(int n, int IREFRA, real DELPRO, int MSTART, int MEND, const int *KXLT, const real *ZDELLO, real XDELLA,
                          const real *COSPH, const real *SINPH, const int *KLON, const int *KLAT, const real *WLAT,
                          const real *WCOR, const real *WLATM1, const real *WCORM1, const real *DP, const real *CGROUP_EXT,
                          const real *OMOSNH2KD_EXT, const real *COSPHM1_EXT, const real *U_EXT, const real *V_EXT,
                          const real *THDC, const real *THDD, const real *SDOT, const real *CURMASK, real *SUMWN, real *WLONN,
                          real *WLATN, real *WCORN, real *WKPMN, real *WMPMN, int *LCFLFAIL) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  const int cur = (IREFRA == 2 || IREFRA == 3);
  real CMTODEG = C_(360.0) / S.CIRC;
  for (int IJ = 0; IJ < n; IJ++) LCFLFAIL[IJ] = 0;
  for (int M = MSTART; M <= MEND; M++) {
    for (int K = 1; K <= NANG; K++) {
      const int *jx = S.JXO[K - 1], *jy = S.JYO[K - 1], *kc = S.KCR[K - 1];
      for (int IJ = 0; IJ < n; IJ++) {
        real CGX[3], CGY[3], ADXP[3], ADYP[3], DXUP[3], DXDW[3], DYUP[3], DYDW[3], WEIGHT[5];
        real CG0 = CGROUP_EXT[(size_t)IJ * NFRE + (M - 1)];
        int KY = KXLT[IJ];
        for (int IC = 1; IC <= 2; IC++) {
          real CGL = CGROUP_EXT[(size_t)KLON[IJ * 2 + (IC - 1)] * NFRE + (M - 1)];
          CGX[IC] = C_(0.5) * (CG0 + CGL) * S.SINTH[K - 1] * COSPHM1_EXT[IJ];
          real CGYP = WLAT[IJ * 2 + (IC - 1)] * CGROUP_EXT[(size_t)KLAT[(IJ * 2 + (IC - 1)) * 2 + 0] * NFRE + (M - 1)] +
                      (C_(1.0) - WLAT[IJ * 2 + (IC - 1)]) * CGROUP_EXT[(size_t)KLAT[(IJ * 2 + (IC - 1)) * 2 + 1] * NFRE + (M - 1)];
          CGY[IC] = C_(0.5) * (CG0 + DP[IJ * 2 + (IC - 1)] * CGYP) * S.COSTH[K - 1];
          real UREL, VREL;
          int ISSU, ISSV;
          if (cur) {
            real UU = U_EXT[IJ] * COSPHM1_EXT[IJ];
            UREL = CGX[IC] + UU;
            ISSU = isamesign(UREL, CGX[IC]);
            real VV = V_EXT[IJ] * C_(0.5) * (C_(1.0) + DP[IJ * 2 + (IC - 1)]);
            VREL = CGY[IC] + VV;
            ISSV = isamesign(VREL, CGY[IC]);
          } else {
            UREL = CGX[IC]; ISSU = 1; VREL = CGY[IC]; ISSV = 1;
          }
          real DXP = -DELPRO * UREL * CMTODEG;
          real DYP = -DELPRO * VREL * CMTODEG;
          ADXP[IC] = FABS(DXP); ADYP[IC] = FABS(DYP);
          DXUP[IC] = ADXP[IC] * ISSU; DXDW[IC] = ADXP[IC] * (1 - ISSU);
          DYUP[IC] = ADYP[IC] * ISSV; DYDW[IC] = ADYP[IC] * (1 - ISSV);
        }
        real ZD = ZDELLO[KY];
        real DXX = ZD - DXUP[jx[1]] - DXDW[jx[0]];
        real DYY = XDELLA - DYUP[jy[1]] - DYDW[jy[0]];
        real GRIDAREAM1 = C_(1.0) / (ZD * XDELLA);
        size_t b = ((size_t)IJ * NANG + (K - 1)) * NR + (M - 1);
        WEIGHT[jy[0]] = DXX * DYUP[jy[0]] * GRIDAREAM1;
        WEIGHT[jy[1]] = DXX * DYDW[jy[1]] * GRIDAREAM1;
        WLATN[(b * 2 + 0) * 2 + 0] = WLAT[IJ * 2 + 0] * WEIGHT[1];
        WLATN[(b * 2 + 0) * 2 + 1] = WLATM1[IJ * 2 + 0] * WEIGHT[1];
        WLATN[(b * 2 + 1) * 2 + 0] = WLAT[IJ * 2 + 1] * WEIGHT[2];
        WLATN[(b * 2 + 1) * 2 + 1] = WLATM1[IJ * 2 + 1] * WEIGHT[2];
        WLONN[b * 2 + (jx[0] - 1)] = DYY * DXUP[jx[0]] * GRIDAREAM1;
        WLONN[b * 2 + (jx[1] - 1)] = DYY * DXDW[jx[1]] * GRIDAREAM1;
        WEIGHT[1] = DXUP[jx[0]] * DYUP[jy[0]] * GRIDAREAM1;
        WEIGHT[2] = DXDW[jx[1]] * DYUP[jy[0]] * GRIDAREAM1;
        WEIGHT[3] = DXUP[jx[0]] * DYDW[jy[1]] * GRIDAREAM1;
        WEIGHT[4] = DXDW[jx[1]] * DYDW[jy[1]] * GRIDAREAM1;
        for (int ICR = 1; ICR <= 4; ICR++) {
          WCORN[(b * 4 + (ICR - 1)) * 2 + 0] = WCOR[IJ * 4 + (kc[ICR - 1] - 1)] * WEIGHT[ICR];
          WCORN[(b * 4 + (ICR - 1)) * 2 + 1] = WCORM1[IJ * 4 + (kc[ICR - 1] - 1)] * WEIGHT[ICR];
        }
        SUMWN[b] = (ZD * (DYDW[jy[0]] + DYUP[jy[1]]) + XDELLA * (DXUP[jx[1]] + DXDW[jx[0]]) -
                    (DXDW[jx[0]] + DXUP[jx[1]]) * (DYDW[jy[0]] + DYUP[jy[1]])) * GRIDAREAM1;
        if (ADXP[1] > ZD || ADYP[1] > XDELLA || ADXP[2] > ZD || ADYP[2] > XDELLA) LCFLFAIL[IJ] = 1;
      }
    }
  }
  /* refraction terms, ctuw.F90:403-527 */
  real DELTH0 = (real)(0.25 * DELPRO) / S.DELTH; /* 0.25 is a default-kind literal, ctuw.F90:407 */
  for (int K = 1; K <= NANG; K++) {
    int KP1 = K + 1; if (KP1 > NANG) KP1 = 1;
    int KM1 = K - 1; if (KM1 < 1) KM1 = NANG;
    real SP = DELTH0 * (S.SINTH[K - 1] + S.SINTH[KP1 - 1]) / S.R;
    real SM = DELTH0 * (S.SINTH[K - 1] + S.SINTH[KM1 - 1]) / S.R;
    for (int IJ = 0; IJ < n; IJ++) {
      int JH = KXLT[IJ];
      real TANPH = SINPH[JH] / COSPH[JH];
      real DRGP = TANPH * SP, DRGM = TANPH * SM;
      real DRDP = C_(0.0), DRDM = C_(0.0), DRCP = C_(0.0), DRCM = C_(0.0);
      if (IREFRA == 1) {
        DRDP = (THDD[IJ * NANG + (K - 1)] + THDD[IJ * NANG + (KP1 - 1)]) * DELTH0;
        DRDM = (THDD[IJ * NANG + (K - 1)] + THDD[IJ * NANG + (KM1 - 1)]) * DELTH0;
      }
      if (cur) {
        DRCP = CURMASK[IJ] * (THDC[IJ * NANG + (K - 1)] + THDC[IJ * NANG + (KP1 - 1)]) * DELTH0;
        DRCM = CURMASK[IJ] * (THDC[IJ * NANG + (K - 1)] + THDC[IJ * NANG + (KM1 - 1)]) * DELTH0;
      }
      for (int M = MSTART; M <= MEND; M++) {
        size_t b = ((size_t)IJ * NANG + (K - 1)) * NR + (M - 1);
        real CG0 = CGROUP_EXT[(size_t)IJ * NFRE + (M - 1)];
        real DTHP, DTHM;
        if (IREFRA == 0) {
          DTHP = DRGP * CG0 + DRCP;
          DTHM = DRGM * CG0 + DRCM;
        } else {
          real OM = OMOSNH2KD_EXT[(size_t)IJ * NFRE + (M - 1)];
          DTHP = DRGP * CG0 + OM * DRDP + DRCP;
          DTHM = DRGM * CG0 + OM * DRDM + DRCM;
        }
        WKPMN[b * 3 + 1] = (DTHP + FABS(DTHP)) + (FABS(DTHM) - DTHM);
        WKPMN[b * 3 + 2] = -DTHP + FABS(DTHP);
        WKPMN[b * 3 + 0] = DTHM + FABS(DTHM);
      }
    }
    if (cur) { /* frequency shifting due to currents, ctuw.F90:506-526 */
      real DELFR0 = C_(0.25) * DELPRO / ((S.FRATIO - 1) * S.ZPI);
      for (int M = MSTART; M <= MEND; M++) {
        int MP1 = M + 1 < NR ? M + 1 : NR;
        int MM1 = M - 1 > 1 ? M - 1 : 1;
        real DFP = DELFR0 / S.FR[M - 1];
        real DFM = DELFR0 / S.FR[MM1 - 1];
        for (int IJ = 0; IJ < n; IJ++) {
          size_t bk = ((size_t)IJ * NANG + (K - 1)) * NR;
          real DTHP = CURMASK[IJ] * (SDOT[bk + (M - 1)] + SDOT[bk + (MP1 - 1)]) * DFP;
          real DTHM = CURMASK[IJ] * (SDOT[bk + (M - 1)] + SDOT[bk + (MM1 - 1)]) * DFM;
          size_t b = bk + (M - 1);
          WMPMN[b * 3 + 1] = (DTHP + FABS(DTHP)) + (FABS(DTHM) - DTHM);
          WMPMN[b * 3 + 2] = (-DTHP + FABS(DTHP)) / S.FRATIO;
          WMPMN[b * 3 + 0] = (DTHM + FABS(DTHM)) * S.FRATIO;
        }
      }
    }
  }
  /* checks + SUMWN accumulation, ctuw.F90:536-687 */
  for (int K = 1; K <= NANG; K++)
    for (int M = MSTART; M <= MEND; M++)
      for (int IJ = 0; IJ < n; IJ++) {
        size_t b = ((size_t)IJ * NANG + (K - 1)) * NR + (M - 1);
        for (int i = 0; i < 4; i++) if (WLATN[b * 4 + i] > C_(1.0) || WLATN[b * 4 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        for (int i = 0; i < 2; i++) if (WLONN[b * 2 + i] > C_(1.0) || WLONN[b * 2 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        for (int i = 0; i < 8; i++) if (WCORN[b * 8 + i] > C_(1.0) || WCORN[b * 8 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        for (int i = 0; i < 3; i++) if (WKPMN[b * 3 + i] > C_(1.0) || WKPMN[b * 3 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
        SUMWN[b] = SUMWN[b] + WKPMN[b * 3 + 1];
        if (cur) {
          for (int i = 0; i < 3; i++) if (WMPMN[b * 3 + i] > C_(1.0) || WMPMN[b * 3 + i] < C_(0.0)) LCFLFAIL[IJ] = 1;
          SUMWN[b] = SUMWN[b] + WMPMN[b * 3 + 1];
        }
        if (SUMWN[b] > C_(1.0) || SUMWN[b] < C_(0.0)) LCFLFAIL[IJ] = 1;
      }
}

/* ctuwdrv.F90:93-118: CTUW, and when the CFL criterion fails with currents and LLCFLCUROFF a second call in which the
 * current refraction / frequency shift terms of the failing points are switched off (CURMASK, ctuw.F90:117-131).
 * CURMASK is an output: 1, or 0 where the second call masked the point.  Returns the number of points still failing. */
int ora_ctuw_gen(int n, int IREFRA, int LLCFLCUROFF, real DELPRO, int MSTART, int MEND, const int *KXLT, const real *ZDELLO,
                 real XDELLA, const real *COSPH, const real *SINPH, const int *KLON, const int *KLAT, const real *WLAT,
                 const real *WCOR, const real *WLATM1, const real *WCORM1, const real *DP, const real *CGROUP_EXT,
                 const real *OMOSNH2KD_EXT, const real *COSPHM1_EXT, const real *U_EXT, const real *V_EXT, const real *THDC,
                 const real *THDD, const real *SDOT, real *CURMASK, real *SUMWN, real *WLONN, real *WLATN, real *WCORN,
                 real *WKPMN, real *WMPMN, int *LCFLFAIL) {
  for (int IJ = 0; IJ < n; IJ++) CURMASK[IJ] = C_(1.0);
  ctuw_gen_call(n, IREFRA, DELPRO, MSTART, MEND, KXLT, ZDELLO, XDELLA, COSPH, SINPH, KLON, KLAT, WLAT, WCOR, WLATM1, WCORM1, DP,
                CGROUP_EXT, OMOSNH2KD_EXT, COSPHM1_EXT, U_EXT, V_EXT, THDC, THDD, SDOT, CURMASK, SUMWN, WLONN, WLATN, WCORN, WKPMN,
                WMPMN, LCFLFAIL);
  int nfail = 0;
  for (int IJ = 0; IJ < n; IJ++) nfail += LCFLFAIL[IJ];
  if (nfail && LLCFLCUROFF && (IREFRA == 2 || IREFRA == 3)) {
    for (int IJ = 0; IJ < n; IJ++) CURMASK[IJ] = LCFLFAIL[IJ] ? C_(0.0) : C_(1.0);
    ctuw_gen_call(n, IREFRA, DELPRO, MSTART, MEND, KXLT, ZDELLO, XDELLA, COSPH, SINPH, KLON, KLAT, WLAT, WCOR, WLATM1, WCORM1,
                  DP, CGROUP_EXT, OMOSNH2KD_EXT, COSPHM1_EXT, U_EXT, V_EXT, THDC, THDD, SDOT, CURMASK, SUMWN, WLONN, WLATN, WCORN,
                  WKPMN, WMPMN, LCFLFAIL);
    nfail = 0;
    for (int IJ = 0; IJ < n; IJ++) nfail += LCFLFAIL[IJ];
  }
  if (!nfail) apply_obstructions(n, MSTART, MEND, WLONN, WLATN, WCORN);
  return nfail;
}

/* propags2.F90:124-192 (IREFRA = 2, 3): every space, direction and frequency neighbour, each term guarded by the
 * "some point has a positive weight" flags of ctuwupdt.F90:263-340, which are rebuilt here from the weights. */
void ora_propags2_gen(int n, const real *F1, real *F3, const int *KLON, const int *KLAT, const int *KCOR, const real *SUMWN,
                      const real *WLONN, const real *WLATN, const real *WCORN, const real *WKPMN, const real *WMPMN, int ND3S,
                      int ND3E) {
  const int NANG = S.NANG, NFRE = S.NFRE, NR = S.NFRE_RED;
  /* LLWLONN(K,M,2), LLWLATN(K,M,2,2), LLWCORN(K,M,4,2), LLWKPMN(K,M,-1:1), LLWMPMN(K,M,-1:1) */
  unsigned char *LL = (unsigned char *)calloc((size_t)NANG * NR * 20, 1);
  for (int IJ = 0; IJ < n; IJ++)
    for (int K = 0; K < NANG; K++)
      for (int M = 0; M < NR; M++) {
        size_t b = ((size_t)IJ * NANG + K) * NR + M;
        unsigned char *l = LL + ((size_t)K * NR + M) * 20;
        for (int i = 0; i < 2; i++) if (WLONN[b * 2 + i] > C_(0.0)) l[i] = 1;
        for (int i = 0; i < 4; i++) if (WLATN[b * 4 + i] > C_(0.0)) l[2 + i] = 1;
        for (int i = 0; i < 8; i++) if (WCORN[b * 8 + i] > C_(0.0)) l[6 + i] = 1;
        for (int i = 0; i < 3; i++) if (WKPMN[b * 3 + i] > C_(0.0)) l[14 + i] = 1;
        for (int i = 0; i < 3; i++) if (WMPMN[b * 3 + i] > C_(0.0)) l[17 + i] = 1;
      }
#define FF1(ij, k, m) F1[((size_t)(ij) * NANG + (k)) * NFRE + (m)]
#pragma omp parallel for schedule(static)
  for (int IJ = 0; IJ < n; IJ++) {
    for (int K = 0; K < NANG; K++) {
      const int *kc = S.KCR[K];
      for (int M = ND3S - 1; M < ND3E; M++) {
        size_t b = ((size_t)IJ * NANG + K) * NR + M;
        const unsigned char *l = LL + ((size_t)K * NR + M) * 20;
        real r = (C_(1.0) - SUMWN[b]) * FF1(IJ, K, M);
        for (int IC = 0; IC < 2; IC++)
          if (l[IC]) r = r + WLONN[b * 2 + IC] * FF1(KLON[IJ * 2 + IC], K, M);
        for (int ICL = 0; ICL < 2; ICL++) {
          for (int IC = 0; IC < 2; IC++)
            if (l[2 + IC * 2 + ICL]) r = r + WLATN[(b * 2 + IC) * 2 + ICL] * FF1(KLAT[(IJ * 2 + IC) * 2 + ICL], K, M);
          for (int ICR = 0; ICR < 4; ICR++)
            if (l[6 + ICR * 2 + ICL]) r = r + WCORN[(b * 4 + ICR) * 2 + ICL] * FF1(KCOR[(IJ * 4 + (kc[ICR] - 1)) * 2 + ICL], K, M);
        }
        for (int IC = -1; IC <= 1; IC += 2) {
          if (l[14 + IC + 1]) r = r + WKPMN[b * 3 + IC + 1] * FF1(IJ, S.KPM[K][IC + 1] - 1, M);
          int MS = M + IC; /* MPM(M,IC), ctuwupdt.F90:96-100 */
          MS = MS < 0 ? 0 : (MS > NR - 1 ? NR - 1 : MS);
          if (l[17 + IC + 1]) r = r + WMPMN[b * 3 + IC + 1] * FF1(IJ, K, MS);
        }
        F3[((size_t)IJ * NANG + K) * NFRE + M] = r;
      }
    }
  }
#undef FF1
  free(LL);
}
