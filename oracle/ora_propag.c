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
  for (int K = 1; K <= NANG; K++) {
    int KP1 = K + 1; if (KP1 > NANG) KP1 = 1;
    int KM1 = K - 1; if (KM1 < 1) KM1 = NANG;
    real SP = DELTH0 * (S.SINTH[K - 1] + S.SINTH[KP1 - 1]) / S.R;
    real SM = DELTH0 * (S.SINTH[K - 1] + S.SINTH[KM1 - 1]) / S.R;
    for (int IJ = 0; IJ < n; IJ++) {
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
  for (int K = 1; K <= NANG; K++)
    for (int M = MSTART; M <= MEND; M++)
      for (int IJ = 0; IJ < n; IJ++) {
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
