// Point-wise routines of IMPLSCH shared by the builds of k_implsch4 (implsch_v4.h) and by the second implementation the tests keep
// (tests/csrc/implsch_v2.h): the scalar slots of a sea point's LDS row, NS_GC, the drag law, STRESSO / TAU_PHI_HF lane per point and across
// the quadrature nodes (stresso.F90, tau_phi_hf.F90), the swell-damping set-up (sinput_ard.F90:179-271), SDIWBK (sdiwbk.F90), TRANSF /
// TRANSF_SNL (transf.F90, transf_snl.F90) and the ice wavenumber AKI_ICE (aki_ice.F90).  (Round 5: split off implsch_v2.h when the
// one-point-per-wavefront kernel k_implsch2 left the product.)
#pragma once
#include "implsch_common.h"

// the scalar slots of a sea point's row in LDS
enum {
  C_WSWAVE = 0, C_WDWAVE, C_TAUW, C_TAUWDIR, C_UFRIC, C_Z0M, C_Z0B, C_CHRNCK, C_AIRD, C_WSTAR, C_RNFAC, C_RAORW,
  C_XS, C_YS, C_PHIWA, C_MIJ, C_F1DCOS3, C_F1DCOS2, C_F1DSIN2, C_F1D, C_UORBT, C_AORB, C_SIGN, C_TEMP2, C_PTURB, C_AIRDPVISC,
  C_EMEAN, C_F1MEAN, C_EMAXDPT, C_DEPTH, C_SDS, C_SPARE,
  C_XSN, C_YSN, C_UST, C_SINU, C_COSU, C_XLOGGZ0, C_SQRTGZ0, C_ZINF, C_SINWD, C_COSWD, C_TWSIN, C_TWCOS, C_ZSUP, C_HALP,
  C_ZPIFRMIJ, C_FR5MIJ,   // ZPIFR(MIJ), FR5(MIJ): filled with C_MIJ so that STRESSO's dependent chain starts from LDS, not from a table in memory
  NSC  // NSC = 48
};

// ---- LLGCBZ0 = T (gravity-capillary roughness model, flag set B) -------------------------------------------------------
// ns_gc.F90:47-49
template <typename T>
__device__ __forceinline__ int ns_gc_d(const DevTab<T>& tb, T USTAR) {
  const T XKS = tb.SQRTGOSURFT / (T(1.48) + T(2.05) * USTAR);
  const int n = (int)(m_log(m_max(XKS * tb.XKM_GC[1], T(1))) * tb.XLOGKRATIOM1_GC) + 1;
  return n < tb.NWAV_GC - 1 ? n : tb.NWAV_GC - 1;
}
// cdm.func.h
template <typename T>
__device__ __forceinline__ T cdm_d(T U) {
  return m_max(m_min(T(0.0006) + T(0.00008) * U, T(0.001) + T(0.0018) * m_exp(-T(0.05) * (U - T(33.)))), T(0.001));
}
// STRESSO's scalar half (stresso.F90:180-229) with TAU_PHI_HF (tau_phi_hf.F90:125-301), one point per lane, in three steps
// so that the 19 quadrature nodes of every point of the block are evaluated side by side (lane = point*JTOT + node) and
// only the sheltering recurrence itself is serial:
//   head : c[C_XS], c[C_YS] (resolved-range stress integrals) -> sheltered friction velocity and direction, node geometry
//   nodes: Y(J), CM1(J), XLOGGZ0 + 2 LOG(CM1(J)) in registers of lane point*JTOT + J
//   tail : the TAUHF / PHIHF recurrences, TAUW, TAUWDIR, PHIWA
template <typename T, bool RARE>
__device__ void stresso_head_pt(const DevTab<T>& tb, T* c) {
  const T AIRD = c[C_AIRD], UFRIC = c[C_UFRIC], Z0M = c[C_Z0M];
  const int MIJ = (int)c[C_MIJ];
  const T XSTRESS = c[C_XS] / m_max(AIRD, T(1));
  const T YSTRESS = c[C_YS] / m_max(AIRD, T(1));
  // USDIRP = ATAN2(TAUPX,TAUPY) is only ever used as SIN/COS(USDIRP) = TAUPX/|TAUP|, TAUPY/|TAUP|; UST = |TAUP|**0.5
  const T sinwd = c[C_SINWD], coswd = c[C_COSWD];
  T UST, SINU, COSU;
  if (tb.TAUWSHELTER == T(0)) { UST = UFRIC; SINU = sinwd; COSU = coswd; }
  else {
    const T TAUPX = UFRIC * UFRIC * sinwd - tb.TAUWSHELTER * XSTRESS;
    const T TAUPY = UFRIC * UFRIC * coswd - tb.TAUWSHELTER * YSTRESS;
    const T h = m_sqrt(TAUPX * TAUPX + TAUPY * TAUPY);
    const bool zero = !(h > T(0));
    SINU = zero ? T(0) : TAUPX / h;
    COSU = zero ? T(1) : TAUPY / h;
    UST = m_sqrt(h);
  }
  c[C_XSN] = XSTRESS; c[C_YSN] = YSTRESS; c[C_UST] = UST;
  c[C_SINU] = SINU; c[C_COSU] = COSU;
  const T X0G = tb.X0TAUHF * tb.G;
  const T OMEGACC = m_max(c[C_ZPIFRMIJ], X0G / UST);
  const T SQRTZ0OG = m_sqrt(Z0M * tb.GM1);
  c[C_XLOGGZ0] = m_log(tb.G * Z0M);
  c[C_SQRTGZ0] = T(1) / SQRTZ0OG;
  c[C_ZINF] = m_log(OMEGACC * SQRTZ0OG);
  // upper limit of the TAUHF quadrature: 0, or the gravity-capillary transition OMEGA_GC(NS_GC(UFRIC)) (tau_phi_hf.F90:127,
  // omegagc.F90:51-55) when LLGCBZ0
  T ZSUP = T(0);
  if ((RARE && tb.LLGCBZ0)) ZSUP = m_min(m_log(tb.OMEGA_GC[ns_gc_d(tb, UFRIC)] * SQRTZ0OG), T(0));
  c[C_ZSUP] = ZSUP;
}
template <typename T>
__device__ __forceinline__ void stresso_node(const DevTab<T>& tb, const T* c, int J, T ZSUP, T& nY, T& nCM1, T& nLC) {
  const T ZINF = c[C_ZINF];
  const T DELZ = m_max((ZSUP - ZINF) / T(JTOT - 1), T(0));
  const T Y = m_exp(ZINF + T(J) * DELZ);
  const T CM1 = (Y * c[C_SQRTGZ0]) * tb.GM1;
  nY = Y; nCM1 = CM1; nLC = c[C_XLOGGZ0] + T(2) * m_log(CM1);
}
// nodeN / nodeP (J, Y, CM1, LC): the J-th quadrature node of the momentum / of the energy flux integral
template <typename T, bool RARE, typename FN, typename FP>
__device__ __forceinline__ void stresso_tail_core(const DevTab<T>& tb, T* c, FN nodeN, FP nodeP, bool store, bool LLPHIWA) {
  const T AIRD = c[C_AIRD], UFRIC = c[C_UFRIC], Z0M = c[C_Z0M], RNFAC = c[C_RNFAC];
  const T F1DCOS3 = c[C_F1DCOS3], F1DCOS2 = c[C_F1DCOS2];
  const int MIJ = (int)c[C_MIJ];
  const bool LTAUWSHELTER = (tb.TAUWSHELTER != T(0));
  T UST = c[C_UST];
  T USTPH = UST;
  const T SQRTGZ0 = c[C_SQRTGZ0];
  const T SQRTZ0OG = m_sqrt(Z0M * tb.GM1);
  const T ZINF = c[C_ZINF];
  const T fr5 = c[C_FR5MIJ];
  const T CONSTTAU = tb.ZPI4GM2 * fr5;
  T CONST1 = T(0), CONST2 = T(0);
  const bool NORMA = tb.LLNORMAGAM != 0;
  if (NORMA) {
    const T CONFG = tb.GAMNCONST * fr5 * RNFAC * SQRTGZ0;
    CONST1 = CONFG * c[C_F1DSIN2];
    CONST2 = CONFG * c[C_F1D];
  }
  T TAUL = UST * UST;
  T DELZ = m_max((c[C_ZSUP] - ZINF) / T(JTOT - 1), T(0));
  T TAUHF = T(0), acc = T(0);
  for (int J = 0; J < JTOT; J++) {
    T Y, CM1, LC;
    nodeN(J, Y, CM1, LC);
    const T ZARG = tb.XKAPPA * f_rcp(UST * CM1 + tb.ZALP);
    const T ZLOG = m_min(LC + ZARG, T(0));
    const T ZBETA = m_pow4(ZLOG) * f_exp(ZLOG);
    const T ZNZ = ZBETA * UST * Y;
    const T GAMNORMA = NORMA ? f_div(T(1) + CONST1 * ZNZ, T(1) + CONST2 * ZNZ) : T(1);
    if (LTAUWSHELTER) {
      const T FNC2 = F1DCOS3 * CONSTTAU * ZBETA * TAUL * tb.WTAUHF[J] * DELZ * GAMNORMA;
      TAUL = m_max(TAUL - tb.TAUWSHELTER * FNC2, T(0));
      UST = f_sqrt(TAUL);
      TAUHF = TAUHF + FNC2;
    } else {
      acc = acc + (ZBETA * tb.WTAUHF[J]) * GAMNORMA;
    }
  }
  if (!LTAUWSHELTER) TAUHF = F1DCOS3 * CONSTTAU * TAUL * acc * DELZ;
  T PHIHF = T(0);
  if (LLPHIWA) {
    TAUL = USTPH * USTPH;
    DELZ = m_max((T(0) - ZINF) / T(JTOT - 1), T(0));  // ZSUP = ZSUPMAX = 0 for the energy flux (tau_phi_hf.F90:246-248)
    const T CONSTPHI = AIRD * tb.ZPI4GM1 * fr5;
    for (int J = 0; J < JTOT; J++) {
      T Y, CM1, LC;
      nodeP(J, Y, CM1, LC);
      const T ZARG = tb.XKAPPA * f_rcp(USTPH * CM1 + tb.ZALP);
      const T ZLOG = m_min(LC + ZARG, T(0));
      const T ZBETA = m_pow4(ZLOG) * f_exp(ZLOG);
      const T ZNZ = ZBETA * UST * Y;
      const T GAMNORMA = NORMA ? f_div(T(1) + CONST1 * ZNZ, T(1) + CONST2 * ZNZ) : T(1);
      if (LTAUWSHELTER) {
        const T FNC2 = ZBETA * TAUL * tb.WTAUHF[J] * DELZ * GAMNORMA;
        TAUL = m_max(TAUL - tb.TAUWSHELTER * F1DCOS3 * CONSTTAU * FNC2, T(0));
        USTPH = f_sqrt(TAUL);
        PHIHF = PHIHF + FNC2 * (T(1) / Y);
      } else {
        PHIHF = PHIHF + ((ZBETA * tb.WTAUHF[J]) * GAMNORMA) * (T(1) / Y);
      }
    }
    if (LTAUWSHELTER) PHIHF = F1DCOS2 * CONSTPHI * SQRTZ0OG * PHIHF;
    else PHIHF = F1DCOS2 * CONSTPHI * SQRTZ0OG * TAUL * PHIHF * DELZ;
  }
  const T XSTRESS = c[C_XSN] + TAUHF * c[C_SINU];
  const T YSTRESS = c[C_YSN] + TAUHF * c[C_COSU];
  const T r = m_sqrt(XSTRESS * XSTRESS + YSTRESS * YSTRESS);
  T TAUW = m_max(r, T(0));
  if (!(RARE && tb.LLGCBZ0)) TAUW = m_min(TAUW, UFRIC * UFRIC * (T(1) / (T(1) + tb.EPS1)));
  if (store) {
    c[C_TAUW] = TAUW;
    // TAUWDIR = ATAN2(XSTRESS,YSTRESS): the next TAUT_Z0 needs COS(WDWAVE-TAUWDIR) only, the angle itself is an output
    const bool zero = !(r > T(0));
    c[C_TWSIN] = zero ? T(0) : XSTRESS / r;
    c[C_TWCOS] = zero ? T(1) : YSTRESS / r;
    if (LLPHIWA) {
      c[C_TAUWDIR] = m_atan2(XSTRESS, YSTRESS);
      c[C_PHIWA] = c[C_PHIWA] + PHIHF;
    }
  }
}
// nY/nCM1/nLC: node values held by lane nbase+J (pulled through the LDS crossbar; every lane of the wave runs this)
template <typename T, bool RARE>
__device__ void stresso_tail_pt(const DevTab<T>& tb, T* c, T nY, T nCM1, T nLC, T pY, T pCM1, T pLC, int nbase, bool store, bool LLPHIWA) {
  stresso_tail_core<T, RARE>(
      tb, c, [&](int J, T& Y, T& CM1, T& LC) { Y = lane_pull(nY, nbase + J); CM1 = lane_pull(nCM1, nbase + J); LC = lane_pull(nLC, nbase + J); },
      [&](int J, T& Y, T& CM1, T& LC) { Y = lane_pull(pY, nbase + J); CM1 = lane_pull(pCM1, nbase + J); LC = lane_pull(pLC, nbase + J); }, store, LLPHIWA);
}
// the whole of STRESSO's scalar half for one point on one lane (no lane exchange): the finishing kernel of the fourth generation
template <typename T, bool RARE>
__device__ __forceinline__ void stresso_point(const DevTab<T>& tb, T* c, bool LLPHIWA) {
  stresso_head_pt<T, RARE>(tb, c);
  const T ZSUP = c[C_ZSUP];
  stresso_tail_core<T, RARE>(
      tb, c, [&](int J, T& Y, T& CM1, T& LC) { stresso_node(tb, c, J, ZSUP, Y, CM1, LC); },
      [&](int J, T& Y, T& CM1, T& LC) { stresso_node(tb, c, J, (RARE && tb.LLGCBZ0) ? T(0) : ZSUP, Y, CM1, LC); }, true, LLPHIWA);
}
// the three steps on the stage's wave: lanes < WPB own a point, lanes < WPB*JTOT a node
template <typename T, int WPB, bool RARE>
__device__ __forceinline__ void stresso_stage(const DevTab<T>& tb, T* sSC, int lane, bool LLPHIWA) {
  static_assert(WPB * JTOT <= 64, "one lane per (point, node)");
  if (lane < WPB) stresso_head_pt<T, RARE>(tb, sSC + lane * NSC);
  WSYNC();
  T nY = T(1), nCM1 = T(1), nLC = T(0), pY, pCM1, pLC;
  if (lane < WPB * JTOT) {
    const int pt = lane / JTOT;
    stresso_node(tb, sSC + pt * NSC, lane - pt * JTOT, sSC[pt * NSC + C_ZSUP], nY, nCM1, nLC);
  }
  pY = nY; pCM1 = nCM1; pLC = nLC;
  if ((RARE && tb.LLGCBZ0) && LLPHIWA && lane < WPB * JTOT) {  // the energy-flux quadrature keeps ZSUP = 0: its own node set
    const int pt = lane / JTOT;
    stresso_node(tb, sSC + pt * NSC, lane - pt * JTOT, T(0), pY, pCM1, pLC);
  }
  const int pt = lane < WPB ? lane : WPB - 1;  // spare lanes shadow the last point (the pulls need the whole wave)
  stresso_tail_pt<T, RARE>(tb, sSC + pt * NSC, nY, nCM1, nLC, pY, pCM1, pLC, pt * JTOT, lane < WPB, LLPHIWA);
}

// scalar set-up of the swell damping (sinput_ard.F90:213-262) from the orbital integrals c[C_UORBT], c[C_AORB]
template <typename T>
__device__ void swell_setup_pt(const DevTab<T>& tb, T* c) {
  const T UFRIC = c[C_UFRIC], Z0M = c[C_Z0M];
  const T NU_AIR = tb.RNU;
  const T FACM1_NU_AIR = T(4) / NU_AIR;
  const T DELABM1 = T(ECWAM_HIP_IAB) / (tb.ABMAX - tb.ABMIN);
  const T UORBT = T(2) * m_sqrt(c[C_UORBT]);
  const T AORB = T(2) * m_sqrt(c[C_AORB]);
  const T RE = FACM1_NU_AIR * UORBT * AORB;
  const T Z0VIS = tb.RNUM / m_max(UFRIC, T(0.0001));
  const T Z0TUB = tb.Z0RAT * m_min(tb.Z0TUBMAX, Z0M);
  const T Z0NOZ = m_max(Z0VIS, Z0TUB);
  const T ZORB = AORB / Z0NOZ;
  const T XI = (m_log10(m_max(ZORB, T(3))) - tb.ABMIN) * DELABM1;
  int IND = (int)XI;
  if (IND > ECWAM_HIP_IAB - 1) IND = ECWAM_HIP_IAB - 1;
  const T DELI1 = m_min(T(1), XI - (T)IND);
  const T DELI2 = T(1) - DELI1;
  const T FWW = tb.SWELLFT[IND] * DELI2 + tb.SWELLFT[IND + 1] * DELI1;
  c[C_TEMP2] = FWW * UORBT;
  T RE_C;
  if (tb.SWELLF6 == T(1)) RE_C = tb.SWELLF4;
  else RE_C = tb.SWELLF4 * m_pow(T(2) / AORB, T(1) - tb.SWELLF6);
  T PVISC, PTURB;
  if (tb.SWELLF7 > T(0)) {
    const T SMOOTH = T(0.5) * m_tanh((RE - RE_C) * tb.SWELLF7M1);
    PTURB = T(0.5) + SMOOTH;
    PVISC = T(0.5) - SMOOTH;
  } else if (RE <= RE_C) { PTURB = T(0); PVISC = T(0.5); }
  else { PTURB = T(0.5); PVISC = T(0); }
  c[C_PTURB] = PTURB;
  c[C_AIRDPVISC] = PVISC * c[C_RAORW];
}

// sdiwbk.F90:88-103, one point per lane
template <typename T>
__device__ T sdiwbk_pt(const DevTab<T>& tb, T EMAXDPT, T EMEAN, T F1MEAN, T DEPTH) {
  if (!(tb.LBIWBK && DEPTH < T(50.0))) return T(0);
  const T ALPH = T(2) * EMAXDPT / EMEAN;
  const T ARG = m_min(ALPH, T(50));
  T Q_OLD = m_exp(-ARG), Q = T(0);
  for (int ic = 0; ic < 15; ic++) {
    const T EXPQ = m_exp(-ARG * (T(1) - Q_OLD));
    Q = Q_OLD - (EXPQ - Q_OLD) / (ARG * EXPQ - T(1));
    const T REL_ERR = m_abs(Q - Q_OLD) / Q_OLD;
    if (REL_ERR < T(0.00001)) break;
    Q_OLD = Q;
  }
  Q = m_min(Q, T(1));
  return T(2) * ALPH * Q * F1MEAN;
}

// transf.F90:44-71
template <typename T>
__device__ T transf_d(const DevTab<T>& tb, T XK, T D) {
  const T EPS = T(0.0001), DKMAX = T(40.0);
  if (D < tb.BATHYMAX && D > T(0)) {
    const T X = XK * D;
    if (X > DKMAX) return T(1);
    const T T_0 = m_tanh(X);
    const T OM = m_sqrt(tb.G * XK * T_0);
    const T C_0 = OM / XK;
    T V_G;
    if (X < EPS) V_G = C_0;
    else V_G = T(0.5) * C_0 * (T(1) + T(2) * X / m_sinh(T(2) * X));
    const T a = T_0 - X * (T(1) - T_0 * T_0);
    const T DV_G = a * a + T(4) * (X * X) * (T_0 * T_0) * (T(1) - T_0 * T_0);
    const T t4 = m_pow4(T_0);
    const T XNL_1 = (T(9) * t4 - T(10) * (T_0 * T_0) + T(9)) / (T(8) * (T_0 * T_0 * T_0));
    const T b = T(2) * V_G - T(0.5) * C_0;
    const T XNL_2 = (b * b / (tb.G * D - V_G * V_G) + T(1)) / X;
    const T XNL = XNL_1 - XNL_2;
    return (XNL * XNL) / (DV_G * (t4 * t4));
  }
  return T(1);
}

// aki_ice.F90:60-112: wave number under an elastic ice sheet, Newton iteration (per lane)
template <typename T>
__device__ T aki_ice_d(T G, T XK, T DEPTH, T RHOW, T CITH) {
  const T YMICE = T(5.5E+9), RMUICE = T(0.3), RHOI = T(922.5), EBS = T(0.000001), AKI_MAX = T(20.0);
  if (CITH <= T(0)) return XK;
  const T FICSTF = (YMICE * (CITH * CITH * CITH) / (T(12) * (T(1) - RMUICE * RMUICE))) / RHOW;
  const T RDH = (RHOI / RHOW) * CITH;
  const T OM2 = G * XK * m_tanh(XK * DEPTH);
  T AKIOLD = T(0);
  T AKI = m_min(XK, m_pow(OM2 / m_max(FICSTF, T(1)), T(0.2)));
  for (int it = 0; it < 200 && m_abs(AKI - AKIOLD) > EBS * AKIOLD && AKI < AKI_MAX; it++) {
    AKIOLD = AKI;
    const T AKID = m_min(DEPTH * AKI, T(50.0));
    const T a2 = AKI * AKI, a4 = a2 * a2;
    const T Fv = FICSTF * (a4 * AKI) + G * AKI - OM2 * (RDH * AKI + T(1) / m_tanh(AKID));
    const T sh = m_sinh(AKID);
    const T FPRIME = T(5) * FICSTF * a4 + G - OM2 * (RDH - DEPTH / (sh * sh));
    AKI = AKI - Fv / FPRIME;
    if (AKI <= T(0)) AKI = AKI_MAX;
  }
  return AKI;
}

// transf_snl.F90:52-85
template <typename T>
__device__ T transf_snl_d(const DevTab<T>& tb, T XK0, T D, T XNU, T SIG_TH) {
  const T EPS = T(0.0001), DKMAX = T(40.0), XKDMIN = T(0.75);
  if (D < tb.BATHYMAX && D > T(0)) {
    T X = XK0 * D;
    if (X > DKMAX) return T(1);
    const T XK = m_max(XK0, XKDMIN / D);
    X = XK * D;
    const T T_0 = m_tanh(X);
    const T T_0_SQ = T_0 * T_0;
    const T OM = m_sqrt(tb.G * XK * T_0);
    const T C_0 = OM / XK;
    const T C_S_SQ = tb.G * D;
    T V_G;
    if (X < EPS) V_G = C_0;
    else V_G = T(0.5) * C_0 * (T(1) + T(2) * X / m_sinh(T(2) * X));
    const T V_G_SQ = V_G * V_G;
    const T a = T_0 - X * (T(1) - T_0_SQ);
    const T DV_G = a * a + T(4) * (X * X) * T_0_SQ * (T(1) - T_0_SQ);
    const T XNL_1 = (T(9) * (T_0_SQ * T_0_SQ) - T(10) * T_0_SQ + T(9)) / (T(8) * T_0_SQ * T_0);
    const T b = T(2) * V_G - T(0.5) * C_0;
    const T XNL_2 = (b * b / (tb.G * D - V_G_SQ) + T(1)) / X;
    const T c = T(2) * C_0 + V_G * (T(1) - T_0_SQ);
    const T XNL_4 = T(1) / (T(4) * T_0) * (c * c) / (C_S_SQ - V_G_SQ);
    const T ALP = (T(1) - V_G_SQ / C_S_SQ) * (C_0 * C_0) / V_G_SQ;
    const T s2 = SIG_TH * SIG_TH;
    const T ZFAC = s2 / (s2 + ALP * (XNU * XNU));
    const T XNL = XNL_1 - XNL_2 + ZFAC * XNL_4;
    const T t4 = T_0_SQ * T_0_SQ;
    const T r = (XNL * XNL) / (DV_G * (t4 * t4));
    return m_max(m_min(T(10), r), T(0.1));
  }
  return T(1);
}

