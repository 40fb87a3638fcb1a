// IMPLSCH (implsch.F90 and the routines it inlines) as one fused CDNA4 kernel.
//
// Mapping: one wavefront (64 lanes) integrates one sea point; lane k owns direction K=k+1
// (NANG <= 48 lanes active).  The point's spectrum F, the functional derivative FLD and the
// source function SL live in wave-private LDS as [M][NAP] tiles (NAP = NANG|1, odd), so that
//   * lane=K row accesses   sX[m*NAP + k]   (all source terms, DIA gather/scatter, 17-tap
//     directional filter) are conflict-free, and
//   * lane=M column accesses sX[m*NAP + kk], kk sequential (the reference's own DO K summation
//     order for the spectral integrals TEMP2(M) = SUM_K F(K,M)) are conflict-free too.
// Per-frequency point properties (WAVNUM, CINV, XK2CG, STOKFAC) stay in registers of lane m and are
// broadcast with v_readlane; module tables are read from a const DevTab with wave-uniform indices.
// HBM traffic per point = the algorithmic minimum: F in, F out, XLLWS out, 5*NFRE + ~50 scalars.
// There is no inter-wave communication: blocks of WPB waves share nothing but the LDS allocation.
#pragma once
#include "dev.h"

#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

template <typename T>
struct Lane {
  int lane, k, NANG, NFRE, NAP;
  bool act;    // lane < NANG
  bool actm;   // lane < NFRE
  // module tables per frequency, lane m holds M=m+1 (broadcast with v_readlane: no scalar loads inside the M loops)
  T rDFIM, rDFIMOFR, rZPIFR, rCOFRM4, rFLMAX;
};

// TEMP2(M) = SUM_K F(K,M) in the reference's order (K sequential), lane m gets M=m+1
template <typename T>
__device__ __forceinline__ T colsum(const T* sF, const Lane<T>& L) {
  T t = T(0);
  if (L.actm) {
    const T* p = sF + L.lane * L.NAP;
    t = p[0];
    for (int kk = 1; kk < L.NANG; kk++) t = t + p[kk];
  }
  return t;
}

// fkmean.F90:94-150
template <typename T>
__device__ void fkmean(const DevTab<T>& tb, const T* sF, const Lane<T>& L, T rWAVNUM, T& EM, T& FM1, T& F1, T& AK, T& XK) {
  const T temp2 = colsum(sF, L);
  T dfim = T(0), dfo = T(0), dff = T(0), ta = T(0), tx = T(0);
  if (L.actm) {
    dfim = tb.DFIM[L.lane]; dfo = tb.DFIMOFR[L.lane]; dff = tb.DFIMFR[L.lane];
    T sq = m_sqrt(rWAVNUM);
    ta = dfim / sq; tx = sq * dfim;
  }
  const T frl = tb.FR[L.NFRE - 1];
  const T DELT25 = tb.WETAIL * frl * tb.DELTH;
  const T COEFM1 = tb.FRTAIL * tb.DELTH;
  const T COEF1 = tb.WP1TAIL * tb.DELTH * frl * frl;
  const T COEFA = COEFM1 * m_sqrt(tb.G) / tb.ZPI;
  const T COEFX = COEF1 * (tb.ZPI / m_sqrt(tb.G));
  const T tl = lane_get(temp2, L.NFRE - 1);
  EM = tb.EPSMIN + usum(dfim * temp2);
  FM1 = tb.EPSMIN + usum(dfo * temp2);
  F1 = tb.EPSMIN + usum(dff * temp2);
  AK = tb.EPSMIN + usum(ta * temp2);
  XK = tb.EPSMIN + usum(tx * temp2);
  EM = EM + DELT25 * tl;
  FM1 = FM1 + COEFM1 * tl;
  FM1 = EM / FM1;
  F1 = F1 + COEF1 * tl;
  F1 = F1 / EM;
  AK = AK + COEFA * tl;
  AK = (EM / AK) * (EM / AK);
  XK = XK + COEFX * tl;
  XK = (XK / EM) * (XK / EM);
}

// femeanws.F90:84-123 ; XLLWS(K,M) is bit M of lane K's mask
template <typename T>
__device__ void femeanws(const DevTab<T>& tb, const T* sF, const Lane<T>& L, unsigned long long xmask, T& FM, T& EMW) {
  T ae = T(0), af = T(0), last = T(0);
  for (int m = 0; m < L.NFRE; m++) {
    T x = ((xmask >> m) & 1ull) ? sF[m * L.NAP + L.k] : T(0);
    ae += lane_get(L.rDFIM, m) * x;
    af += lane_get(L.rDFIMOFR, m) * x;
    last = x;
  }
  if (!L.act) { ae = T(0); af = T(0); last = T(0); }
  const T t2 = usum(last);
  const T DELT25 = tb.WETAIL * tb.FR[L.NFRE - 1] * tb.DELTH;
  const T DELT2 = tb.FRTAIL * tb.DELTH;
  T em = tb.EPSMIN + usum(ae);
  T fm = tb.EPSMIN + usum(af);
  em = em + DELT25 * t2;
  fm = fm + DELT2 * t2;
  FM = em / fm;
  EMW = em;
}

// chnkmin.F90:58
template <typename T>
__device__ __forceinline__ T chnkmin(const DevTab<T>& tb, T U10) {
  return tb.ALPHAMIN + (tb.ALPHA - tb.ALPHAMIN) * T(0.5) * (T(1) - m_tanh(U10 - tb.CHNKMIN_U));
}

// taut_z0.F90:288-340 (LLGCBZ0 = F branch).  All lanes compute the same scalars.
template <typename T>
__device__ void taut_z0_c(const DevTab<T>& tb, int IUSFG, T UTOP, T COSDIFF, T TAUW, T& USTAR, T& Z0, T& Z0B, T& CHRNCK);
template <typename T>
__device__ void taut_z0_a(const DevTab<T>& tb, int IUSFG, T UTOP, T UDIR, T TAUW, T TAUWDIR, T& USTAR, T& Z0, T& Z0B, T& CHRNCK) {
  taut_z0_c(tb, IUSFG, UTOP, m_cos(UDIR - TAUWDIR), TAUW, USTAR, Z0, Z0B, CHRNCK);
}
// same with COS(UDIR-TAUWDIR) supplied by the caller
template <typename T>
__device__ void taut_z0_c(const DevTab<T>& tb, int IUSFG, T UTOP, T COSDIFF, T TAUW, T& USTAR, T& Z0, T& Z0B, T& CHRNCK) {
  const T TWOXMP1 = T(3.0);
  const T XLOGXL = m_log(tb.XNLEV);
  const T US2TOTAUW = T(1) + tb.EPS1;
  const T TAUWACT = m_max(TAUW * COSDIFF, tb.EPSMIN);
  const T TAUWEFF = TAUWACT * US2TOTAUW;
  T XMIN, ALPHAOG;
  if (tb.LLCAPCHNK) {
    T cm = chnkmin(tb, UTOP);
    XMIN = T(0.15) * (tb.ALPHA - cm);
    ALPHAOG = cm * tb.GM1;
  } else {
    XMIN = T(0);
    ALPHAOG = tb.ALPHA * tb.GM1;
  }
  const T XKUTOP = tb.XKAPPA * UTOP;
  T USTOLD = (1 - IUSFG) * UTOP * m_sqrt(m_min(tb.ACD + tb.BCD * UTOP, tb.CDMAX)) + IUSFG * USTAR;
  T TAUOLD = m_max(USTOLD * USTOLD, TAUWEFF);
  USTAR = m_sqrt(TAUOLD);
  T USTM1 = T(1) / m_max(USTAR, tb.EPSUS);
  T Z0CH = T(0);
  for (int it = 0; it < 18; it++) {
    T X = m_max(f_div(TAUWACT, TAUOLD), XMIN);
    const T omx = f_rcp(T(1) - X);
    Z0CH = ALPHAOG * TAUOLD * f_sqrt(omx);
    T Z0VIS = tb.RNUM * USTM1;
    T Z0TOT = Z0CH + Z0VIS;
    T XOLOGZ0 = f_rcp(XLOGXL - f_log(Z0TOT));
    T Fv = USTAR - XKUTOP * XOLOGZ0;
    T ZZ = f_div(USTM1 * (Z0CH * (T(2) - TWOXMP1 * X) * omx - Z0VIS), Z0TOT);
    T DELF = T(1) - XKUTOP * XOLOGZ0 * XOLOGZ0 * ZZ;
    if (DELF != T(0)) USTAR = USTAR - f_div(Fv, DELF);
    T TAUNEW = m_max(USTAR * USTAR, TAUWEFF);
    USTAR = f_sqrt(TAUNEW);
    if (TAUNEW == TAUOLD) break;
    USTM1 = f_rcp(m_max(USTAR, tb.EPSUS));
    TAUOLD = TAUNEW;
  }
  Z0 = Z0CH;
  Z0B = ALPHAOG * TAUOLD;
  CHRNCK = m_max(tb.G * Z0 * USTM1 * USTM1, tb.ALPHAMIN);
}

// wsigstar.F90:87-129
template <typename T>
__device__ T wsigstar(const DevTab<T>& tb, T WSWAVE, T UFRIC, T Z0M, T WSTAR) {
  const T ONETHIRD = T(1) / T(3), SIG_NMAX = T(0.9);
  const T C1 = T(1.03E-3), C2 = T(0.04E-3), P1 = T(1.48), P2 = T(-0.21);
  const T w3 = T(0.5) * tb.XKAPPA * (WSTAR * WSTAR * WSTAR);  // BG_GUST = 0
  if (tb.LLGCBZ0 || tb.LLNORMAGAM) {
    T U10M1 = T(1) / m_max(WSWAVE, tb.WSPMIN);
    T Z0VIS = tb.RNUM / m_max(UFRIC, tb.EPSUS);
    T ZCHAR = tb.G * (Z0M - Z0VIS) / m_max(UFRIC * UFRIC, tb.EPSUS);
    ZCHAR = m_max(m_min(ZCHAR, tb.ALPHAMAX), tb.ALPHAMIN);
    T BCD_LOC = tb.BCDLIN * m_sqrt(ZCHAR);
    T C_D = tb.ACDLIN + BCD_LOC * WSWAVE;
    T SIG_CONV = T(1) + T(0.5) * WSWAVE / C_D * BCD_LOC;
    return m_min(SIG_NMAX, SIG_CONV * U10M1 * m_pow(T(0) + w3, ONETHIRD));
  }
  T U10 = UFRIC * (T(1) / tb.XKAPPA) * (m_log(T(10)) - m_log(Z0M));
  U10 = m_max(U10, tb.WSPMIN);
  T U10M1 = T(1) / U10;
  T C2U10P1 = C2 * m_pow(U10, P1);
  T U10P2 = m_pow(U10, P2);
  T C_D = (C1 + C2U10P1) * U10P2;
  T DC_DDU = (P2 * C1 + (P1 + P2) * C2U10P1) * U10P2 * U10M1;
  T SIG_CONV = T(1) + T(0.5) * U10 / C_D * DC_DDU;
  return m_min(SIG_NMAX, SIG_CONV * U10M1 * m_pow(T(0) + w3, ONETHIRD));
}

// sinput_ard.F90:153-520.  Writes FLD and SPOS (into the SL tile); SL = FLD*F is rebuilt by the caller.
template <typename T, int NGST, bool LLSNEG>
__device__ void sinput_ard(const DevTab<T>& tb, const T* sF, T* sFLD, T* sSPOS, const Lane<T>& L, T rWAVNUM, T rCINV, T rXK2CG,
                           T WDWAVE, T WSWAVE, T UFRIC, T Z0M, T coswdif, T sinwdif2, T RAORW, T WSTAR, T RNFAC,
                           unsigned long long& xmask) {
  const T AVG_GST = T(1) / T(NGST);
  const T CONST1 = tb.BETAMAXOXKAPPA2;
  const T CONSTN = tb.DELTH / (tb.XKAPPA * tb.ZPI);
  const T ABS_TAUWSHELTER = m_abs(tb.TAUWSHELTER);
  const bool LTAUWSHELTER = (ABS_TAUWSHELTER != T(0));
  const bool LLNORMAGAM = tb.LLNORMAGAM != 0;
  T SIG_N = T(0);
  if (NGST > 1) SIG_N = wsigstar(tb, WSWAVE, UFRIC, Z0M, WSTAR);
  T CSTRNFAC = T(0);
  if (LLNORMAGAM) CSTRNFAC = CONSTN * RNFAC / RAORW;

  T FU = T(0), FUD = T(0), NU_AIR = T(0), TEMP2 = T(0), PTURB = T(0), AIRD_PVISC = T(0);
  if (LLSNEG) {
    NU_AIR = tb.RNU;
    const T FACM1_NU_AIR = T(4) / NU_AIR;
    FU = m_abs(tb.SWELLF3);
    FUD = tb.SWELLF2;
    const T DELABM1 = T(ECWAM_HIP_IAB) / (tb.ABMAX - tb.ABMIN);
    const T temp = colsum(sF, L);
    T w1 = T(0), w2 = T(0);
    if (L.actm) {
      T sig = tb.ZPIFR[L.lane];
      w2 = tb.DFIM[L.lane];
      w1 = w2 * (sig * sig);
    }
    T UORBT = tb.EPSMIN + usum(w1 * temp);
    T AORB = tb.EPSMIN + usum(w2 * temp);
    UORBT = T(2) * m_sqrt(UORBT);
    AORB = T(2) * m_sqrt(AORB);
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
    TEMP2 = FWW * UORBT;
    T RE_C;
    if (tb.SWELLF6 == T(1)) RE_C = tb.SWELLF4;
    else RE_C = tb.SWELLF4 * m_pow(T(2) / AORB, T(1) - tb.SWELLF6);
    T PVISC;
    if (tb.SWELLF7 > T(0)) {
      T SMOOTH = T(0.5) * m_tanh((RE - RE_C) * tb.SWELLF7M1);
      PTURB = T(0.5) + SMOOTH;
      PVISC = T(0.5) - SMOOTH;
    } else if (RE <= RE_C) { PTURB = T(0); PVISC = T(0.5); }
    else { PTURB = T(0.5); PVISC = T(0); }
    AIRD_PVISC = PVISC * RAORW;
  }

  T USTP[2], USTPM1[2], XSTRESS[2], YSTRESS[2], TAUX[2], TAUY[2], USDIRP[2], UCN[2], UCNZALPD[2], GAMNORMA[2];
  if (NGST == 1) USTP[0] = UFRIC;
  else { USTP[0] = UFRIC * (T(1) + SIG_N); USTP[1] = UFRIC * (T(1) - SIG_N); }
#pragma unroll
  for (int ig = 0; ig < NGST; ig++) USTPM1[ig] = T(1) / m_max(USTP[ig], tb.EPSUS);
  T ROGOROAIR = T(0);
  if (LTAUWSHELTER) {
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      XSTRESS[ig] = T(0); YSTRESS[ig] = T(0);
      T USG2 = USTP[ig] * USTP[ig];
      TAUX[ig] = USG2 * m_sin(WDWAVE);
      TAUY[ig] = USG2 * m_cos(WDWAVE);
    }
    ROGOROAIR = tb.G / RAORW;
  }
  GAMNORMA[0] = T(1); GAMNORMA[1] = T(1);
  const T sinthk = tb.SINTH[L.k], costhk = tb.COSTH[L.k];
  xmask = 0ull;
  // per-frequency scalars that need a transcendental: evaluated once, lane m for M=m+1, broadcast in the loop
  T rZCN = T(0), rCOEF5 = T(0);
  if (L.actm) {
    rZCN = m_log(rWAVNUM * Z0M);
    if (LLSNEG) rCOEF5 = -tb.SWELLF5 * T(2) * m_sqrt(T(2) * NU_AIR * tb.ZPIFR[L.lane]);
  }
  T COSU[2], SINU[2];  // cos/sin of the sheltered stress direction USDIRP = ATAN2(TAUPX,TAUPY)

  for (int m = 0; m < L.NFRE; m++) {
    const T SIG = lane_get(L.rZPIFR, m);
    const T SIG2 = SIG * SIG;
    const T CONST = SIG * CONST1;
    const T cinv_m = lane_get(rCINV, m), wavnum_m = lane_get(rWAVNUM, m);
    T COEF = T(0), COEF5 = T(0);
    if (LLSNEG) {
      COEF = -tb.SWELLF * T(16) * SIG2 / tb.G;
      COEF5 = lane_get(rCOEF5, m);
    }
    T CONSTF = T(0);
    if (LTAUWSHELTER) {
#pragma unroll
      for (int ig = 0; ig < NGST; ig++) {
        // sinput_ard.F90:360-364.  COS(TH(K)-USDIRP) is formed below from cos/sin(USDIRP) = TAUPY/|TAUP|, TAUPX/|TAUP|
        // (no ATAN2/COS per lane) and USTP = |TAUP|**0.5 as two square roots: algebraically identical evaluation.
        T TAUPX = TAUX[ig] - ABS_TAUWSHELTER * XSTRESS[ig];
        T TAUPY = TAUY[ig] - ABS_TAUWSHELTER * YSTRESS[ig];
        const T h = f_sqrt(TAUPX * TAUPX + TAUPY * TAUPY);
        const bool zero = !(h > T(0));
        const T rh = f_rcp(h);
        COSU[ig] = zero ? T(1) : TAUPY * rh;
        SINU[ig] = zero ? T(0) : TAUPX * rh;
        USTP[ig] = f_sqrt(h);
        USTPM1[ig] = f_rcp(m_max(USTP[ig], tb.EPSUS));
      }
      CONSTF = ROGOROAIR * cinv_m * lane_get(L.rDFIM, m);
    }
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      UCN[ig] = USTP[ig] * cinv_m;
      UCNZALPD[ig] = tb.XKAPPA * f_rcp(UCN[ig] + tb.ZALP);
    }
    const T ZCN = lane_get(rZCN, m);
    const T CNSN = CONST * RAORW;
    T XNGAMCONST = T(0);
    if (LLNORMAGAM) XNGAMCONST = CSTRNFAC * lane_get(rXK2CG, m);
    T DSTAB1 = T(0), TEMP1 = T(0);
    if (LLSNEG) {
      DSTAB1 = COEF5 * AIRD_PVISC * wavnum_m;
      TEMP1 = COEF * RAORW;
    }
    const T f = sF[m * L.NAP + L.k];
    T g0[2], ds[2];
    bool xl = false, grow[2];
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      T coslp = LTAUWSHELTER ? (costhk * COSU[ig] + sinthk * SINU[ig]) : coswdif;
      T gam0 = T(0), ZLOG = T(0);
      bool neg = false;
      if (coslp > T(0.01)) {
        ZLOG = ZCN + UCNZALPD[ig] * f_rcp(coslp);
        neg = ZLOG < T(0);
      }
      // rows in which no direction grows (waves outrunning the wind): wave-uniform skip of the growth rate, its
      // normalisation and the stress reductions -- they would all add exact zeros
      grow[ig] = __builtin_amdgcn_ballot_w64(neg) != 0ull;
      if (grow[ig]) {
        if (neg) {
          T ZLOG2X = ZLOG * ZLOG * (coslp * UCN[ig]);
          gam0 = f_exp(ZLOG) * ZLOG2X * ZLOG2X * CNSN;
          xl = true;
        }
        if (LLNORMAGAM) {
          T a = L.act ? gam0 * f : T(0);
          T SUMF = usum(a);
          T SUMFSIN2 = usum(a * sinwdif2);
          T ZNZ = XNGAMCONST * USTPM1[ig];
          GAMNORMA[ig] = (T(1) + ZNZ * SUMFSIN2) / (T(1) + ZNZ * SUMF);
        }
      }
      T dstab = T(0);
      if (LLSNEG) {
        T DSTAB2 = TEMP1 * (TEMP2 + (FU + FUD * coslp) * USTP[ig]);
        dstab = DSTAB1 + PTURB * DSTAB2;
      }
      g0[ig] = gam0;
      ds[ig] = dstab;
    }
    T SLP_AVG = T(0), FLP_AVG = T(0);
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      T SLP = g0[ig] * GAMNORMA[ig];
      T FLP = SLP + ds[ig];
      SLP = SLP * f;
      if (LTAUWSHELTER && grow[ig]) {
        T sx = L.act ? SLP * (CONSTF * sinthk) : T(0);
        T sy = L.act ? SLP * (CONSTF * costhk) : T(0);
        XSTRESS[ig] = XSTRESS[ig] + usum(sx);
        YSTRESS[ig] = YSTRESS[ig] + usum(sy);
      }
      if (ig == 0) { SLP_AVG = SLP; FLP_AVG = FLP; }
      else { SLP_AVG = SLP_AVG + SLP; FLP_AVG = FLP_AVG + FLP; }
    }
    if (L.act) {
      sSPOS[m * L.NAP + L.k] = AVG_GST * SLP_AVG;
      sFLD[m * L.NAP + L.k] = AVG_GST * FLP_AVG;
    }
    if (xl) xmask |= (1ull << m);
  }
}

// tau_phi_hf.F90:125-301 (LLGCBZ0 = F: ZSUP = 0)
template <typename T>
__device__ void tau_phi_hf(const DevTab<T>& tb, const T* sF, const Lane<T>& L, int MIJ, bool LTAUWSHELTER, T Z0M, T AIRD, T RNFAC,
                           T coswdif, T sinwdif2, T& UST, T& TAUHF, T& PHIHF, bool LLPHIHF) {
  const T X0G = tb.X0TAUHF * tb.G;
  T USTPH = UST;
  const T XLOGGZ0 = m_log(tb.G * Z0M);
  const T OMEGACC = m_max(tb.ZPIFR[MIJ - 1], X0G / UST);
  const T SQRTZ0OG = m_sqrt(Z0M * tb.GM1);
  const T SQRTGZ0 = T(1) / SQRTZ0OG;
  const T ZINF = m_log(OMEGACC * SQRTZ0OG);
  const T fr5 = tb.FR5[MIJ - 1];
  const T CONSTTAU = tb.ZPI4GM2 * fr5;
  const T fm = L.act ? sF[(MIJ - 1) * L.NAP + L.k] : T(0);
  const T cosw = m_max(coswdif, T(0));
  const T fc2 = fm * cosw * cosw;
  const T F1DCOS3 = tb.DELTH * usum(fc2 * cosw);
  const T F1DCOS2 = tb.DELTH * usum(fc2);
  T CONST1 = T(0), CONST2 = T(0);
  if (tb.LLNORMAGAM) {
    const T F1DSIN2 = tb.DELTH * usum(fm * sinwdif2);
    const T F1D = tb.DELTH * usum(fm);
    const T CONFG = tb.GAMNCONST * fr5 * RNFAC * SQRTGZ0;
    CONST1 = CONFG * F1DSIN2;
    CONST2 = CONFG * F1D;
  }
  const T ZSUP = T(0);
  T TAUL = UST * UST;
  T DELZ = m_max((ZSUP - ZINF) / T(JTOT - 1), T(0));
  // the 19 integration nodes do not depend on the sheltered friction velocity: evaluate Y, CM1 and
  // XLOGGZ0+2*LOG(CM1) lane-parallel (lane J), leaving one EXP, one divide and one SQRT per sequential step
  T rY = T(1), rCM1 = T(1), rLC = T(0), rYI = T(1);
  const bool NORMA = tb.LLNORMAGAM != 0;
  if (L.lane < JTOT) {
    rY = m_exp(ZINF + T(L.lane) * DELZ);
    rYI = T(1) / rY;
    rCM1 = (rY * SQRTGZ0) * tb.GM1;
    rLC = XLOGGZ0 + T(2) * m_log(rCM1);
  }
  TAUHF = T(0);
  if (LTAUWSHELTER) {
    for (int J = 0; J < JTOT; J++) {
      const T Y = lane_get(rY, J);
      const T CM1 = lane_get(rCM1, J);
      T ZARG = tb.XKAPPA * f_rcp(UST * CM1 + tb.ZALP);
      T ZLOG = m_min(lane_get(rLC, J) + ZARG, T(0));
      T ZBETA = m_pow4(ZLOG) * f_exp(ZLOG);
      T ZNZ = ZBETA * UST * Y;
      const T GAMNORMA = NORMA ? f_div(T(1) + CONST1 * ZNZ, T(1) + CONST2 * ZNZ) : T(1);
      T FNC2 = F1DCOS3 * CONSTTAU * ZBETA * TAUL * tb.WTAUHF[J] * DELZ * GAMNORMA;
      TAUL = m_max(TAUL - tb.TAUWSHELTER * FNC2, T(0));
      UST = f_sqrt(TAUL);
      TAUHF = TAUHF + FNC2;
    }
  } else {
    for (int J = 0; J < JTOT; J++) {
      const T Y = lane_get(rY, J);
      const T CM1 = lane_get(rCM1, J);
      T ZARG = tb.XKAPPA * f_rcp(UST * CM1 + tb.ZALP);
      T ZLOG = m_min(lane_get(rLC, J) + ZARG, T(0));
      T ZBETA = m_pow4(ZLOG) * f_exp(ZLOG);
      T FNC2 = ZBETA * tb.WTAUHF[J];
      T ZNZ = ZBETA * UST * Y;
      const T GAMNORMA = NORMA ? f_div(T(1) + CONST1 * ZNZ, T(1) + CONST2 * ZNZ) : T(1);
      TAUHF = TAUHF + FNC2 * GAMNORMA;
    }
    TAUHF = F1DCOS3 * CONSTTAU * TAUL * TAUHF * DELZ;
  }
  PHIHF = T(0);
  if (LLPHIHF) {
    TAUL = USTPH * USTPH;
    DELZ = m_max((T(0) - ZINF) / T(JTOT - 1), T(0));
    const T CONSTPHI = AIRD * tb.ZPI4GM1 * fr5;
    if (LTAUWSHELTER) {
      for (int J = 0; J < JTOT; J++) {
        const T Y = lane_get(rY, J);
        const T CM1 = lane_get(rCM1, J);
        T ZARG = tb.XKAPPA * f_rcp(USTPH * CM1 + tb.ZALP);
        T ZLOG = m_min(lane_get(rLC, J) + ZARG, T(0));
        T ZBETA = m_pow4(ZLOG) * f_exp(ZLOG);
        T ZNZ = ZBETA * UST * Y;
        const T GAMNORMA = NORMA ? f_div(T(1) + CONST1 * ZNZ, T(1) + CONST2 * ZNZ) : T(1);
        T FNC2 = ZBETA * TAUL * tb.WTAUHF[J] * DELZ * GAMNORMA;
        TAUL = m_max(TAUL - tb.TAUWSHELTER * F1DCOS3 * CONSTTAU * FNC2, T(0));
        USTPH = f_sqrt(TAUL);
        PHIHF = PHIHF + FNC2 * lane_get(rYI, J);
      }
      PHIHF = F1DCOS2 * CONSTPHI * SQRTZ0OG * PHIHF;
    } else {
      for (int J = 0; J < JTOT; J++) {
        const T Y = lane_get(rY, J);
        const T CM1 = lane_get(rCM1, J);
        T ZARG = tb.XKAPPA * f_rcp(USTPH * CM1 + tb.ZALP);
        T ZLOG = m_min(lane_get(rLC, J) + ZARG, T(0));
        T ZBETA = m_pow4(ZLOG) * f_exp(ZLOG);
        T ZNZ = ZBETA * UST * Y;
        const T GAMNORMA = NORMA ? f_div(T(1) + CONST1 * ZNZ, T(1) + CONST2 * ZNZ) : T(1);
        T FNC2 = ZBETA * tb.WTAUHF[J] * GAMNORMA;
        PHIHF = PHIHF + FNC2 * lane_get(rYI, J);
      }
      PHIHF = F1DCOS2 * CONSTPHI * SQRTZ0OG * TAUL * PHIHF * DELZ;
    }
  }
}

// sdissip_ard.F90:117-314 (SSDSC3 = 0): SL = FLD*F + D*F, FLD += D.  NTAPC > 0: compile-time tap count.
template <typename T, int NTAPC>
__device__ void sdissip_ard(const DevTab<T>& tb, const T* sF, T* sFLD, T* sSL, const Lane<T>& L, T rWAVNUM, T rXK2CG, T UFRIC,
                            T coswdif, T RAORW) {
  const int ntap = NTAPC > 0 ? NTAPC : tb.NTAP;
  const T TPIINV = T(1) / tb.ZPI;
  const T TMP03 = T(1) / (tb.SDSBR * tb.MICHE);
  const T SSDSC6M1 = T(1) - tb.SSDSC6;
  T wgt[NTAPC > 0 ? NTAPC : 1];
  int idx[NTAPC > 0 ? NTAPC : 1];
  if (NTAPC > 0) {
#pragma unroll
    for (int j = 0; j < NTAPC; j++) { wgt[j] = tb.SATWEIGHTS[j][L.k]; idx[j] = tb.INDICESSAT[j][L.k]; }
  }
  const T rFACSAT = rWAVNUM * TPIINV * rXK2CG;  // lane m
  T FACTURB = T(0);
  const bool turb = (tb.SSDSC5 != T(0));
  if (turb) FACTURB = (T(2) * tb.SSDSC5 / tb.G) * RAORW * UFRIC * UFRIC;
  for (int m = 0; m < L.NFRE; m++) {
    const T* row = sF + m * L.NAP;
    T b = T(0);
    if (NTAPC > 0) {
#pragma unroll
      for (int j = 0; j < NTAPC; j++) b = b + wgt[j] * row[idx[j]];
    } else {
      for (int j = 0; j < ntap; j++) b = b + tb.SATWEIGHTS[j][L.k] * row[tb.INDICESSAT[j][L.k]];
    }
    b = b * lane_get(rFACSAT, m);
    const T bth0 = umax(L.act ? b : T(0));
    const T SSDSC2_SIG = tb.SSDSC2 * lane_get(L.rZPIFR, m);
    const T ZCOEF = SSDSC2_SIG * tb.SSDSC6;
    const T ZCOEFM1 = SSDSC2_SIG * SSDSC6M1;
    const T a0 = m_max(T(0), bth0 * TMP03 - tb.SSDSC4);
    const T a1 = m_max(T(0), b * TMP03 - tb.SSDSC4);
    T D = ZCOEF * (a0 * a0) + ZCOEFM1 * (a1 * a1);
    if (turb) D = D - (tb.ZPIFR[m] * lane_get(rWAVNUM, m) * FACTURB) * coswdif;
    if (L.act) {
      const T f = row[L.k];
      const T fld = sFLD[m * L.NAP + L.k];
      sSL[m * L.NAP + L.k] = fld * f + D * f;
      sFLD[m * L.NAP + L.k] = fld + D;
    }
  }
}

// snonlin.F90:126-494 (ISNONLIN = 0).  Lanes = K; MC and KH sequential as in the reference.
template <typename T>
__device__ void snonlin(const DevTab<T>& tb, const T* sF, T* sFLD, T* sSL, const Lane<T>& L, T DEPTH, T AKMEAN) {
  T ENHFR = m_max(T(0.75) * DEPTH * AKMEAN, T(0.5));
  ENHFR = T(1) + (T(5.5) / ENHFR) * (T(1) - T(.833) * ENHFR) * m_exp(-T(1.25) * ENHFR);
  const int MFR1STFR = -tb.MFRSTLW + 1;
  const int MFRLSTFR = L.NFRE - tb.KFRH + MFR1STFR;
  const int NAP = L.NAP, NFRE = L.NFRE;
  int K1[2], K2[2], K11[2], K21[2];
#pragma unroll
  for (int kh = 0; kh < 2; kh++) {
    K1[kh] = tb.K1W[kh][L.k]; K2[kh] = tb.K2W[kh][L.k]; K11[kh] = tb.K11W[kh][L.k]; K21[kh] = tb.K21W[kh][L.k];
  }
  volatile T* vSL = sSL;
  volatile T* vFL = sFLD;
#define ADDS(kk, m1, v) do { if (L.act) vSL[((m1) - 1) * NAP + (kk)] = vSL[((m1) - 1) * NAP + (kk)] + (v); } while (0)
#define ADDF(kk, m1, v) do { if (L.act) vFL[((m1) - 1) * NAP + (kk)] = vFL[((m1) - 1) * NAP + (kk)] + (v); } while (0)
  for (int MC = 1; MC <= tb.MLSTHG; MC++) {
    const int MP = tb.IKP[MC - 1], MP1 = tb.IKP1[MC - 1], MM = tb.IKM[MC - 1], MM1 = tb.IKM1[MC - 1];
    const int IC = tb.INLCOEF[MC - 1][0], IP = tb.INLCOEF[MC - 1][1], IP1 = tb.INLCOEF[MC - 1][2];
    const int IM = tb.INLCOEF[MC - 1][3], IM1 = tb.INLCOEF[MC - 1][4];
    const T* R = tb.RNLCOEF[MC - 1];
    const T FTAIL = R[0], GW1 = R[1], GW2 = R[2], GW3 = R[3], GW4 = R[4];
    const T FKLAMPA = R[5], FKLAMPB = R[6], FKLAMP2 = R[7], FKLAMP1 = R[8];
    const T FKLAPA2 = R[9], FKLAPB2 = R[10], FKLAP12 = R[11], FKLAP22 = R[12];
    const T GW5 = R[13], GW6 = R[14], GW7 = R[15], GW8 = R[16];
    const T FKLAMMA = R[17], FKLAMMB = R[18], FKLAMM2 = R[19], FKLAMM1 = R[20];
    const T FKLAMA2 = R[21], FKLAMB2 = R[22], FKLAM12 = R[23], FKLAM22 = R[24];
    const T FTEMP = tb.AF11[MC - 1] * ENHFR;
    const int branch = (MC > MFR1STFR && MC < MFRLSTFR) ? 0 : (MC >= MFRLSTFR ? 1 : 2);
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      const int k1 = K1[kh], k2 = K2[kh], k11 = K11[kh], k21 = K21[kh];
      const T SAP = GW1 * sF[IP * NAP + k1] + GW2 * sF[IP * NAP + k11] + GW3 * sF[IP1 * NAP + k1] + GW4 * sF[IP1 * NAP + k11];
      const T SAM = GW5 * sF[IM * NAP + k2] + GW6 * sF[IM * NAP + k21] + GW7 * sF[IM1 * NAP + k2] + GW8 * sF[IM1 * NAP + k21];
      T FIJ = sF[IC * NAP + L.k];
      if (branch != 0) FIJ = FIJ * FTAIL;
      T FAD1 = FIJ * (SAP + SAM);
      const T FAD2 = FAD1 - T(2) * SAP * SAM;
      FAD1 = FAD1 + FAD2;
      const T FCEN = FTEMP * FIJ;
      const T AD = FAD2 * FCEN;
      const T DELAD = FAD1 * FTEMP;
      const T DELAP = (FIJ - T(2) * SAM) * tb.DAL1 * FCEN;
      const T DELAM = (FIJ - T(2) * SAP) * tb.DAL2 * FCEN;
      if (branch == 0) {
        ADDS(L.k, MC, -T(2) * AD); ADDF(L.k, MC, -T(2) * DELAD);
        ADDS(k2, MM, AD * FKLAMM1); ADDF(k2, MM, DELAM * FKLAM12);
        ADDS(k21, MM, AD * FKLAMM2); ADDF(k21, MM, DELAM * FKLAM22);
        ADDS(k2, MM1, AD * FKLAMMA); ADDF(k2, MM1, DELAM * FKLAMA2);
        ADDS(k21, MM1, AD * FKLAMMB); ADDF(k21, MM1, DELAM * FKLAMB2);
        ADDS(k1, MP, AD * FKLAMP1); ADDF(k1, MP, DELAP * FKLAP12);
        ADDS(k11, MP, AD * FKLAMP2); ADDF(k11, MP, DELAP * FKLAP22);
        ADDS(k1, MP1, AD * FKLAMPA); ADDF(k1, MP1, DELAP * FKLAPA2);
        ADDS(k11, MP1, AD * FKLAMPB); ADDF(k11, MP1, DELAP * FKLAPB2);
      } else if (branch == 1) {
        ADDS(k2, MM, AD * FKLAMM1); ADDF(k2, MM, DELAM * FKLAM12);
        ADDS(k21, MM, AD * FKLAMM2); ADDF(k21, MM, DELAM * FKLAM22);
        if (MM1 <= NFRE) {
          ADDS(k2, MM1, AD * FKLAMMA); ADDF(k2, MM1, DELAM * FKLAMA2);
          ADDS(k21, MM1, AD * FKLAMMB); ADDF(k21, MM1, DELAM * FKLAMB2);
          if (MC <= NFRE) {
            ADDS(L.k, MC, -T(2) * AD); ADDF(L.k, MC, -T(2) * DELAD);
            if (MP <= NFRE) {
              ADDS(k1, MP, AD * FKLAMP1); ADDF(k1, MP, DELAP * FKLAP12);
              ADDS(k11, MP, AD * FKLAMP2); ADDF(k11, MP, DELAP * FKLAP22);
              if (MP1 <= NFRE) {
                ADDS(k1, MP1, AD * FKLAMPA); ADDF(k1, MP1, DELAP * FKLAPA2);
                ADDS(k11, MP1, AD * FKLAMPB); ADDF(k11, MP1, DELAP * FKLAPB2);
              }
            }
          }
        }
      } else {
        if (MM1 >= 1) {
          ADDS(k2, MM1, AD * FKLAMMA); ADDF(k2, MM1, DELAM * FKLAMA2);
          ADDS(k21, MM1, AD * FKLAMMB); ADDF(k21, MM1, DELAM * FKLAMB2);
        }
        ADDS(L.k, MC, -T(2) * AD); ADDF(L.k, MC, -T(2) * DELAD);
        ADDS(k1, MP, AD * FKLAMP1); ADDF(k1, MP, DELAP * FKLAP12);
        ADDS(k11, MP, AD * FKLAMP2); ADDF(k11, MP, DELAP * FKLAP22);
        ADDS(k1, MP1, AD * FKLAMPA); ADDF(k1, MP1, DELAP * FKLAPA2);
        ADDS(k11, MP1, AD * FKLAMPB); ADDF(k11, MP1, DELAP * FKLAPB2);
      }
    }
  }
#undef ADDS
#undef ADDF
}

// snonlin.F90:126-494 in "pull" form (used when the DIA tables have their regular structure, DevTab::DIA_PULL):
//   * K1W/K2W are rotations of the direction index and K11W/K21W their +-1 neighbours, so the 4-point gathers
//     SAP/SAM become one lane rotation of a locally combined value, and the scatters into (K1,MP), (K11,MP), ... become
//     pulls of AD/DELAP/DELAM through the inverse rotation (ds_bpermute: crossbar only) plus a one-lane DPP rotate;
//   * the target rows of interaction MC are MC-4, MC-3, MC, MC+2, MC+3, so SL/FLD increments are accumulated in an
//     8-row register ring (compile-time slots via unrolling MC by 8) and each row is added to the LDS tile once,
//     when it leaves the window.  Rows outside 1..NFRE are dropped, which is what the reference's edge branches do.
// LDS instructions per MC: 5 own-column reads of F + 12 bpermutes + 2 RMW, against 9+36 per (MC,KH) in scatter form.
template <typename T>
__device__ void snonlin_pull(const DevTab<T>& tb, const T* sF, T* sFLD, T* sSL, const Lane<T>& L, T DEPTH, T AKMEAN) {
  T ENHFR = m_max(T(0.75) * DEPTH * AKMEAN, T(0.5));
  ENHFR = T(1) + (T(5.5) / ENHFR) * (T(1) - T(.833) * ENHFR) * m_exp(-T(1.25) * ENHFR);
  const int NAP = L.NAP, NFRE = L.NFRE, NANG = L.NANG, k = L.k, lane = L.lane;
  const int MFR1STFR = -tb.MFRSTLW + 1;
  const int MFRLSTFR = NFRE - tb.KFRH + MFR1STFR;
  // every +-1 rotation is folded into a second pull index (ds_bpermute runs on the LDS crossbar, the VALU is the bound here)
  int k1[2], k2[2], k11[2], k21[2], ik1[2], ik2[2], ik1s[2], ik2s[2];
#pragma unroll
  for (int kh = 0; kh < 2; kh++) {
    k1[kh] = tb.K1W[kh][k]; k2[kh] = tb.K2W[kh][k]; k11[kh] = tb.K11W[kh][k]; k21[kh] = tb.K21W[kh][k];
    ik1[kh] = tb.IK1[kh][k]; ik2[kh] = tb.IK2[kh][k];
    // increments sent to K11 (K21) arrive one lane further along D11 (D21): column c takes them from the sender of c-D
    const int c1 = k - tb.D11[kh], c2 = k - tb.D21[kh];
    ik1s[kh] = tb.IK1[kh][c1 < 0 ? c1 + NANG : (c1 >= NANG ? c1 - NANG : c1)];
    ik2s[kh] = tb.IK2[kh][c2 < 0 ? c2 + NANG : (c2 >= NANG ? c2 - NANG : c2)];
  }
  T aS[8], aF[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { aS[i] = T(0); aF[i] = T(0); }
  for (int MCb = 0; MCb < tb.MLSTHG; MCb += 8) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int MC = MCb + 1 + j;
      if (MC <= tb.MLSTHG) {
        constexpr int dummy = 0;
        const int c0 = (1 + j) & 7, cm = (1 + j + 4) & 7, cm1 = (1 + j + 5) & 7, cp = (1 + j + 2) & 7, cp1 = (1 + j + 3) & 7;  // rows MC, MC-4, MC-3, MC+2, MC+3
        const int IC = tb.INLCOEF[MC - 1][0], IP = tb.INLCOEF[MC - 1][1], IP1 = tb.INLCOEF[MC - 1][2];
        const int IM = tb.INLCOEF[MC - 1][3], IM1 = tb.INLCOEF[MC - 1][4];
        const T* R = tb.RNLCOEF[MC - 1];
        const T FTAIL = R[0], GW1 = R[1], GW2 = R[2], GW3 = R[3], GW4 = R[4];
        const T FKLAMPA = R[5], FKLAMPB = R[6], FKLAMP2 = R[7], FKLAMP1 = R[8];
        const T FKLAPA2 = R[9], FKLAPB2 = R[10], FKLAP12 = R[11], FKLAP22 = R[12];
        const T GW5 = R[13], GW6 = R[14], GW7 = R[15], GW8 = R[16];
        const T FKLAMMA = R[17], FKLAMMB = R[18], FKLAMM2 = R[19], FKLAMM1 = R[20];
        const T FKLAMA2 = R[21], FKLAMB2 = R[22], FKLAM12 = R[23], FKLAM22 = R[24];
        const T FTEMP = tb.AF11[MC - 1] * ENHFR;
        const bool mid = (MC > MFR1STFR && MC < MFRLSTFR);
        const T fIP = sF[IP * NAP + k], fIP1 = sF[IP1 * NAP + k], fIM = sF[IM * NAP + k], fIM1 = sF[IM1 * NAP + k];
        T FIJ = sF[IC * NAP + k];
        if (!mid) FIJ = FIJ * FTAIL;
        const T up = GW1 * fIP + GW3 * fIP1, vp = GW2 * fIP + GW4 * fIP1;
        const T um = GW5 * fIM + GW7 * fIM1, vm = GW6 * fIM + GW8 * fIM1;
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
          const T SAP = lane_pull(up, k1[kh]) + lane_pull(vp, k11[kh]);
          const T SAM = lane_pull(um, k2[kh]) + lane_pull(vm, k21[kh]);
          T FAD1 = FIJ * (SAP + SAM);
          const T FAD2 = FAD1 - T(2) * SAP * SAM;
          FAD1 = FAD1 + FAD2;
          const T FCEN = FTEMP * FIJ;
          const T AD = FAD2 * FCEN;
          const T DELAD = FAD1 * FTEMP;
          const T DELAP = (FIJ - T(2) * SAM) * tb.DAL1 * FCEN;
          const T DELAM = (FIJ - T(2) * SAP) * tb.DAL2 * FCEN;
          // increments arriving at column c: from the lane whose K2 (K1) is c, and from the one whose K21 (K11) is c
          const T A2 = lane_pull(AD, ik2[kh]), D2 = lane_pull(DELAM, ik2[kh]);
          const T A1 = lane_pull(AD, ik1[kh]), P1 = lane_pull(DELAP, ik1[kh]);
          const T A2s = lane_pull(AD, ik2s[kh]), D2s = lane_pull(DELAM, ik2s[kh]);
          const T A1s = lane_pull(AD, ik1s[kh]), P1s = lane_pull(DELAP, ik1s[kh]);
          aS[c0] -= T(2) * AD;
          aF[c0] -= T(2) * DELAD;
          aS[cm] += A2 * FKLAMM1 + A2s * FKLAMM2;
          aF[cm] += D2 * FKLAM12 + D2s * FKLAM22;
          aS[cm1] += A2 * FKLAMMA + A2s * FKLAMMB;
          aF[cm1] += D2 * FKLAMA2 + D2s * FKLAMB2;
          aS[cp] += A1 * FKLAMP1 + A1s * FKLAMP2;
          aF[cp] += P1 * FKLAP12 + P1s * FKLAP22;
          aS[cp1] += A1 * FKLAMPA + A1s * FKLAMPB;
          aF[cp1] += P1 * FKLAPA2 + P1s * FKLAPB2;
        }
        (void)dummy;
        const int r = MC - 4;  // this row receives nothing from later interactions
        if (r >= 1 && r <= NFRE && L.act) {
          sSL[(r - 1) * NAP + k] += aS[cm];
          sFLD[(r - 1) * NAP + k] += aF[cm];
        }
        aS[cm] = T(0);
        aF[cm] = T(0);
      }
    }
  }
}

template <typename T, int WPB>
__global__ void __launch_bounds__(64 * WPB) k_implsch(const DevTab<T>* __restrict__ tp, int kijs, int kijl, T* __restrict__ fl1,
                                                      const T* __restrict__ wvprpt, T* __restrict__ ffa, T* __restrict__ intfa,
                                                      int* __restrict__ mij_out, T* __restrict__ xllws, double* __restrict__ /*w2n: variant 2 only*/,
                                                      T* __restrict__ dbg) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const DevTab<T>& tb = *tp;
  const int SKIP = tb.DBG_SKIP;  // 0 in production; timing diagnostics only
  const int wave = threadIdx.x >> 6;
  const int ij = kijs + blockIdx.x * WPB + wave;
  if (ij >= kijl) return;  // wave-uniform; no block-level barrier is used anywhere below
  Lane<T> L;
  L.lane = threadIdx.x & 63;
  L.NANG = tb.NANG; L.NFRE = tb.NFRE; L.NAP = tb.NANG | 1;
  L.act = L.lane < L.NANG; L.actm = L.lane < L.NFRE;
  L.k = L.act ? L.lane : 0;
  {
    const int mi = L.actm ? L.lane : 0;
    L.rDFIM = tb.DFIM[mi]; L.rDFIMOFR = tb.DFIMOFR[mi]; L.rZPIFR = tb.ZPIFR[mi]; L.rCOFRM4 = tb.COFRM4[mi]; L.rFLMAX = tb.FLMAX[mi];
  }
  const int NANG = L.NANG, NFRE = L.NFRE, NAP = L.NAP, N = NANG * NFRE;
  const int tile = NFRE * NAP;
  T* sF = reinterpret_cast<T*>(smem_raw) + (size_t)wave * (3 * tile + 64);
  T* sFLD = sF + tile;
  T* sSL = sFLD + tile;
  T* sScr = sSL + tile;

  // ---- load the spectrum FL1[ij][K][M] (coalesced) into the [M][NAP] tile
  {
    const T* g = fl1 + (size_t)ij * N;
    const float rnf = 1.0f / (float)NFRE;  // e/NFRE by a float multiply: (e+0.5)/NFRE is never within 1e-2 of an integer
    for (int e = L.lane; e < N; e += 64) {
      const int kk = (int)(((float)e + 0.5f) * rnf), mm = e - kk * NFRE;
      sF[mm * NAP + kk] = g[e];
    }
  }
  // per-frequency point properties: lane m holds M=m+1
  T rWAVNUM = T(1), rCINV = T(0), rXK2CG = T(0), rSTOKFAC = T(0);
  {
    const T* wp = wvprpt + (size_t)ij * ECWAM_HIP_NWPR * NFRE;
    if (L.actm) { rWAVNUM = wp[L.lane]; rCINV = wp[2 * NFRE + L.lane]; rXK2CG = wp[3 * NFRE + L.lane]; rSTOKFAC = wp[4 * NFRE + L.lane]; }
  }
  const T ffv = (L.lane < ECWAM_HIP_NFF) ? ffa[(size_t)ij * ECWAM_HIP_NFF + L.lane] : T(0);
  const T AIRD = lane_get(ffv, 0), WDWAVE = lane_get(ffv, 1), CICOVER = lane_get(ffv, 2), WSWAVE = lane_get(ffv, 3);
  const T WSTAR = lane_get(ffv, 4), USTRA = lane_get(ffv, 5), VSTRA = lane_get(ffv, 6);
  T UFRIC = lane_get(ffv, 7), TAUW = lane_get(ffv, 8), TAUWDIR = lane_get(ffv, 9), Z0M = lane_get(ffv, 10);
  T Z0B = lane_get(ffv, 11), CHRNCK = lane_get(ffv, 12);
  const T EMAXDPT = lane_get(ffv, 14), DEPTH = lane_get(ffv, 15);
  WSYNC();

  // ---- implsch.F90:183-203
  const T DELT = T(tb.IDELT);
  const T DELTM = T(1) / DELT;
  const T DELT5 = tb.XIMP * DELT;
  const T RAORW = m_max(AIRD, T(1)) * tb.ROWATERM1;
  const T thk = tb.TH[L.k];
  const T coswdif = m_cos(thk - WDWAVE);
  T sinwdif2 = m_sin(thk - WDWAVE);
  sinwdif2 = sinwdif2 * sinwdif2;

  // ---- SDEPTHLIM (sdepthlim.F90:64-78, semean.F90:82-120)
  if (tb.LBIWBK) {
    const T t2 = colsum(sF, L);
    T EM = tb.EPSMIN + usum(L.actm ? tb.DFIM[L.lane] * t2 : T(0));
    EM = EM + (tb.WETAIL * tb.FR[NFRE - 1] * tb.DELTH) * lane_get(t2, NFRE - 1);
    const T s = m_min(EMAXDPT / EM, T(1));
    WSYNC();
    if (L.act)
      for (int m = 0; m < NFRE; m++) sF[m * NAP + L.k] = m_max(sF[m * NAP + L.k] * s, tb.EPSMIN);
    WSYNC();
  }
  // ---- FKMEAN
  T EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN;
  fkmean(tb, sF, L, rWAVNUM, EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN);
  const T cpos = m_max(T(0), coswdif);
  const T FLM = (T(1) - T(0.9) * m_min(CICOVER, T(0.99))) * tb.FLMIN * (cpos * cpos);

  // ---- 2x SINFLX (sinflx.F90:105-183)
  T FMEANWS = T(0), PHIWA = T(0), EMW;
  int MIJ = NFRE;
  T rRH = T(0);  // RHOWGDFTH, lane m
  unsigned long long xmask = 0ull;
  T RNFAC = T(1);
  if (tb.LLNORMAGAM && tb.LLCAPCHNK) RNFAC = T(1) + tb.DTHRN_A * (T(1) + m_tanh(WSWAVE - tb.DTHRN_U));
  for (int ICALL = 1; ICALL <= 2; ICALL++) {
    const int IUSFG = (ICALL == 1) ? 0 : 1;
    if (ICALL == 1) {
      if (L.act) sF[(NFRE - 1) * NAP + L.k] = m_max(sF[(NFRE - 1) * NAP + L.k], FLM);
      WSYNC();
    }
    if (!(SKIP & 16)) taut_z0_a(tb, IUSFG, WSWAVE, WDWAVE, TAUW, TAUWDIR, UFRIC, Z0M, Z0B, CHRNCK);
    if (SKIP & 1) {
    } else if (ICALL == 1)
      sinput_ard<T, 1, false>(tb, sF, sFLD, sSL, L, rWAVNUM, rCINV, rXK2CG, WDWAVE, WSWAVE, UFRIC, Z0M, coswdif, sinwdif2, RAORW,
                              WSTAR, RNFAC, xmask);
    else
      sinput_ard<T, 2, true>(tb, sF, sFLD, sSL, L, rWAVNUM, rCINV, rXK2CG, WDWAVE, WSWAVE, UFRIC, Z0M, coswdif, sinwdif2, RAORW,
                             WSTAR, RNFAC, xmask);
    WSYNC();
    femeanws(tb, sF, L, xmask, FMEANWS, EMW);
    // FRCUTINDEX (frcutindex.F90:84-108)
    {
      const T FPMH = tb.TAILFACTOR / tb.FR[0];
      const T FPPM = tb.TAILFACTOR_PM * tb.G / (tb.FRIC * tb.ZPIFR[0]);
      if (CICOVER <= tb.CITHRSH_TAIL) {
        T FM2 = m_max(FMEANWS, FMEAN) * FPMH;
        T FPM = FPPM / m_max(UFRIC, tb.EPSMIN);
        T FPM4 = m_max(FM2, FPM);
        MIJ = m_nint(m_log10(FPM4) * tb.FLOGSPRDM1) + 1;
        MIJ = MIJ < 1 ? 1 : (MIJ > NFRE ? NFRE : MIJ);
      } else MIJ = NFRE;
      MIJ = __builtin_amdgcn_readfirstlane(MIJ);
      rRH = T(0);
      if (L.actm && L.lane + 1 <= MIJ) {
        rRH = tb.RHOWG_DFIM[L.lane];
        if (L.lane + 1 == MIJ && MIJ != NFRE) rRH = T(0.5) * rRH;
      }
    }
    // STRESSO (stresso.F90:125-229)
    if (!(SKIP & 2)) {
      const bool LLPHIWA = (ICALL == 2);
      T ax = T(0), at = T(0), ap = T(0);
      for (int m = 0; m < NFRE; m++) {
        const T spos = sSL[m * NAP + L.k];
        const T rh = lane_get(rRH, m);
        ax += (rh * lane_get(rCINV, m)) * spos;
        if (LLPHIWA) {
          at += rh * spos;
          ap += (sFLD[m * NAP + L.k] * sF[m * NAP + L.k] - spos) * tb.RHOWG_DFIM[m];
        }
      }
      if (!L.act) { ax = T(0); at = T(0); ap = T(0); }
      T XSTRESS = usum(ax * tb.SINTH[L.k]);
      T YSTRESS = usum(ax * tb.COSTH[L.k]);
      PHIWA = T(0);
      if (LLPHIWA) PHIWA = usum(ap) + usum(at);
      XSTRESS = XSTRESS / m_max(AIRD, T(1));
      YSTRESS = YSTRESS / m_max(AIRD, T(1));
      bool LTAUWSHELTER;
      T USDIRP, UST;
      if (tb.TAUWSHELTER == T(0)) { LTAUWSHELTER = false; USDIRP = WDWAVE; UST = UFRIC; }
      else {
        LTAUWSHELTER = true;
        T TAUPX = UFRIC * UFRIC * m_sin(WDWAVE) - tb.TAUWSHELTER * XSTRESS;
        T TAUPY = UFRIC * UFRIC * m_cos(WDWAVE) - tb.TAUWSHELTER * YSTRESS;
        USDIRP = m_atan2(TAUPX, TAUPY);
        UST = m_pow(TAUPX * TAUPX + TAUPY * TAUPY, T(0.25));
      }
      T TAUHF, PHIHF;
      tau_phi_hf(tb, sF, L, MIJ, LTAUWSHELTER, Z0M, AIRD, RNFAC, coswdif, sinwdif2, UST, TAUHF, PHIHF, LLPHIWA);
      XSTRESS = XSTRESS + TAUHF * m_sin(USDIRP);
      YSTRESS = YSTRESS + TAUHF * m_cos(USDIRP);
      TAUW = m_max(m_sqrt(XSTRESS * XSTRESS + YSTRESS * YSTRESS), T(0));
      TAUWDIR = m_atan2(XSTRESS, YSTRESS);
      if (!tb.LLGCBZ0) TAUW = m_min(TAUW, UFRIC * UFRIC * (T(1) / (T(1) + tb.EPS1)));
      if (LLPHIWA) PHIWA = PHIWA + PHIHF;
    }
  }
  if (dbg && L.lane == 0) {
    T* d = dbg + (size_t)ij * 32;
    d[0] = EMEAN; d[1] = FMEAN; d[2] = F1MEAN; d[3] = AKMEAN; d[4] = XKMEAN; d[5] = FMEANWS; d[6] = PHIWA;
  }

  // ---- SDISSIP (also rebuilds SL = FLD*F from the wind input)
  if (SKIP & 4) {
  } else if (tb.NTAP == 17) sdissip_ard<T, 17>(tb, sF, sFLD, sSL, L, rWAVNUM, rXK2CG, UFRIC, coswdif, RAORW);
  else if (tb.NTAP == 11) sdissip_ard<T, 11>(tb, sF, sFLD, sSL, L, rWAVNUM, rXK2CG, UFRIC, coswdif, RAORW);
  else if (tb.NTAP == 7) sdissip_ard<T, 7>(tb, sF, sFLD, sSL, L, rWAVNUM, rXK2CG, UFRIC, coswdif, RAORW);
  else sdissip_ard<T, 0>(tb, sF, sFLD, sSL, L, rWAVNUM, rXK2CG, UFRIC, coswdif, RAORW);
  WSYNC();
  // ---- SNONLIN
  if (SKIP & 8) {
  } else if (tb.DIA_PULL) snonlin_pull(tb, sF, sFLD, sSL, L, DEPTH, AKMEAN);
  else snonlin(tb, sF, sFLD, sSL, L, DEPTH, AKMEAN);
  WSYNC();

  // ---- SSOURCE, SDIWBK, SBOTTOM, new spectrum, WNFLUXES integrals: one pass (implsch.F90:294-395)
  T SDS = T(0);
  const bool shallow_brk = tb.LBIWBK && (DEPTH < T(50.0));
  if (shallow_brk) {  // sdiwbk.F90:88-103
    const T ALPH = T(2) * EMAXDPT / EMEAN;
    const T ARG = m_min(ALPH, T(50));
    T Q_OLD = m_exp(-ARG), Q = T(0);
    for (int ic = 0; ic < 15; ic++) {
      T EXPQ = m_exp(-ARG * (T(1) - Q_OLD));
      Q = Q_OLD - (EXPQ - Q_OLD) / (ARG * EXPQ - T(1));
      T REL_ERR = m_abs(Q - Q_OLD) / Q_OLD;
      if (REL_ERR < T(0.00001)) break;
      Q_OLD = Q;
    }
    Q = m_min(Q, T(1));
    SDS = T(2) * ALPH * Q * F1MEAN;
  }
  T rSBO = T(0);  // sbottom.F90:79-89, lane m
  if (L.actm && L.lane < tb.NFRE_RED && DEPTH < tb.BATHYMAX) {
    T ARG = m_min(T(2) * DEPTH * rWAVNUM, T(50));
    rSBO = (-T(2) * T(0.038) * tb.GM1) * rWAVNUM / m_sinh(ARG);
  }
  const T USFM = UFRIC * m_max(FMEANWS, FMEAN);
  T a_t = T(0), a_x = T(0);
  for (int m = (SKIP & 32) ? NFRE : 0; m < NFRE; m++) {
    T fld = sFLD[m * NAP + L.k], sl = sSL[m * NAP + L.k];
    const T f = sF[m * NAP + L.k];
    T ss = T(0);
    if (tb.LCFLX && tb.LWVFLX_SNL) ss = f_div(sl, m_max(T(1) - DELT5 * fld, T(1)));
    if (shallow_brk && m < tb.NFRE_RED) { sl = sl - SDS * f; fld = fld - SDS; }
    const T sbo = lane_get(rSBO, m);
    if (m < tb.NFRE_RED) { sl = sl + sbo * f; fld = fld + sbo; }
    const T GTEMP1 = m_max(T(1) - DELT5 * fld, T(1));
    const T GTEMP2 = f_div(DELT * sl, GTEMP1);
    const T FLHAB = m_min(m_abs(GTEMP2), USFM * (lane_get(L.rCOFRM4, m) * DELT));
    T fn = f + m_sign(FLHAB, GTEMP2);
    fn = m_max(fn, FLM);
    const T flmax = lane_get(L.rFLMAX, m);
    ss = ss + DELTM * m_min(flmax - fn, T(0));
    fn = m_min(fn, flmax);
    if (L.act) sF[m * NAP + L.k] = fn;
    const T rh = lane_get(rRH, m);
    a_t += rh * ss;
    a_x += (lane_get(rCINV, m) * rh) * ss;
  }
  WSYNC();

  // ---- WNFLUXES (wnfluxes.F90:147-330), LWNEMOCOUWRS = F
  T TAUXD = T(0), TAUYD = T(0), TAUOCXD = T(0), TAUOCYD = T(0), TAUOC = T(0), PHIOCD = T(0), PHIEPS = T(0), PHIAW = T(0);
  if (tb.LCFLX) {
    if (!L.act) { a_t = T(0); a_x = T(0); }
    const T PHILF = usum(a_t);
    const T XSTRESS = usum(a_x * tb.SINTH[L.k]);
    const T YSTRESS = usum(a_x * tb.COSTH[L.k]);
    const T EPSUS3 = tb.EPSUS * m_sqrt(tb.EPSUS);
    const T ZCITHRS = tb.CIBLOCK;
    const T CITHRSH_INV = T(1) / m_max(tb.CITHRSH, T(0.01));
    const T ZMAXEXP = T(10);
    T OOVAL = T(1), USTAR = UFRIC;
    if (tb.LICERUN && tb.LWAMRSETCI && CICOVER > ZCITHRS) {
      OOVAL = m_exp(-m_min(m_pow4(CICOVER * CITHRSH_INV), ZMAXEXP));
      const T U10P = m_max(WSWAVE, tb.EPSU10);
      const T CD_BULK = m_min((T(1.03E-3) + T(0.04E-3) * m_pow(U10P, T(1.48))) * m_pow(U10P, T(-0.21)), T(0.003));
      const T CD_WAVE = (UFRIC / U10P) * (UFRIC / U10P);
      const T CD_ICE = OOVAL * CD_WAVE + (T(1) - OOVAL) * CD_BULK;
      USTAR = m_max(m_sqrt(CD_ICE) * U10P, tb.EPSUS);
    }
    const T TAU = AIRD * m_max(USTAR * USTAR, tb.EPSUS);
    TAUXD = TAU * m_sin(WDWAVE);
    TAUYD = TAU * m_cos(WDWAVE);
    TAUOCXD = TAUXD - OOVAL * XSTRESS;
    TAUOCYD = TAUYD - OOVAL * YSTRESS;
    const T TAUO = m_sqrt(TAUOCXD * TAUOCXD + TAUOCYD * TAUOCYD);
    TAUOC = m_min(m_max(TAUO / TAU, tb.TAUOCMIN), tb.TAUOCMAX);
    if (tb.LWCOUAST && (USTRA != T(0) || VSTRA != T(0))) {
      TAUXD = USTRA; TAUOCXD = USTRA * TAUOC; TAUYD = VSTRA; TAUOCYD = VSTRA * TAUOC;
    }
    const T XN = AIRD * m_max(USTAR * USTAR * USTAR, EPSUS3);
    PHIOCD = OOVAL * (PHILF - PHIWA) + (T(1) - OOVAL) * T(-3.75) * XN;
    PHIEPS = m_min(m_max(PHIOCD / XN, tb.PHIEPSMIN), tb.PHIEPSMAX);
    PHIOCD = PHIEPS * XN;
    PHIAW = OOVAL * PHIWA / XN + (T(1) - OOVAL) * T(3.75);
  }

  // ---- second FKMEAN / FEMEANWS, IMPHFTAIL, SETICE, STOKESDRIFT (implsch.F90:422-462)
  fkmean(tb, sF, L, rWAVNUM, EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN);
  T EMEANWS;
  femeanws(tb, sF, L, xmask, FMEANWS, EMEANWS);
  {
    // imphftail.F90:73-87: lane m holds TEMP2(M) = (1/XK2CG(M)/WAVNUM(M)) / TEMP1
    T rT = T(1) / rXK2CG / rWAVNUM;
    const T TEMP1 = lane_get(rT, MIJ - 1);
    rT = rT / TEMP1;
    const T tf = sF[(MIJ - 1) * NAP + L.k];
    for (int m = MIJ; m < NFRE; m++) {
      const T tm = lane_get(rT, m);
      if (L.act) sF[m * NAP + L.k] = m_max(tm * tf, FLM);
    }
  }
  if (tb.LICERUN && tb.LMASKICE) {  // setice.F90:67-86
    T CIREDUC, ICEFREE;
    if (CICOVER > tb.CITHRSH) { CIREDUC = m_max(tb.EPSMIN, T(1) - CICOVER); ICEFREE = T(0); }
    else { CIREDUC = T(0); ICEFREE = T(1); }
    const T add = (CIREDUC * tb.FLMIN) * (cpos * cpos);
    if (L.act)
      for (int m = 0; m < NFRE; m++) sF[m * NAP + L.k] = sF[m * NAP + L.k] * ICEFREE + add;
  }
  T USTOKES, VSTOKES;
  {  // stokesdrift.F90:89-142
    const int MO = tb.NFRE_ODD;
    const T fo = tb.FR[MO - 1];
    const T CONST = T(2) * tb.DELTH * (tb.ZPI * tb.ZPI * tb.ZPI) / tb.G * m_pow4(fo);
    T a = T(0);
    for (int m = 0; m < MO; m++) a += (lane_get(rSTOKFAC, m) * tb.DFIM_SIM[m]) * sF[m * NAP + L.k];
    a += CONST * sF[(MO - 1) * NAP + L.k];
    if (!L.act) a = T(0);
    USTOKES = usum(a * tb.SINTH[L.k]);
    VSTOKES = usum(a * tb.COSTH[L.k]);
    if (tb.LICERUN && tb.LWAMRSETCI && CICOVER > tb.CITHRSH) {
      USTOKES = T(0.016) * WSWAVE * m_sin(WDWAVE) * (T(1) - CICOVER);
      VSTOKES = T(0.016) * WSWAVE * m_cos(WDWAVE) * (T(1) - CICOVER);
    }
    USTOKES = m_min(m_max(USTOKES, T(-1.5)), T(1.5));
    VSTOKES = m_min(m_max(VSTOKES, T(-1.5)), T(1.5));
  }
  // XLLWS as reals into the SL tile for the coalesced store
  if (L.act)
    for (int m = 0; m < NFRE; m++) sSL[m * NAP + L.k] = ((xmask >> m) & 1ull) ? T(1) : T(0);
  WSYNC();

  // ---- store FL1, XLLWS (coalesced) and the per-point scalars
  {
    T* g = fl1 + (size_t)ij * N;
    T* gx = xllws + (size_t)ij * N;
    const float rnf = 1.0f / (float)NFRE;
    for (int e = L.lane; e < N; e += 64) {
      const int kk = (int)(((float)e + 0.5f) * rnf), mm = e - kk * NFRE;
      g[e] = sF[mm * NAP + kk];
      gx[e] = sSL[mm * NAP + kk];
    }
  }
  if (L.lane == 0) {
    sScr[7] = UFRIC; sScr[8] = TAUW; sScr[9] = TAUWDIR; sScr[10] = Z0M; sScr[11] = Z0B; sScr[12] = CHRNCK;
    sScr[16 + 2] = USTOKES; sScr[16 + 3] = VSTOKES;
    sScr[16 + 5] = TAUXD; sScr[16 + 6] = TAUYD; sScr[16 + 7] = TAUOCXD; sScr[16 + 8] = TAUOCYD; sScr[16 + 9] = TAUOC;
    sScr[16 + 10] = T(0); sScr[16 + 11] = T(0); sScr[16 + 12] = PHIOCD; sScr[16 + 13] = PHIEPS; sScr[16 + 14] = PHIAW;
    mij_out[ij] = MIJ;
  }
  WSYNC();
  if (L.lane >= 7 && L.lane <= 12) ffa[(size_t)ij * ECWAM_HIP_NFF + L.lane] = sScr[L.lane];
  if (L.lane < ECWAM_HIP_NINTF) {
    const int i = L.lane;
    const bool fluxes = tb.LCFLX && (i >= 5 && i <= 14);
    if (i == 2 || i == 3 || fluxes) intfa[(size_t)ij * ECWAM_HIP_NINTF + i] = sScr[16 + i];
    if (tb.LWFLUX && (i == 0 || i == 1)) {
      T v = (i == 0) ? ((EMEANWS < tb.WSEMEAN_MIN) ? tb.WSEMEAN_MIN : EMEANWS)
                     : ((EMEANWS < tb.WSEMEAN_MIN) ? T(2) * tb.FR[NFRE - 1] : FMEANWS);
      intfa[(size_t)ij * ECWAM_HIP_NINTF + i] = v;
    }
  }
}

