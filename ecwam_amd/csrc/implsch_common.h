// Helpers shared by the IMPLSCH kernels (implsch_v2.h: one sea point per wavefront, lane = direction; implsch_v4.h: several points per
// wavefront on adjacent direction pairs): the wavefront fence, the lane bookkeeping of the one-point layout, FKMEAN / FEMEANWS in
// that layout, and the lane-per-point scalar routines TAUT_Z0 (LLGCBZ0 = F) and WSIGSTAR.
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
