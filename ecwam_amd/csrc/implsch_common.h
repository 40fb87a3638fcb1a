// Helpers shared by the IMPLSCH kernels (implsch_v4.h: several points per wavefront on adjacent direction pairs; tests/csrc/implsch_v2.h: the
// tests' one-point-per-wavefront second implementation): the wavefront fence and the lane-per-point scalar routines CHNKMIN, TAUT_Z0
// (LLGCBZ0 = F) and WSIGSTAR.
#pragma once
#include "dev.h"

#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
  } while (0)

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

