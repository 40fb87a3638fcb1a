// TEST INFRASTRUCTURE: wave-level helpers of k_implsch2 (lane = direction): the lane context, column sums, FKMEAN, FEMEANWS.
#pragma once
#include "implsch_common.h"

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

