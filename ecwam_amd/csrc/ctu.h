// CTU weight arithmetic (ctuw.F90, IREFRA = 0) and the PROPAGS2 stencil (propags2.F90:107-116) for one element or one pair of
// frequencies: shared by the advection kernels of propag.hip and by the advecting tile load of k_implsch4 (implsch_v4.h, ADV builds),
// floating-point contraction off in every helper so that all of them round identically.
#pragma once
#include "dev.h"

#ifndef ECWAM_HIP_CTU_STRICT
#define ECWAM_HIP_CTU_STRICT 0   // 1: the on-the-fly weights in ctuw.F90's order of operations (bit-identical to the stored-weight scheme)
#endif

// propags2.F90:107-116 for one element, in the reference's association order; contraction off so that every kernel that
// applies the stencil (stored weights, vectorised, on-the-fly weights) rounds identically
template <typename T>
__device__ __forceinline__ T ctu_stencil(T w0, T w1, T w2, T w3, T w4, T w5, T w6, T w7, T f0, T f1, T f2, T f3, T f4, T f5, T f6,
                                         T f7) {
#pragma clang fp contract(off)
  T r = (T(1) - w0) * f0;
  r = r + w1 * f1;
  r = r + w2 * f2;
  r = r + w3 * f3;
  r = r + w4 * f4;
  r = r + w5 * f5;
  r = r + w6 * f6;
  r = r + w7 * f7;
  return r;
}

// ctuw.F90:146-275 (space weights), :407-484 (great-circle refraction WKPMN), :536-608 (range checks, SUMWN)
// restricted to what PROPAGS2 reads for IREFRA=0 (ISSU=ISSV=1 => DXDW=DYDW=0).
//
// The arithmetic lives in two helpers shared by k_ctuw (weights stored once, the reference's scheme) and k_propags2_otf
// (weights rebuilt inside the stencil): floating-point contraction is switched off in them so that both kernels produce
// the same bits whatever the surrounding code looks like.
template <typename T>
struct CtuBase {  // direction-independent part for one (point, frequency)
  T h[2];   // 0.5*(CG(IJ)+CG(KLON(IC)))
  T hy[2];  // 0.5*(CG(IJ)+DP(IC)*CGYP(IC))
  T cg0;
};
template <typename T>
__device__ __forceinline__ CtuBase<T> ctu_base(T cg0, const T cgl[2], const T cgy0[2], const T cgy1[2], const T wl[2], const T dp[2]) {
#pragma clang fp contract(off)
  CtuBase<T> b;
  b.cg0 = cg0;
  for (int ic = 0; ic < 2; ic++) {
    b.h[ic] = T(0.5) * (cg0 + cgl[ic]);
    const T cgyp = wl[ic] * cgy0[ic] + (T(1) - wl[ic]) * cgy1[ic];
    b.hy[ic] = T(0.5) * (cg0 + dp[ic] * cgyp);
  }
  return b;
}
// w8 = SUMWN, WLONN(JXO(K,1)), WLATN(JYO(K,1),1:2), WCORN(1,1:2), WKPMN(-1), WKPMN(+1); returns the CFL / range failure flag
template <typename T>
__device__ __forceinline__ bool ctu_w8(const CtuBase<T>& b, T sink, T cosk, T cpm1, T zd, T xdella, T ga, T delpro, T cmtodeg,
                                       int jx0, int jx1, int jy0, int jy1, T wl_jy0, T wc_kc, T tsp, T tsm, T* w8) {
#pragma clang fp contract(off)
  T adxp[2], adyp[2];
  bool fail = false;
  for (int ic = 0; ic < 2; ic++) {
    const T cgx = b.h[ic] * sink * cpm1;
    const T cgy = b.hy[ic] * cosk;
    adxp[ic] = m_abs(-delpro * cgx * cmtodeg);
    adyp[ic] = m_abs(-delpro * cgy * cmtodeg);
    if (adxp[ic] > zd || adyp[ic] > xdella) fail = true;
  }
  const T dxx = zd - adxp[jx1];
  const T dyy = xdella - adyp[jy1];
  const T wgt_lat = dxx * adyp[jy0] * ga;  // WEIGHT(JYO(K,1))
  const T wlatn1 = wl_jy0 * wgt_lat;
  const T wlatn2 = (T(1) - wl_jy0) * wgt_lat;
  const T wlonn = dyy * adxp[jx0] * ga;
  const T wgt_cor = adxp[jx0] * adyp[jy0] * ga;  // WEIGHT(1)
  const T wcorn1 = wc_kc * wgt_cor;
  const T wcorn2 = (T(1) - wc_kc) * wgt_cor;
  T sumwn = (zd * adyp[jy1] + xdella * adxp[jx1] - adxp[jx1] * adyp[jy1]) * ga;
  const T dthp = tsp * b.cg0;  // TANPH*SP*CG
  const T dthm = tsm * b.cg0;
  const T wk0 = (dthp + m_abs(dthp)) + (m_abs(dthm) - dthm);
  const T wkp = -dthp + m_abs(dthp);
  const T wkm = dthm + m_abs(dthm);
  sumwn = sumwn + wk0;
  const T one = T(1), zero = T(0);
  if (wlatn1 > one || wlatn1 < zero || wlatn2 > one || wlatn2 < zero || wlonn > one || wlonn < zero || wcorn1 > one ||
      wcorn1 < zero || wcorn2 > one || wcorn2 < zero || wk0 > one || wk0 < zero || wkp > one || wkp < zero || wkm > one ||
      wkm < zero || sumwn > one || sumwn < zero)
    fail = true;
  w8[0] = sumwn; w8[1] = wlonn; w8[2] = wlatn1; w8[3] = wlatn2; w8[4] = wcorn1; w8[5] = wcorn2; w8[6] = wkm; w8[7] = wkp;
  return fail;
}
// ctu_w8 + ctu_stencil on TWO frequencies at a time as packed-fp32 operands (v_pk_mul_f32 / v_pk_add_f32): the same operations
// in the same order per component, contraction off, hence the same bits as the scalar helpers -- at half the instruction issue
// (the advection kernel spends 2/3 of its time issuing vector instructions).  No checks here: k_ctuw does them once.
typedef float F2 __attribute__((ext_vector_type(2)));
template <typename T>
using CtuV2 = T __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ CtuV2<T> v_abs2(CtuV2<T> x) { CtuV2<T> r = {m_abs(x.x), m_abs(x.y)}; return r; }
// (double precision: the same source as plain operations on the two components)
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_w8_stencil_pk(CtuV2<T> h0, CtuV2<T> h1, CtuV2<T> hy0, CtuV2<T> hy1, CtuV2<T> cg0, T sink, T cosk, T cpm1, T zd,
                                                      T xdella, T ga, CtuV2<T> delpro, T cmtodeg, int jx0, int jy0, T wl_jy0,
                                                      T wc_kc, CtuV2<T> tsp, CtuV2<T> tsm, CtuV2<T> f0, CtuV2<T> f1, CtuV2<T> f2, CtuV2<T> f3, CtuV2<T> f4,
                                                      CtuV2<T> f5, CtuV2<T> f6, CtuV2<T> f7) {
#pragma clang fp contract(off)
  typedef CtuV2<T> F;
  const F adx0 = v_abs2<T>(-delpro * (h0 * sink * cpm1) * cmtodeg), adx1 = v_abs2<T>(-delpro * (h1 * sink * cpm1) * cmtodeg);
  const F ady0 = v_abs2<T>(-delpro * (hy0 * cosk) * cmtodeg), ady1 = v_abs2<T>(-delpro * (hy1 * cosk) * cmtodeg);
  const F adx_a = jx0 ? adx1 : adx0, adx_b = jx0 ? adx0 : adx1;   // ADXP(JXO(K,1)), ADXP(JXO(K,2))
  const F ady_a = jy0 ? ady1 : ady0, ady_b = jy0 ? ady0 : ady1;
  const F dxx = zd - adx_b;
  const F dyy = xdella - ady_b;
  const F wgt_lat = dxx * ady_a * ga;
  const F wlatn1 = wl_jy0 * wgt_lat;
  const F wlatn2 = (T(1) - wl_jy0) * wgt_lat;
  const F wlonn = dyy * adx_a * ga;
  const F wgt_cor = adx_a * ady_a * ga;
  const F wcorn1 = wc_kc * wgt_cor;
  const F wcorn2 = (T(1) - wc_kc) * wgt_cor;
  F sumwn = (zd * ady_b + xdella * adx_b - adx_b * ady_b) * ga;
  const F dthp = tsp * cg0;
  const F dthm = tsm * cg0;
  const F wk0 = (dthp + v_abs2<T>(dthp)) + (v_abs2<T>(dthm) - dthm);
  const F wkp = -dthp + v_abs2<T>(dthp);
  const F wkm = dthm + v_abs2<T>(dthm);
  sumwn = sumwn + wk0;
  F r = (T(1) - sumwn) * f0;
  r = r + wlonn * f1;
  r = r + wlatn1 * f2;
  r = r + wlatn2 * f3;
  r = r + wcorn1 * f4;
  r = r + wcorn2 * f5;
  r = r + wkm * f6;
  r = r + wkp * f7;
  return r;
}
// The same weights and stencil with every absolute value and every selection taken out of the per-element code -- for a caller that
// prepares, per (point, frequency), |h| and |hy| already ordered by the direction's quadrant (ha = |h(JXO(K,1))|, hb = |h(JXO(K,2))|, the same
// for hy with JYO) and, per (point, direction), the four non-negative numbers a2 = MAX(2 TSP, 0), p2 = MAX(-2 TSP, 0), b2 = MAX(-2 TSM, 0),
// m2 = MAX(2 TSM, 0) with TSP = TANPH SP, TSM = TANPH SM (the advecting tile load of k_implsch4, implsch_v4.h).  Same bits as
// ctu_w8_stencil_pk, because
//   * rounding is symmetric in the sign: |fl(a b)| = fl(|a| |b|), so ABS(-DELPRO CGX CMTODEG) is the same product of magnitudes
//     (DELPRO, CMTODEG > 0; asink = |SINTH(K)|, acosk = |COSTH(K)|, acpm1 = |COSPHM1|);
//   * doubling is exact: with x = fl(TSP CG) and CG >= 0 (a group velocity), x + |x| is fl((2 TSP) CG) where TSP >= 0 and 0 otherwise,
//     and -x + |x|, |y| - y, y + |y| likewise: WKPMN(0) = a2 CG + b2 CG, WKPMN(1) = p2 CG, WKPMN(-1) = m2 CG.
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_w8_stencil_abs(CtuV2<T> ha, CtuV2<T> hb, CtuV2<T> hya, CtuV2<T> hyb, CtuV2<T> cg0, T asink, T acosk, T acpm1,
                                                       T zd, T xdella, T ga, T delpro, T cmtodeg, T wl, T omwl, T wc, T omwc, T a2, T b2, T p2, T m2,
                                                       CtuV2<T> f0, CtuV2<T> f1, CtuV2<T> f2, CtuV2<T> f3, CtuV2<T> f4, CtuV2<T> f5, CtuV2<T> f6,
                                                       CtuV2<T> f7) {
#pragma clang fp contract(off)
  typedef CtuV2<T> F;
  const F adx_a = delpro * (ha * asink * acpm1) * cmtodeg, adx_b = delpro * (hb * asink * acpm1) * cmtodeg;
  const F ady_a = delpro * (hya * acosk) * cmtodeg, ady_b = delpro * (hyb * acosk) * cmtodeg;
  const F dxx = zd - adx_b;
  const F dyy = xdella - ady_b;
  const F wgt_lat = dxx * ady_a * ga;
  const F wlatn1 = wl * wgt_lat;
  const F wlatn2 = omwl * wgt_lat;
  const F wlonn = dyy * adx_a * ga;
  const F wgt_cor = adx_a * ady_a * ga;
  const F wcorn1 = wc * wgt_cor;
  const F wcorn2 = omwc * wgt_cor;
  F sumwn = (zd * ady_b + xdella * adx_b - adx_b * ady_b) * ga;
  const F wk0 = a2 * cg0 + b2 * cg0;
  const F wkp = p2 * cg0;
  const F wkm = m2 * cg0;
  sumwn = sumwn + wk0;
  F r = (T(1) - sumwn) * f0;
  r = r + wlonn * f1;
  r = r + wlatn1 * f2;
  r = r + wlatn2 * f3;
  r = r + wcorn1 * f4;
  r = r + wcorn2 * f5;
  r = r + wkm * f6;
  r = r + wkp * f7;
  return r;
}
// The weights with every factor that does not depend on all three of (point, frequency, direction) hoisted out of the per-element code, and
// the fused multiply-adds the expressions offer (round 6; the product form of the on-the-fly weights).  Per (point, frequency) the caller
// prepares xa = |h| COSPHM1 DELPRO CMTODEG and ya = |hy| DELPRO CMTODEG for both neighbours (ctu_fast_planes), ordered by the direction's
// quadrant; per (point, direction) ab2 = MAX(2 TSP, 0) + MAX(-2 TSM, 0), p2 = MAX(-2 TSP, 0), m2 = MAX(2 TSM, 0); per point zdg = ZDELLO GA and
// xdg = XDELLA GA.  ABS(-DELPRO CGX CMTODEG) is then ONE product xa |SINTH(K)| (four in ctuw.F90's order), and so on: 30 packed operations
// per pair of frequencies instead of 54.  Every weight goes through fewer roundings than in ctuw.F90's order of operations; against the
// stored-weight scheme (k_ctuw + k_propags2, the reference's order: the bit-identity partner) the weights differ by a few units in the last
// place (tests: <= 4 eps of the weight's scale 1).  -DECWAM_HIP_CTU_STRICT=1 builds the kernels with ctuw.F90's order instead.
template <typename T>
__device__ __forceinline__ void ctu_fast_planes(const CtuBase<T>& b, T acpm1, T dc, T* xa0, T* xa1, T* ya0, T* ya1) {   // dc = DELPRO CMTODEG
#pragma clang fp contract(off)
  *xa0 = (m_abs(b.h[0]) * acpm1) * dc; *xa1 = (m_abs(b.h[1]) * acpm1) * dc;
  *ya0 = m_abs(b.hy[0]) * dc; *ya1 = m_abs(b.hy[1]) * dc;
}
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_v2fma(CtuV2<T> a, CtuV2<T> b, CtuV2<T> c) {
  CtuV2<T> r;
  if constexpr (sizeof(T) == 4) r = __builtin_elementwise_fma(a, b, c);
  else { r.x = __builtin_fma(a.x, b.x, c.x); r.y = __builtin_fma(a.y, b.y, c.y); }
  return r;
}
template <typename T>
struct CtuFastW8 {      // SUMWN, WLONN(JXO(K,1)), WLATN(JYO(K,1),1:2), WCORN(1,1:2), WKPMN(-1), WKPMN(+1) for a pair of frequencies
  CtuV2<T> sumwn, wlon, wlat1, wlat2, wcor1, wcor2, wkm, wkp;
};
template <typename T>
__device__ __forceinline__ CtuFastW8<T> ctu_fast_w8(CtuV2<T> xa, CtuV2<T> xb, CtuV2<T> ya, CtuV2<T> yb, CtuV2<T> cg0, T asink, T acosk, T zd, T xdella, T ga,
                                                    T zdg, T xdg, T wl, T omwl, T wc, T omwc, CtuV2<T> ab2, CtuV2<T> p2, CtuV2<T> m2) {
#pragma clang fp contract(off)      // the fused multiply-adds are the ones written out: the same bits in every translation unit
  typedef CtuV2<T> F;
  const F adx_a = xa * asink, adx_b = xb * asink, ady_a = ya * acosk, ady_b = yb * acosk;
  const F dxx = zd - adx_b, dyy = xdella - ady_b;
  const F yag = ady_a * ga, xag = adx_a * ga, xbg = adx_b * ga;
  const F wgt_lat = dxx * yag;
  const F wgt_cor = adx_a * yag;
  CtuFastW8<T> w;
  w.wlon = dyy * xag;
  F sumwn = ab2 * cg0;                                                  // WKPMN(0)
  sumwn = ctu_v2fma<T>(F{zdg, zdg}, ady_b, sumwn);
  sumwn = ctu_v2fma<T>(F{xdg, xdg}, adx_b, sumwn);
  w.sumwn = ctu_v2fma<T>(-xbg, ady_b, sumwn);
  w.wlat1 = wl * wgt_lat; w.wlat2 = omwl * wgt_lat;
  w.wcor1 = wc * wgt_cor; w.wcor2 = omwc * wgt_cor;
  w.wkm = m2 * cg0; w.wkp = p2 * cg0;
  return w;
}
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_fast_apply(const CtuFastW8<T>& w, CtuV2<T> f0, CtuV2<T> f1, CtuV2<T> f2, CtuV2<T> f3, CtuV2<T> f4, CtuV2<T> f5,
                                                   CtuV2<T> f6, CtuV2<T> f7) {
#pragma clang fp contract(off)
  CtuV2<T> r = ctu_v2fma<T>(-w.sumwn, f0, f0);                          // (1 - SUMWN) F
  r = ctu_v2fma<T>(w.wlon, f1, r);
  r = ctu_v2fma<T>(w.wlat1, f2, r);
  r = ctu_v2fma<T>(w.wlat2, f3, r);
  r = ctu_v2fma<T>(w.wcor1, f4, r);
  r = ctu_v2fma<T>(w.wcor2, f5, r);
  r = ctu_v2fma<T>(w.wkm, f6, r);
  r = ctu_v2fma<T>(w.wkp, f7, r);
  return r;
}
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_fast_stencil(CtuV2<T> xa, CtuV2<T> xb, CtuV2<T> ya, CtuV2<T> yb, CtuV2<T> cg0, T asink, T acosk, T zd, T xdella,
                                                     T ga, T zdg, T xdg, T wl, T omwl, T wc, T omwc, CtuV2<T> ab2, CtuV2<T> p2, CtuV2<T> m2, CtuV2<T> f0,
                                                     CtuV2<T> f1, CtuV2<T> f2, CtuV2<T> f3, CtuV2<T> f4, CtuV2<T> f5, CtuV2<T> f6, CtuV2<T> f7) {
  return ctu_fast_apply<T>(ctu_fast_w8<T>(xa, xb, ya, yb, cg0, asink, acosk, zd, xdella, ga, zdg, xdg, wl, omwl, wc, omwc, ab2, p2, m2), f0, f1, f2, f3, f4,
                           f5, f6, f7);
}
// the direction's four numbers of the great-circle term from TSP2 = 2 TANPH SP, TSM2 = 2 TANPH SM (see ctu_w8_stencil_abs)
template <typename T>
__device__ __forceinline__ void ctu_fast_dir(T tanph, T sp2, T sm2, T& ab2, T& p2, T& m2) {
#pragma clang fp contract(off)
  const T tsp2 = tanph * sp2, tsm2 = tanph * sm2;
  ab2 = m_max(tsp2, T(0)) + m_max(-tsm2, T(0));
  p2 = m_max(-tsp2, T(0));
  m2 = m_max(tsm2, T(0));
}
// The weights with refraction (IREFRA = 1, 2, 3; ctuw.F90:146-275, 403-527) for the lanes whose advection velocity keeps the sign of the group
// velocity (no upwind switch: the downwind weights are zero), in the hoisted form of ctu_fast_w8.  xa .. yb as there; su, sva, svb: the current's
// contribution to ADXP / ADYP with the sign of the group velocity component, (U COSPHM1) DELPRO CMTODEG SIGN(SINTH) and
// (V (1 + DP(IC)) / 2) DELPRO CMTODEG SIGN(COSTH) for the two neighbours in the direction's order (zero without currents); dthp, dthm: the
// complete theta-dot sums, fdp, fdm: those of the frequency shift (zero without currents).  neg: a component came out negative -- the
// current turns the advection velocity round: the caller takes the general path for this lane.
template <typename T>
struct CtuFastGenW {
  CtuV2<T> sumwn, wlon, wlat1, wlat2, wcor1, wcor2, wkm, wkp, wmm, wmp;
};
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_v2max0(CtuV2<T> a) { return CtuV2<T>{m_max(a.x, T(0)), m_max(a.y, T(0))}; }
template <typename T>
__device__ __forceinline__ CtuFastGenW<T> ctu_fast_wgen(CtuV2<T> xa, CtuV2<T> xb, CtuV2<T> ya, CtuV2<T> yb, T asink, T acosk, T su, T sva, T svb, T zd,
                                                        T xdella, T ga, T zdg, T xdg, T wl, T omwl, T wc, T omwc, CtuV2<T> dthp, CtuV2<T> dthm,
                                                        CtuV2<T> fdp, CtuV2<T> fdm, T fratio, T rfratio, bool& neg) {
#pragma clang fp contract(off)
  typedef CtuV2<T> F;
  const F adx_a = ctu_v2fma<T>(xa, F{asink, asink}, F{su, su}), adx_b = ctu_v2fma<T>(xb, F{asink, asink}, F{su, su});
  const F ady_a = ctu_v2fma<T>(ya, F{acosk, acosk}, F{sva, sva}), ady_b = ctu_v2fma<T>(yb, F{acosk, acosk}, F{svb, svb});
  const F lo = {m_min(m_min(adx_a.x, adx_b.x), m_min(ady_a.x, ady_b.x)), m_min(m_min(adx_a.y, adx_b.y), m_min(ady_a.y, ady_b.y))};
  neg = lo.x < T(0) || lo.y < T(0);
  const F dxx = zd - adx_b, dyy = xdella - ady_b;
  const F yag = ady_a * ga, xag = adx_a * ga, xbg = adx_b * ga;
  const F wgt_lat = dxx * yag;
  const F wgt_cor = adx_a * yag;
  CtuFastGenW<T> w;
  w.wlon = dyy * xag;
  // WKPMN(0) + WMPMN(0): (x + |x|) + (|y| - y) = 2 MAX(x, 0) + 2 MAX(-y, 0)
  const F two = {T(2), T(2)};
  F sumwn = two * (ctu_v2max0<T>(dthp) + ctu_v2max0<T>(-dthm)) + two * (ctu_v2max0<T>(fdp) + ctu_v2max0<T>(-fdm));
  sumwn = ctu_v2fma<T>(F{zdg, zdg}, ady_b, sumwn);
  sumwn = ctu_v2fma<T>(F{xdg, xdg}, adx_b, sumwn);
  w.sumwn = ctu_v2fma<T>(-xbg, ady_b, sumwn);
  w.wlat1 = wl * wgt_lat; w.wlat2 = omwl * wgt_lat;
  w.wcor1 = wc * wgt_cor; w.wcor2 = omwc * wgt_cor;
  w.wkm = two * ctu_v2max0<T>(dthm); w.wkp = two * ctu_v2max0<T>(-dthp);
  w.wmm = (two * ctu_v2max0<T>(fdm)) * fratio; w.wmp = (two * ctu_v2max0<T>(-fdp)) * rfratio;
  return w;
}
// propags2.F90:127-192 for such a lane: own, the five upwind space neighbours, the direction and frequency neighbours of the own spectrum
template <typename T>
__device__ __forceinline__ CtuV2<T> ctu_fast_apply_gen(const CtuFastGenW<T>& w, CtuV2<T> f0, CtuV2<T> flon, CtuV2<T> fla1, CtuV2<T> fla2, CtuV2<T> fco1,
                                                       CtuV2<T> fco2, CtuV2<T> fkm, CtuV2<T> fkp, CtuV2<T> fmm, CtuV2<T> fmp) {
#pragma clang fp contract(off)
  CtuV2<T> r = ctu_v2fma<T>(-w.sumwn, f0, f0);
  r = ctu_v2fma<T>(w.wlon, flon, r);
  r = ctu_v2fma<T>(w.wlat1, fla1, r);
  r = ctu_v2fma<T>(w.wcor1, fco1, r);
  r = ctu_v2fma<T>(w.wlat2, fla2, r);
  r = ctu_v2fma<T>(w.wcor2, fco2, r);
  r = ctu_v2fma<T>(w.wkm, fkm, r);
  r = ctu_v2fma<T>(w.wmm, fmm, r);
  r = ctu_v2fma<T>(w.wkp, fkp, r);
  r = ctu_v2fma<T>(w.wmp, fmp, r);
  return r;
}
// per-point scalars of the weights (ctuw.F90:146-170, 407-420)
template <typename T>
struct CtuPoint {
  T zd, cpm1, ga, tanph, dp[2], wl[2], wc[4];
};
template <typename T>
__device__ __forceinline__ CtuPoint<T> ctu_point(int ij, int ngy, const int* __restrict__ kxlt, const T* __restrict__ zdello, T xdella,
                                                 const T* __restrict__ cosph, const T* __restrict__ sinph, const T* __restrict__ wlat,
                                                 const T* __restrict__ wcor, const T* __restrict__ cosphm1) {
#pragma clang fp contract(off)
  CtuPoint<T> p;
  const int ky = kxlt[ij];
  p.zd = zdello[ky];
  p.cpm1 = cosphm1[ij];
  p.ga = T(1) / (p.zd * xdella);
  p.tanph = sinph[ky] / cosph[ky];
  for (int ic = 0; ic < 2; ic++) {
    int kk = ky + 1 + 2 * (ic + 1) - 3;  // 1-based row of the neighbour latitude, ctuwini.F90:159-162
    kk = kk < 1 ? 1 : (kk > ngy ? ngy : kk);
    p.dp[ic] = cosph[kk - 1] * p.cpm1;
    p.wl[ic] = wlat[ij * 2 + ic];
  }
  for (int ic = 0; ic < 4; ic++) p.wc[ic] = wcor[ij * 4 + ic];
  return p;
}
// TANPH*SP and TANPH*SM factors of direction k (ctuw.F90:407-420): DELTH0*(SINTH(K)+SINTH(K+-1))/R
template <typename T>
__device__ __forceinline__ void ctu_dirfac(const DevTab<T>* tab, int k, T delth0, T tanph, T& tsp, T& tsm) {
#pragma clang fp contract(off)
  const int kp1 = tab->KPM[k][2], km1 = tab->KPM[k][0];
  const T sp = delth0 * (tab->SINTH[k] + tab->SINTH[kp1]) / tab->R;
  const T sm = delth0 * (tab->SINTH[k] + tab->SINTH[km1]) / tab->R;
  tsp = tanph * sp;
  tsm = tanph * sm;
}

// Sub-grid obstructions (LSUBGRID, ctuw.F90:703-733): after the checks, the space weights of the neighbours are scaled by the
// transmission coefficients OBS[ij][8][NFRE] = OBSLAT(IJ,M,1:2), OBSLON(IJ,M,1:2), OBSCOR(IJ,M,1:4); SUMWN keeps its value
// (what the obstruction blocks is lost).  o points at plane 0 of (ij, m), planes are `stride` apart.
template <typename T>
__device__ __forceinline__ void ctu_obstruct8(T* w8, const T* o, int stride, int jx0, int jy0, int kc) {
#pragma clang fp contract(off)
  const T olon = o[(2 + jx0) * stride], olat = o[jy0 * stride], ocor = o[(4 + kc) * stride];
  w8[1] = w8[1] * olon;
  w8[2] = w8[2] * olat; w8[3] = w8[3] * olat;
  w8[4] = w8[4] * ocor; w8[5] = w8[5] * ocor;
}
