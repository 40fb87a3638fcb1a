// Device-side tables and wavefront helpers shared by the ecwam_hip kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ecwam_hip.h"

#define MAXA ECWAM_HIP_MAXANG
#define MAXF ECWAM_HIP_MAXFRE
#define MAXMC ECWAM_HIP_MAXMC
#define MAXTAP ECWAM_HIP_MAXTAP
#define MAXGC ECWAM_HIP_MAXGC
#define JTOT ECWAM_HIP_JTOT_TAUHF

// Module-level read-only state of the reference (YOWFRED, YOWPHYS, YOWINDN, YOWPCONS, YOWCOUP, YOWICE,
// YOWTABL, YOWUBUF selectors) in the working precision T.  One instance lives in device memory per
// context; kernels take a pointer to it and read it with wave-uniform (scalar) or per-lane loads.
template <typename T>
struct DevTab {
  int NANG, NFRE, NFRE_RED, NFRE_ODD, IDELT;
  int LLGCBZ0, LLNORMAGAM, LLCAPCHNK, LBIWBK, LICERUN, LMASKICE, LWAMRSETCI;
  int LWVFLX_SNL, LWFLUX, LCFLX, LWNEMOCOU, LWCOU, LWCOUAST, LWNEMOCOUWRS;
  int LWNEMOTAUOC, LWNEMOCOUSEND, LWNEMOCOUSTK;
  int IPHYS, IDAMPING;  // 0: SINPUT_JAN + SDISSIP_JAN, 1: SINPUT_ARD + SDISSIP_ARD (sinput.F90:102, sdissip.F90:76)
  int ICODE;     // 3: forcing by U10 (TAUT_Z0 gives u*); 1, 2: forcing by u* (Z0WAVE, U10 from the log profile; airsea.F90:93-117)
  int ISNONLIN;  // 0: DIA depth scaling from AKMEAN, 1: TRANSF per interaction frequency (snonlin.F90:126-150)
  int LCISCAL, LCIWA2, LCIWA3;  // sea-ice attenuation (implsch.F90:312-339, sdice2.F90, sdice3.F90)
  int LCIWA1, LWNEMOCOUIBR, LWNEMOCOUSTRN, NICT, NICH;  // SDICE1 scattering table, ice break-up coupling, CIMSSTRN
  int NSDSNTH, NTAP, MFRSTLW, MLSTHG, KFRH, NWAV_GC;
  // timing diagnostics only (env ECWAM_HIP_DEBUG_SKIP): bit mask of IMPLSCH phases to skip -- 1 SINPUT, 2 STRESSO scalars,
  // 4 SDISSIP, 8 DIA, 16 TAUT_Z0, 32 spectrum store of the update, 64 WSIGSTAR/swell set-up/SDIWBK.  0 in production.
  int DBG_SKIP;
  T XIMP, G, GM1, PI, ZPI, ZPI4GM1, ZPI4GM2, EPSMIN, ROWATER, ROWATERM1, EPSUS, EPSU10, ACD, BCD, ACDLIN, BCDLIN, CDMAX;
  T TAUOCMIN, TAUOCMAX, PHIEPSMIN, PHIEPSMAX, WSEMEAN_MIN, CIRC, R;
  T FRATIO, WETAIL, FRTAIL, WP1TAIL, FRIC, DELTH, FLOGSPRDM1;
  T XKAPPA, XNLEV, RNU, RNUM, BETAMAXOXKAPPA2, BMAXOKAP, GAMNCONST, ZALP, ALPHA, ALPHAMIN, ALPHAMAX, CHNKMIN_U;
  T TAUWSHELTER, DTHRN_A, DTHRN_U, TAILFACTOR, TAILFACTOR_PM, ANG_GC_A, ANG_GC_B, ANG_GC_C, RN1_RN, ALPHAPMAX;
  T SWELLF, SWELLF2, SWELLF3, SWELLF4, SWELLF5, SWELLF6, SWELLF7, SWELLF7M1, Z0RAT, Z0TUBMAX, ABMIN, ABMAX;
  T SDSBR, SSDSC2, SSDSC3, SSDSC4, SSDSC5, SSDSC6, MICHE;
  T EGRCRV, AFCRV, BFCRV;
  T X0TAUHF, EPS1, FLMIN, CITHRSH, CIBLOCK, CITHRSH_TAIL, ZALPWRS, BATHYMAX, WSPMIN, WSPMIN_RESET_TAUW;
  T DAL1, DAL2, XLOGKRATIOM1_GC, SQRTGOSURFT;
  T CDICWA, ZALPFACB, ZALPFACX;
  T ZIBRW_THRSH, TICMIN, DTIC, DHIC, HICMIN;
  T CIDEAC[36 * 16];  // [ih][it], cigetdeac.F90:64-71: NICH = 36, NICT = 16
  T CDIS, DELTA_SDIS, CDISVIS;
  // per-frequency
  T FR[MAXF], DFIM[MAXF], DFIMOFR[MAXF], DFIMFR[MAXF], DFIM_SIM[MAXF], RHOWG_DFIM[MAXF], ZPIFR[MAXF], FR5[MAXF];
  T COFRM4[MAXF], FLMAX[MAXF];
  // per-direction
  T TH[MAXA], COSTH[MAXA], SINTH[MAXA];
  T WTAUHF[JTOT];
  T SWELLFT[ECWAM_HIP_IAB + 1];  // 1-based like the reference
  // DIA (index values 0-based here)
  int IKP[MAXMC], IKP1[MAXMC], IKM[MAXMC], IKM1[MAXMC];  // 1-based frequency values as in the reference
  T AF11[MAXMC];
  int K1W[2][MAXA], K2W[2][MAXA], K11W[2][MAXA], K21W[2][MAXA];  // [kh][k], 0-based direction
  // pull form of the DIA (implsch.hip snonlin_pull): inverse rotations IK*[kh][c] = the lane whose K*W target is c,
  // D11/D21[kh] = K11W-K1W / K21W-K2W (+1 or -1 mod NANG); DIA_PULL = 1 when the tables have the rotation structure
  int IK1[2][MAXA], IK2[2][MAXA], D11[2], D21[2], DIA_PULL;
  int INLCOEF[MAXMC][5];                                         // 0-based frequency
  T RNLCOEF[MAXMC][25];
  // k_implsch4: one 32-word record per interaction frequency, read with 16-byte loads: words 0..11 the gather set (FTAIL, GW1..GW8,
  // AF11, 2 spare), words 12..27 the scatter set (RNLCOEF(6:13), RNLCOEF(18:25)); V4_ROWS = 1 when INLCOEF equals the clamped
  // MC, MC+2, MC+3, MC-4, MC-3
  alignas(16) T DIACF[MAXMC][32];
  // the same interaction coefficients in separable form (inisnonlin.F90:186-241: every GW / FKLAM* coefficient is a frequency factor times
  // one of the two angular interpolation weights CL11 | ACL1 (+ leg) or CL21 | ACL2 (- leg)): per MC the frequency factors
  // [0..3] GP, GP1, GM, GM1 (SAP = CL11 W(K1) + ACL1 W(K11), W = GP F(:,IP) + GP1 F(:,IP1); SAM likewise), [4..7] FKLAMP, FKLAMP1, their squares,
  // [8..11] FKLAMM, FKLAMM1, their squares; [12..15] GP, GP1, GM, GM1 of the NEXT interaction (the sweep stages the rows of interaction
  // MC + 1 at the end of interaction MC); DIAANG = CL11, ACL1, CL21, ACL2 and their squares
  alignas(16) T DIAW[MAXMC][16];
  T DIAANG[8];
  // the words of DIACF / DIAW the window form of the sweep reads per interaction, contiguous (V4_RECPF: fetched one interaction ahead):
  // 0..6 = DIACF 9, 10, 11, 28, 29, 30, 31; 8..19 = DIAW 4..15; row MLSTHG repeats the last interaction (the prefetch of the last one)
  alignas(16) T DIAREC[MAXMC + 1][20];
  // word 31: FTAIL as the sweep applies it (1 between MFR1STFR and MFRLSTFR); words 10, 11, 28..30 of the record of interaction MC: ZPIFR of the (clamped) row MC-3 and COFRM4, FLMAX, RHOWG_DFIM, ZPIFR of row
  // MC-5, which that interaction updates -- the record is one scalar load, a lane table costs a v_readlane per value
  // SINPUT_ARD's per-frequency constants as one 8-word record per row: ZPIFR, DFIM, -SWELLF5 2 SQRT(2 NU_AIR SIG), -SWELLF 16 SIG**2 / G
  // (sinput_ard.F90:343-347), RHOWG_DFIM, DFIMOFR, DFIMFR, 1 spare
  alignas(32) T SINROW[MAXF][8];
  int V4_ROWS;
  // saturation filter: [k2][k] so that lanes (k) read consecutive words
  int INDICESSAT[MAXTAP][MAXA];  // 0-based direction
  T SATWEIGHTS[MAXTAP][MAXA];
  // CTU selectors, 0-based
  int KPM[MAXA][3], JXO[MAXA][2], JYO[MAXA][2], KCR[MAXA][4];
  // gravity-capillary tables, 1-based like the reference
  T XK_GC[MAXGC], XKM_GC[MAXGC], OMEGA_GC[MAXGC], OMXKM3_GC[MAXGC], CM_GC[MAXGC], C2OSQRTVG_GC[MAXGC];
  T XKMSQRTVGOC2_GC[MAXGC], OM3GMKM_GC[MAXGC], DELKCC_GC_NS[MAXGC], DELKCC_OMXKM3_GC[MAXGC];
};

// ---- precision-generic math ---------------------------------------------------------------------
__device__ __forceinline__ float m_sin(float x) { return sinf(x); }
__device__ __forceinline__ double m_sin(double x) { return sin(x); }
__device__ __forceinline__ float m_cos(float x) { return cosf(x); }
__device__ __forceinline__ double m_cos(double x) { return cos(x); }
__device__ __forceinline__ float m_exp(float x) { return expf(x); }
__device__ __forceinline__ double m_exp(double x) { return exp(x); }
__device__ __forceinline__ float m_log(float x) { return logf(x); }
__device__ __forceinline__ double m_log(double x) { return log(x); }
__device__ __forceinline__ float m_log10(float x) { return log10f(x); }
__device__ __forceinline__ double m_log10(double x) { return log10(x); }
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double m_tanh(double x) { return tanh(x); }
__device__ __forceinline__ float m_sinh(float x) { return sinhf(x); }
__device__ __forceinline__ double m_sinh(double x) { return sinh(x); }
__device__ __forceinline__ float m_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float m_pow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double m_pow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double m_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float m_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double m_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float m_floor(float x) { return floorf(x); }
__device__ __forceinline__ double m_floor(double x) { return floor(x); }
__device__ __forceinline__ float m_sign(float a, float b) { return copysignf(a, b); }
__device__ __forceinline__ double m_sign(double a, double b) { return copysign(a, b); }
// SIGN(MIN(ABS(g), lim), g) for lim >= 0 (the limiter of implsch.F90:386-388) = g clamped to [-lim, lim]: one v_med3_f32
__device__ __forceinline__ float m_clamp_sym(float g, float lim) { return __builtin_amdgcn_fmed3f(g, -lim, lim); }
__device__ __forceinline__ double m_clamp_sym(double g, double lim) { return copysign(fmin(fabs(g), lim), g); }
__device__ __forceinline__ int m_nint(float x) { return (int)lroundf(x); }
__device__ __forceinline__ int m_nint(double x) { return (int)lround(x); }
// ---- IMPLSCH hot-loop math: single precision goes straight to the hardware transcendental unit (v_rcp/v_sqrt/v_exp/
// v_log, <= 1 ulp each, no range fix-up code: the arguments on these paths are bounded, see the call sites);
// double precision keeps the library routines for EXP / LOG (a Taylor EXP without special cases measured the same).  f_exp(x) = 2^(x*log2e): relative error <= |x|*1.2e-7.
// ECWAM_HIP_STRICT (build variants of ecwam_amd/build.py): bit 0 -- single-precision divisions, reciprocals and
// square roots correctly rounded; bit 1 -- single-precision EXP / LOG through the library routines (<= 1 ulp).  0 in the product build.
#ifndef ECWAM_HIP_STRICT
#define ECWAM_HIP_STRICT 0
#endif
#if ECWAM_HIP_STRICT & 1
__device__ __forceinline__ float f_div(float a, float b) { return a / b; }
#else
__device__ __forceinline__ float f_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#endif
// double precision: hardware seed (v_rcp_f64 / v_rsq_f64, ~2^-26) + two Newton steps in FMA form: < 2 ulp, no IEEE
// special-case sequence (div_scale/div_fmas/div_fixup).  Arguments on these paths are finite, non-zero and normal.
__device__ __forceinline__ double f_rcp(double b);
__device__ __forceinline__ double f_div(double a, double b) {
  const double r = f_rcp(b);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
#if ECWAM_HIP_STRICT & 1
__device__ __forceinline__ float f_rcp(float b) { return 1.0f / b; }
#else
__device__ __forceinline__ float f_rcp(float b) { return __builtin_amdgcn_rcpf(b); }
#endif
__device__ __forceinline__ double f_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
#if ECWAM_HIP_STRICT & 1
__device__ __forceinline__ float f_sqrt(float x) { return sqrtf(x); }
#else
__device__ __forceinline__ float f_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#endif
__device__ __forceinline__ double f_sqrt(double x) {   // x >= 0, finite: seed, one coupled Newton step for SQRT and 1/(2 SQRT), one correction
  const double y = __builtin_amdgcn_rsq(x);
  double s = x * y, h = 0.5 * y;
  const double r = fma(-h, s, 0.5);
  s = fma(s, r, s);
  h = fma(h, r, h);
  s = fma(fma(-s, s, x), h, s);
  return x == 0.0 ? 0.0 : s;
}
#if ECWAM_HIP_STRICT & 1
__device__ __forceinline__ float f_rsq(float x) { return 1.0f / sqrtf(x); }
#else
__device__ __forceinline__ float f_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
#endif
__device__ __forceinline__ double f_rsq(double x) {   // x > 0 (the callers discard the result of x = 0)
  double y = __builtin_amdgcn_rsq(x);
  y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
  y = fma(0.5 * y, fma(-x * y, y, 1.0), y);
  return y;
}
#if ECWAM_HIP_STRICT & 2
__device__ __forceinline__ float f_exp(float x) { return expf(x); }
#else
__device__ __forceinline__ float f_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
#endif
__device__ __forceinline__ double f_exp(double x) { return exp(x); }
#if ECWAM_HIP_STRICT & 2
__device__ __forceinline__ float f_log(float x) { return logf(x); }
#else
__device__ __forceinline__ float f_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
#endif
__device__ __forceinline__ double f_log(double x) { return log(x); }
// a / b for the implicit update (implsch.F90:384-386: GTEMP2 = DELT SL / MAX(1 - DELT5 FLD, 1)).  Where the dissipation is stiff, F + GTEMP2
// = F / GTEMP1 is a difference of nearly equal numbers: a relative error e of the quotient comes out as e GTEMP1 in the new spectrum
// (GTEMP1 reaches 10^2..10^3 in the tail), and the 1 ulp of the bare reciprocal doubled the distance of the significant wave height to the
// double-precision answer (1.7e-6 against 7e-7 for correctly rounded arithmetic, profiles/r03_sp_error_attribution.txt).  One Newton
// step on the quotient: within half an ulp + 2^-46 relative of a / b, two fused multiply-adds.
__device__ __forceinline__ float f_div_r(float a, float b) {
  const float r = __builtin_amdgcn_rcpf(b);
  const float q = a * r;
  return __builtin_fmaf(__builtin_fmaf(-q, b, a), r, q);
}
__device__ __forceinline__ double f_div_r(double a, double b) { return f_div(a, b); }
// site-selective exactness for the error attribution (tests/diag/sp_error_attribution.py): bit SITE of ECWAM_HIP_STRICT makes the
// operation at the call sites tagged SITE correctly rounded (the variant builds drop -fno-hip-fp32-correctly-rounded-divide-sqrt)
template <int SITE> __device__ __forceinline__ float fs_rcp(float x) { if constexpr ((ECWAM_HIP_STRICT & SITE) != 0) return 1.0f / x; else return f_rcp(x); }
template <int SITE> __device__ __forceinline__ double fs_rcp(double x) { return f_rcp(x); }
template <int SITE> __device__ __forceinline__ float fs_div(float a, float b) { if constexpr ((ECWAM_HIP_STRICT & SITE) != 0) return a / b; else return f_div(a, b); }
template <int SITE> __device__ __forceinline__ double fs_div(double a, double b) { return f_div(a, b); }
template <int SITE> __device__ __forceinline__ float fs_rsq(float x) { if constexpr ((ECWAM_HIP_STRICT & SITE) != 0) return 1.0f / sqrtf(x); else return f_rsq(x); }
template <int SITE> __device__ __forceinline__ double fs_rsq(double x) { return f_rsq(x); }
template <int SITE> __device__ __forceinline__ float fs_sqrt(float x) { if constexpr ((ECWAM_HIP_STRICT & SITE) != 0) return sqrtf(x); else return f_sqrt(x); }
template <int SITE> __device__ __forceinline__ double fs_sqrt(double x) { return f_sqrt(x); }
template <typename T>
__device__ __forceinline__ T m_pow4(T x) {
  T x2 = x * x;
  return x2 * x2;
}

// ---- wavefront helpers (wave64) -------------------------------------------------------------------
// value held by lane `l` (l wave-uniform) -> every lane, through SGPRs (v_readlane_b32)
__device__ __forceinline__ float lane_get(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ int lane_get(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double lane_get(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// ---- DPP cross-lane moves (no LDS traffic): dst lane reads the lane selected by CTRL -----------------
// CTRL: 0xB1 quad_perm[1,0,3,2], 0x4E quad_perm[2,3,0,1], 0x141 row_half_mirror, 0x140 row_mirror,
//       0x142 row_bcast:15, 0x143 row_bcast:31, 0x138 wave_shr:1 (lane c <- c-1), 0x130 wave_shl:1 (lane c <- c+1)
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v, float old = 0.f) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ double dpp_mov(double v, double old = 0.0) {
  int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROWMASK, 0xF, false);
  int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}

// Wave-uniform reductions: 4 DPP butterfly steps inside each row of 16 lanes, two row broadcasts, and a
// v_readlane of lane 63 (result in SGPRs).  Inactive lanes must contribute the identity (0 / -inf..0).
template <typename T>
__device__ __forceinline__ T usum(T v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  v += dpp_mov<0x142, 0xA>(v);
  v += dpp_mov<0x143, 0xC>(v);
  return lane_get(v, 63);
}
template <typename T>
__device__ __forceinline__ T umax(T v) {  // for values >= 0 (identity 0)
  v = m_max(v, dpp_mov<0xB1>(v, v));
  v = m_max(v, dpp_mov<0x4E>(v, v));
  v = m_max(v, dpp_mov<0x141>(v, v));
  v = m_max(v, dpp_mov<0x140>(v, v));
  v = m_max(v, dpp_mov<0x142, 0xA>(v, v));
  v = m_max(v, dpp_mov<0x143, 0xC>(v, v));
  return lane_get(v, 63);
}

// ---- multi-value reductions (gfx950 v_permlane32_swap / v_permlane16_swap): fold the wave so that each 16-lane row (or
// 32-lane half) carries a different quantity, then ONE DPP butterfly reduces all of them.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// a <- [a.lanes0-31 | b.lanes0-31], b <- [a.lanes32-63 | b.lanes32-63]
__device__ __forceinline__ void swap32(float& a, float& b) {
  u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
// rows of 16 lanes: a <- [a.r0, b.r0, a.r2, b.r2], b <- [a.r1, b.r1, a.r3, b.r3]
__device__ __forceinline__ void swap16(float& a, float& b) {
  u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap32(double& a, double& b) {
  u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
  u32x2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  u32x2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}
// two wave sums at once (9 VALU instead of 14).  Inactive lanes must hold 0.
template <typename T>
__device__ __forceinline__ void usum2(T a, T b, T& sa, T& sb) {
  swap32(a, b);
  T v = a + b;  // lanes 0-31: a_k + a_{k+32}, lanes 32-63: b
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  v += dpp_mov<0x142, 0xA>(v);
  sa = lane_get(v, 31); sb = lane_get(v, 63);
}
// four wave sums at once (14 VALU instead of 28)
template <typename T>
__device__ __forceinline__ void usum4(T a, T b, T c, T d, T& sa, T& sb, T& sc, T& sd) {
  swap16(a, b);
  T ab = a + b;  // rows: [a0+a1, b0+b1, a2+a3, b2+b3]
  swap16(c, d);
  T cd = c + d;
  swap32(ab, cd);
  T v = ab + cd;  // rows: [A, B, C, D] partial sums
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  sa = lane_get(v, 0); sb = lane_get(v, 16); sc = lane_get(v, 32); sd = lane_get(v, 48);
}
// two wave maxima at once, values >= 0 (identity 0)
template <typename T>
__device__ __forceinline__ void umax2(T a, T b, T& ma, T& mb) {
  swap32(a, b);
  T v = m_max(a, b);
  v = m_max(v, dpp_mov<0xB1>(v, v));
  v = m_max(v, dpp_mov<0x4E>(v, v));
  v = m_max(v, dpp_mov<0x141>(v, v));
  v = m_max(v, dpp_mov<0x140>(v, v));
  v = m_max(v, dpp_mov<0x142, 0xA>(v, v));
  ma = lane_get(v, 31); mb = lane_get(v, 63);
}

// rotation by one lane inside the first n lanes: rot_up: lane c <- c-1 (lane 0 <- n-1); rot_dn: lane c <- c+1 (n-1 <- 0)
template <typename T>
__device__ __forceinline__ T rot_up(T v, int lane, int n) {
  const T w = lane_get(v, n - 1);
  const T s = dpp_mov<0x138>(v);
  return lane == 0 ? w : s;
}
template <typename T>
__device__ __forceinline__ T rot_dn(T v, int lane, int n) {
  const T w = lane_get(v, 0);
  const T s = dpp_mov<0x130>(v);
  return lane == n - 1 ? w : s;
}
// arbitrary pull: every lane reads lane src (ds_bpermute: LDS crossbar, no LDS memory)
__device__ __forceinline__ float lane_pull(float v, int src) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}
__device__ __forceinline__ double lane_pull(double v, int src) {
  int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
  int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
