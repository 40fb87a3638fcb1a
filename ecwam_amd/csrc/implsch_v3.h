// IMPLSCH, third kernel generation (flag set A, single precision, NANG = 36): THREE sea points per wavefront, 18 lanes per
// point, lane j of a point holds the direction pair (K = j, j + 18) as one packed-fp32 operand (layout and primitives
// validated phase by phase in snonlin3.hip).  Included by implsch.hip after implsch_v2.h: the lane-per-point scalar stages
// (TAUT_Z0, STRESSO / TAU_PHI_HF, WSIGSTAR, swell set-up, SDIWBK) are the functions of k_implsch2 -- a wave's three points are
// exactly the three-point batch they work on, so every wave is autonomous (no block barriers).
//
// LDS per wave: the pair tile F [3][M][18][2], a table of per-frequency factors [3][M][V3_NFAC], the SINPUT factors of the current
// call [3][M][8], the row integrals of SINPUT [3][M][4] and the point scalars [3][NSC].  There is no FLD tile: the wind-input
// coefficient of the second SINFLX call is parked in the point's XLLWS output rows (read back and overwritten with the flag by the
// sweep) and the dissipation coefficient is evaluated inside the sweep for the row that is about to be updated.
#pragma once

typedef float V3F2 __attribute__((ext_vector_type(2)));
#define V3G 18
#define V3P 3
#define V3_NFAC 6
#define V3_MAXTAP 17
// per (point, M); the frequency-only module tables (DFIM, ZPIFR, RHOWG_DFIM ...) are read from the DevTab with a uniform index (scalar loads)
enum { FA_WAVNUM = 0, FA_CINV, FA_STOK, FA_SQ, FA_SBO, FA_XK2CG };   // FA_SQ = SQRT(WAVNUM)

__device__ __forceinline__ float v3_bperm(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ V3F2 v3_pull(V3F2 v, int addr, bool swap) {
  const float a = v3_bperm(addr, v.x), b = v3_bperm(addr, v.y);
  V3F2 r;
  r.x = swap ? b : a;
  r.y = swap ? a : b;
  return r;
}
__device__ __forceinline__ V3F2 v3_same(V3F2 v, int addr) {
  V3F2 r;
  r.x = v3_bperm(addr, v.x);
  r.y = v3_bperm(addr, v.y);
  return r;
}
struct V3Rot { int a9, a3, a6, a1, a2; };
// sums over the 18 lanes of a point (x and y are independent quantities), every lane gets them
__device__ __forceinline__ V3F2 v3_allsum(V3F2 v, const V3Rot& r) {
  v = v + v3_same(v, r.a9);
  v = v + (v3_same(v, r.a3) + v3_same(v, r.a6));
  v = v + (v3_same(v, r.a1) + v3_same(v, r.a2));
  return v;
}
__device__ __forceinline__ float v3_allmax(float v, const V3Rot& r) {
  v = fmaxf(v, v3_bperm(r.a9, v));
  v = fmaxf(v, fmaxf(v3_bperm(r.a3, v), v3_bperm(r.a6, v)));
  v = fmaxf(v, fmaxf(v3_bperm(r.a1, v), v3_bperm(r.a2, v)));
  return v;
}

struct V3Ctx {
  int lane, p, j, NANG, NFRE, N;
  bool grp, act;
  const float* tF;     // this lane's pair in row 0 of its point's tile
  float* tFw;
  const float* tP;     // row 0 of the point's tile
  const float* tFac;   // factor table of the point
  float* tSin;         // SINPUT factors of the point [M][8]
  float* tRow;         // row integrals of the point [M][4]
  float* c;            // scalars of the point [NSC]
  V3Rot rot;
  V3F2 sinth, costh;
  // module tables per frequency, lane m holds M = m+1: broadcast with v_readlane inside the M loops (no scalar loads there)
  float rDFIM, rDFIMOFR, rDFIMFR, rZPIFR, rRHOWG, rCOFRM4, rFLMAX, rC5;   // rC5 = -SWELLF5*2*SQRT(2*NU_AIR*SIG) (sinput_ard.F90:343)
};

// SINPUT_ARD (sinput_ard.F90:153-520) for one SINFLX call; see k_sinput3 (snonlin3.hip) for the layout.  Outputs: XLLWS masks of the
// two halves, row integrals into tRow (X, Y, S), FEMEANWS integrands (wse: x = SUM DFIM*F, y = SUM DFIMOFR*F over the windsea
// bins; wslast = windsea part of the last row), apl (negative wind input per direction pair), and -- LLSNEG -- the wind-input
// coefficient of every row into gfl (the point's XLLWS rows).
template <int NGST, bool LLSNEG>
__device__ void v3_sinput(const DevTab<float>& tb, const V3Ctx& L, float UFRIC, float Z0M, float RAORW, float SIG_N, float TEMP2, float PTURB,
                          float AIRD_PVISC, float sinwd, float coswd, float* __restrict__ gfl, unsigned long long& xm0,
                          unsigned long long& xm1, V3F2& wse, V3F2& wslast, V3F2& apl) {
  const int NFRE = L.NFRE, NANG = L.NANG;
  const float CONST1 = tb.BETAMAXOXKAPPA2, ABS_TAUWSHELTER = fabsf(tb.TAUWSHELTER);
  const float FU = fabsf(tb.SWELLF3), FUD = tb.SWELLF2, ROGOROAIR = tb.G / RAORW;
  const float AVG = 1.0f / (float)NGST;
  if (L.grp)
    for (int m = L.j; m < NFRE; m += V3G) L.tSin[m] = m_log(L.tFac[m * V3_NFAC + FA_WAVNUM] * Z0M);
  WSYNC();
  const float XKAPPA = tb.XKAPPA, ZALP = tb.ZALP;
  float USTP[2], XSTRESS[2] = {0.f, 0.f}, YSTRESS[2] = {0.f, 0.f}, TAUX[2], TAUY[2];
  if (NGST == 1) USTP[0] = UFRIC;
  else { USTP[0] = UFRIC * (1.0f + SIG_N); USTP[1] = UFRIC * (1.0f - SIG_N); }
#pragma unroll
  for (int ig = 0; ig < NGST; ig++) {
    const float USG2 = USTP[ig] * USTP[ig];
    TAUX[ig] = USG2 * sinwd;
    TAUY[ig] = USG2 * coswd;
  }
  xm0 = 0ull; xm1 = 0ull;
  wse = V3F2{0.f, 0.f}; wslast = V3F2{0.f, 0.f}; apl = V3F2{0.f, 0.f};
  for (int m = 0; m < NFRE; m++) {
    const float SIGm = lane_get(L.rZPIFR, m);
    const float ZCN = L.tSin[m], cinv_m = L.tFac[m * V3_NFAC + FA_CINV];
    const float CONSTF = ROGOROAIR * cinv_m * lane_get(L.rDFIM, m);
    const float DSTAB1 = LLSNEG ? (lane_get(L.rC5, m) * AIRD_PVISC) * L.tFac[m * V3_NFAC + FA_WAVNUM] : 0.f;
    const float CNSN = (SIGm * CONST1) * RAORW;
    const float TEMP1 = LLSNEG ? (-tb.SWELLF * 16.0f * (SIGm * SIGm) / tb.G) * RAORW : 0.f;
    const V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
    V3F2 SLP[2], FLP[2];
    bool xl0 = false, xl1 = false;
#pragma unroll
    for (int ig = 0; ig < NGST; ig++) {
      const float TAUPX = TAUX[ig] - ABS_TAUWSHELTER * XSTRESS[ig];
      const float TAUPY = TAUY[ig] - ABS_TAUWSHELTER * YSTRESS[ig];
      const float h2 = TAUPX * TAUPX + TAUPY * TAUPY;
      const bool zero = !(h2 > 0.f);
      const float rh = f_rsq(h2);
      const float h = zero ? 0.f : h2 * rh;
      const float COSU = zero ? 1.f : TAUPY * rh, SINU = zero ? 0.f : TAUPX * rh;
      USTP[ig] = f_sqrt(h);
      const float UCN = USTP[ig] * cinv_m;
      const float UCNZALPD = XKAPPA * f_rcp(UCN + ZALP);
      const V3F2 coslp = L.costh * COSU + L.sinth * SINU;
      V3F2 gam0 = {0.f, 0.f};
      {
        const bool c0 = coslp.x > 0.01f, c1 = coslp.y > 0.01f;
        const float Z0 = ZCN + UCNZALPD * f_rcp(coslp.x), Z1 = ZCN + UCNZALPD * f_rcp(coslp.y);
        const bool n0 = c0 && (Z0 < 0.f), n1 = c1 && (Z1 < 0.f);
        if (__builtin_amdgcn_ballot_w64(n0 || n1) != 0ull) {
          const V3F2 ZL = {Z0, Z1};
          const V3F2 Z2X = ZL * ZL * (coslp * UCN);
          const V3F2 ex = {f_exp(Z0), f_exp(Z1)};
          const V3F2 g = ex * Z2X * Z2X * CNSN;
          gam0.x = n0 ? g.x : 0.f;
          gam0.y = n1 ? g.y : 0.f;
          xl0 = xl0 || n0;
          xl1 = xl1 || n1;
        }
      }
      V3F2 dstab = {0.f, 0.f};
      if (LLSNEG) {
        const V3F2 DSTAB2 = TEMP1 * (TEMP2 + (FU + FUD * coslp) * USTP[ig]);
        dstab = DSTAB1 + PTURB * DSTAB2;
      }
      FLP[ig] = gam0 + dstab;
      SLP[ig] = gam0 * f;
    }
    V3F2 sp = SLP[0], fl = FLP[0];
    if (NGST == 2) { sp = sp + SLP[1]; fl = fl + FLP[1]; }
    sp = AVG * sp;
    fl = AVG * fl;
    const bool anygrow = __builtin_amdgcn_ballot_w64(xl0 || xl1) != 0ull;
    if (anygrow) {
      float xrow = 0.f, yrow = 0.f;
#pragma unroll
      for (int ig = 0; ig < NGST; ig++) {
        const V3F2 sx = L.grp ? SLP[ig] * L.sinth : V3F2{0.f, 0.f}, sy = L.grp ? SLP[ig] * L.costh : V3F2{0.f, 0.f};
        const V3F2 xs = v3_allsum(V3F2{sx.x + sx.y, sy.x + sy.y}, L.rot);
        XSTRESS[ig] = XSTRESS[ig] + CONSTF * xs.x;
        YSTRESS[ig] = YSTRESS[ig] + CONSTF * xs.y;
        xrow += xs.x;
        yrow += xs.y;
      }
      float srow = 0.f;
      if (LLSNEG) srow = v3_allsum(L.grp ? V3F2{sp.x + sp.y, 0.f} : V3F2{0.f, 0.f}, L.rot).x;
      if (L.grp && L.j == 0) { float* r = L.tRow + m * 3; r[0] = AVG * xrow; r[1] = AVG * yrow; r[2] = srow; }
    } else if (L.grp && L.j == 0) { float* r = L.tRow + m * 3; r[0] = 0.f; r[1] = 0.f; r[2] = 0.f; }
    if (LLSNEG) {
      apl = apl + (fl * f - sp) * lane_get(L.rRHOWG, m);
      if (L.act) *reinterpret_cast<V3F2*>(gfl + (size_t)m * NANG) = fl;
    }
    if (xl0) xm0 |= (1ull << m);
    if (xl1) xm1 |= (1ull << m);
    const V3F2 x = {xl0 ? f.x : 0.f, xl1 ? f.y : 0.f};
    const float* fac = L.tFac + m * V3_NFAC;
    wse = wse + V3F2{lane_get(L.rDFIM, m), lane_get(L.rDFIMOFR, m)} * (x.x + x.y);
    wslast = x;
  }
  WSYNC();
}

template <int WPB>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 2))) k_implsch3(const DevTab<float>* __restrict__ tp, int kijs, int kijl, float* __restrict__ fl1,
                                                       const float* __restrict__ wvprpt, float* __restrict__ ffa,
                                                       float* __restrict__ intfa, int* __restrict__ mij_out, float* __restrict__ xllws) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const DevTab<float>& tb = *tp;
  V3Ctx L;
  L.NANG = tb.NANG; L.NFRE = tb.NFRE; L.N = L.NANG * L.NFRE;
  const int NANG = L.NANG, NFRE = L.NFRE, N = L.N;
  const int wave = threadIdx.x >> 6;
  L.lane = threadIdx.x & 63;
  L.p = L.lane / V3G; L.j = L.lane - L.p * V3G;
  const int lane = L.lane, j = L.j;
  const int ij0 = kijs + (blockIdx.x * WPB + wave) * V3P;
  if (ij0 >= kijl) return;
  const int n = kijl - ij0 < V3P ? kijl - ij0 : V3P;   // points of this wave
  if (tb.DBG_SKIP == 100) return;
  L.grp = L.p < V3P;
  L.act = L.grp && L.p < n;
  const int p = L.grp ? L.p : 0;
  const int ij = ij0 + (L.act ? L.p : 0);
  const int per_wave = (V3P * (N + V3_NFAC * NFRE + NFRE + 3 * NFRE) + V3P * NSC + 3) & ~3;
  float* sF = reinterpret_cast<float*>(smem_raw) + (size_t)wave * per_wave;
  float* sFac = sF + V3P * N;
  float* sSin = sFac + V3P * V3_NFAC * NFRE;
  float* sRow = sSin + V3P * NFRE;
  float* sSC = sRow + V3P * 3 * NFRE;   // [3][NSC]
  L.tP = sF + p * N; L.tF = L.tP + 2 * j; L.tFw = sF + p * N + 2 * j;
  L.tFac = sFac + p * V3_NFAC * NFRE; L.tSin = sSin + p * NFRE; L.tRow = sRow + p * 3 * NFRE; L.c = sSC + p * NSC;
  const int base = p * V3G;
#define V3_ROT(r) (4 * (base + ((j + (r)) >= V3G ? j + (r) - V3G : j + (r))))
  L.rot.a9 = V3_ROT(9); L.rot.a3 = V3_ROT(3); L.rot.a6 = V3_ROT(6); L.rot.a1 = V3_ROT(1); L.rot.a2 = V3_ROT(2);
#undef V3_ROT
  {
    const int mi = lane < NFRE ? lane : 0;
    L.rDFIM = tb.DFIM[mi]; L.rDFIMOFR = tb.DFIMOFR[mi]; L.rDFIMFR = tb.DFIMFR[mi]; L.rZPIFR = tb.ZPIFR[mi]; L.rRHOWG = tb.RHOWG_DFIM[mi];
    L.rCOFRM4 = tb.COFRM4[mi]; L.rFLMAX = tb.FLMAX[mi];
    L.rC5 = -tb.SWELLF5 * 2.0f * m_sqrt(2.0f * tb.RNU * tb.ZPIFR[mi]);
  }
  L.sinth = V3F2{tb.SINTH[j], tb.SINTH[j + V3G]};
  L.costh = V3F2{tb.COSTH[j], tb.COSTH[j + V3G]};
  float* c = L.c;

  // ---- spectra F[ij][K][M] (coalesced) -> pair tile; a short last wave replicates its last point
  {
    const float rnf = 1.0f / (float)NFRE;
    const bool full = (n == V3P);   // the three blocks are contiguous in memory: one flat loop keeps all the loads in flight
    for (int q = 0; q < (full ? 1 : V3P); q++) {
      const float* g = fl1 + (size_t)(ij0 + (q < n ? q : n - 1)) * N;
      float* t = sF + q * N;
      const int cnt = full ? V3P * N : N;
      for (int e = lane; e < cnt; e += 64) {
        const int kq = (int)(((float)e + 0.5f) * rnf);          // e / NFRE (exact for these sizes), kq = point * NANG + K
        const int m = e - kq * NFRE;
        const int pq = (kq >= 2 * NANG) ? 2 : (kq >= NANG ? 1 : 0), k = kq - pq * NANG;
        t[pq * N + m * NANG + 2 * (k >= V3G ? k - V3G : k) + (k >= V3G ? 1 : 0)] = g[e];
      }
    }
  }
  // ---- point scalars + first TAUT_Z0, one lane per point (sinflx.F90:105-122)
  if (lane < V3P) {
    const int pid = ij0 + (lane < n ? lane : n - 1);
    const float* ff = ffa + (size_t)pid * ECWAM_HIP_NFF;
    float* q = sSC + lane * NSC;
    const float AIRD = ff[0], WDWAVE = ff[1], WSWAVE = ff[3];
    q[C_AIRD] = AIRD; q[C_WDWAVE] = WDWAVE; q[C_WSWAVE] = WSWAVE; q[C_WSTAR] = ff[4];
    q[C_TAUW] = ff[8]; q[C_TAUWDIR] = ff[9];
    q[C_RAORW] = m_max(AIRD, 1.0f) * tb.ROWATERM1; q[C_EMAXDPT] = ff[14]; q[C_DEPTH] = ff[15];
    q[C_SINWD] = m_sin(WDWAVE); q[C_COSWD] = m_cos(WDWAVE);
    q[C_RNFAC] = 1.0f;
    float UFRIC = ff[7], Z0M = ff[10], Z0B = ff[11], CHRNCK = ff[12];
    taut_z0_a(tb, 0, WSWAVE, WDWAVE, ff[8], ff[9], UFRIC, Z0M, Z0B, CHRNCK);
    q[C_UFRIC] = UFRIC; q[C_Z0M] = Z0M; q[C_Z0B] = Z0B; q[C_CHRNCK] = CHRNCK;
    q[C_SPARE] = ff[2];   // CICOVER
  }
  // ---- per-frequency factors of the point: lane j fills M = j+1 and j+19
  if (L.grp) {
    const float* wp = wvprpt + (size_t)ij * ECWAM_HIP_NWPR * NFRE;
    const float DEPTH = ffa[(size_t)ij * ECWAM_HIP_NFF + 15];
    const float TPIINV = 1.0f / tb.ZPI;
    for (int m = j; m < NFRE; m += V3G) {
      float* f = sFac + (p * NFRE + m) * V3_NFAC;
      const float WAVNUM = wp[m], XK2CG = wp[3 * NFRE + m];
      const float sq = m_sqrt(WAVNUM);
      f[FA_WAVNUM] = WAVNUM; f[FA_CINV] = wp[2 * NFRE + m]; f[FA_XK2CG] = XK2CG; f[FA_SQ] = sq;
      f[FA_STOK] = (m < tb.NFRE_ODD) ? wp[4 * NFRE + m] * tb.DFIM_SIM[m] : 0.f;
      float sbo = 0.f;   // sbottom.F90:79-89
      if (m < tb.NFRE_RED && DEPTH < tb.BATHYMAX) sbo = (-2.0f * 0.038f * tb.GM1) * WAVNUM / m_sinh(m_min(2.0f * DEPTH * WAVNUM, 50.0f));
      f[FA_SBO] = sbo;
    }
  }
  WSYNC();
  if (tb.DBG_SKIP == 101) return;   // diagnostics: time up to here (tools/time_implsch_v3.py)
  const float AIRD = c[C_AIRD], WSWAVE = c[C_WSWAVE], WSTAR = c[C_WSTAR], RAORW = c[C_RAORW], EMAXDPT = c[C_EMAXDPT], DEPTH = c[C_DEPTH];
  const float sinwd = c[C_SINWD], coswd = c[C_COSWD], CICOVER = c[C_SPARE];
  const V3F2 coswdif = L.costh * coswd + L.sinth * sinwd;
  const float frl = tb.FR[NFRE - 1];
  const float DELT25 = tb.WETAIL * frl * tb.DELTH;
  const V3F2 z2 = {0.f, 0.f};

  // weighted integrals over (K, M): per lane over M, then one all-reduce per pair of quantities
  auto fkmean3 = [&](float& EM, float& FM1, float& F1, float& AK, float& XK) {
    V3F2 s0 = z2, s1 = z2, s2 = z2;   // (EM, FM), (F1, AK), (XK, last row)
    for (int m = 0; m < NFRE; m++) {
      const V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
      const float t = L.grp ? f.x + f.y : 0.f;
      const float* fa = L.tFac + m * V3_NFAC;
      s0 = s0 + V3F2{lane_get(L.rDFIM, m), lane_get(L.rDFIMOFR, m)} * t;
      const float dfm = lane_get(L.rDFIM, m), sqm = fa[FA_SQ];
      s1 = s1 + V3F2{lane_get(L.rDFIMFR, m), dfm / sqm} * t;
      s2.x = s2.x + (sqm * dfm) * t;
      if (m == NFRE - 1) s2.y = t;
    }
    s0 = v3_allsum(s0, L.rot); s1 = v3_allsum(s1, L.rot); s2 = v3_allsum(s2, L.rot);
    const float COEFM1 = tb.FRTAIL * tb.DELTH;
    const float COEF1 = tb.WP1TAIL * tb.DELTH * frl * frl;
    const float COEFA = COEFM1 * m_sqrt(tb.G) / tb.ZPI;
    const float COEFX = COEF1 * (tb.ZPI / m_sqrt(tb.G));
    const float tl = s2.y;
    EM = tb.EPSMIN + s0.x + DELT25 * tl;
    FM1 = EM / (tb.EPSMIN + s0.y + COEFM1 * tl);
    F1 = (tb.EPSMIN + s1.x + COEF1 * tl) / EM;
    AK = tb.EPSMIN + s1.y + COEFA * tl;
    AK = (EM / AK) * (EM / AK);
    XK = tb.EPSMIN + s2.x + COEFX * tl;
    XK = (XK / EM) * (XK / EM);
  };

  // ---- SDEPTHLIM (sdepthlim.F90:64-78, semean.F90:82-120)
  if (tb.LBIWBK) {
    V3F2 s = z2;
    for (int m = 0; m < NFRE; m++) {
      const V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
      const float t = L.grp ? f.x + f.y : 0.f;
      s.x = s.x + lane_get(L.rDFIM, m) * t;
      if (m == NFRE - 1) s.y = t;
    }
    s = v3_allsum(s, L.rot);
    const float EM = tb.EPSMIN + s.x + DELT25 * s.y;
    const float sc = m_min(EMAXDPT / EM, 1.0f);
    if (L.grp)
      for (int m = 0; m < NFRE; m++) {
        V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG) * sc;
        f.x = m_max(f.x, tb.EPSMIN); f.y = m_max(f.y, tb.EPSMIN);
        *reinterpret_cast<V3F2*>(L.tFw + m * NANG) = f;
      }
    WSYNC();
  }
  float EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN;
  fkmean3(EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN);
  const V3F2 cpos = {m_max(0.f, coswdif.x), m_max(0.f, coswdif.y)};
  const V3F2 FLM = ((1.0f - 0.9f * m_min(CICOVER, 0.99f)) * tb.FLMIN) * (cpos * cpos);
  if (L.grp) {   // sinflx.F90:124-128
    V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + (NFRE - 1) * NANG);
    f.x = m_max(f.x, FLM.x); f.y = m_max(f.y, FLM.y);
    *reinterpret_cast<V3F2*>(L.tFw + (NFRE - 1) * NANG) = f;
  }
  WSYNC();
  {  // orbital velocity / displacement integrals of the swell damping (sinput_ard.F90:213-222)
    V3F2 s = z2;
    for (int m = 0; m < NFRE; m++) {
      const V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
      const float t = L.grp ? f.x + f.y : 0.f;
      const float* fa = L.tFac + m * V3_NFAC;
      const float w2 = lane_get(L.rDFIM, m), sig = lane_get(L.rZPIFR, m);
      s = s + V3F2{w2 * (sig * sig), w2} * t;
    }
    s = v3_allsum(s, L.rot);
    if (L.grp && j == 0) { c[C_UORBT] = tb.EPSMIN + s.x; c[C_AORB] = tb.EPSMIN + s.y; c[C_EMEAN] = EMEAN; c[C_F1MEAN] = F1MEAN; }
  }
  if (tb.DBG_SKIP == 102) return;
  float UFRIC = c[C_UFRIC], Z0M = c[C_Z0M];

  // per-point cut-off index, RHOWG_DFIM weights of the rows below it, stress sums and the F(:,MIJ) integrals of TAU_PHI_HF
  auto femws_finish = [&](V3F2 wse, V3F2 wslast, float& FM, float& EMW) {
    const V3F2 s = v3_allsum(L.grp ? wse : z2, L.rot);
    const float t2 = v3_allsum(L.grp ? V3F2{wslast.x + wslast.y, 0.f} : z2, L.rot).x;
    const float em = tb.EPSMIN + s.x + DELT25 * t2;
    const float fm = tb.EPSMIN + s.y + (tb.FRTAIL * tb.DELTH) * t2;
    FM = em / fm;
    EMW = em;
  };
  auto frcutindex3 = [&](float FMEANWS, float UF) -> int {
    const float FPMH = tb.TAILFACTOR / tb.FR[0];
    const float FPPM = tb.TAILFACTOR_PM * tb.G / (tb.FRIC * tb.ZPIFR[0]);
    int MIJ = NFRE;
    if (CICOVER <= tb.CITHRSH_TAIL) {
      const float FPM4 = m_max(m_max(FMEANWS, FMEAN) * FPMH, FPPM / m_max(UF, tb.EPSMIN));
      MIJ = m_nint(m_log10(FPM4) * tb.FLOGSPRDM1) + 1;
      MIJ = MIJ < 1 ? 1 : (MIJ > NFRE ? NFRE : MIJ);
    }
    return MIJ;
  };
  auto rrh = [&](int m, int MIJ) -> float {   // RHOWGDFTH(M), zero above MIJ, halved at MIJ (frcutindex.F90:98-107)
    float r = 0.f;
    if (m + 1 <= MIJ) {
      r = tb.RHOWG_DFIM[m];
      if (m + 1 == MIJ && MIJ != NFRE) r = 0.5f * r;
    }
    return r;
  };
  auto post_stress = [&](int MIJ, V3F2 apl, bool phiwa) {
    V3F2 s = z2;
    float sp = 0.f;
    if (L.grp)
      for (int m = j; m < NFRE; m += V3G) {
        const float w = rrh(m, MIJ);
        const float* r = L.tRow + m * 3;
        const float wx = w * L.tFac[m * V3_NFAC + FA_CINV];
        s = s + V3F2{wx * r[0], wx * r[1]};
        sp += w * r[2];
      }
    s = v3_allsum(s, L.rot);
    float PH = 0.f;
    if (phiwa) PH = v3_allsum(L.grp ? V3F2{apl.x + apl.y + sp, 0.f} : z2, L.rot).x;
    const V3F2 fm = L.grp ? *reinterpret_cast<const V3F2*>(L.tF + (MIJ - 1) * NANG) : z2;
    const V3F2 fc2 = fm * cpos * cpos, fc3 = fc2 * cpos;
    const V3F2 h = v3_allsum(V3F2{fc3.x + fc3.y, fc2.x + fc2.y}, L.rot);
    if (L.grp && j == 0) {
      c[C_XS] = s.x; c[C_YS] = s.y; c[C_F1DCOS3] = tb.DELTH * h.x; c[C_F1DCOS2] = tb.DELTH * h.y; c[C_F1DSIN2] = 0.f; c[C_F1D] = 0.f;
      c[C_MIJ] = (float)MIJ;
      if (phiwa) c[C_PHIWA] = PH;
    }
  };

  // ---- first SINFLX call (sinflx.F90:105-183): MIJ and the wave stress only
  unsigned long long xm0, xm1;
  V3F2 wse, wslast, apl;
  float FMEANWS, EMW;
  v3_sinput<1, false>(tb, L, UFRIC, Z0M, RAORW, 0.f, 0.f, 0.f, 0.f, sinwd, coswd, nullptr, xm0, xm1, wse, wslast, apl);
  femws_finish(wse, wslast, FMEANWS, EMW);
  int MIJ = frcutindex3(FMEANWS, UFRIC);
  post_stress(MIJ, apl, false);
  WSYNC();
  if (tb.DBG_SKIP == 103) return;
  // ---- stage 2: STRESSO scalars, second TAUT_Z0, WSIGSTAR, swell set-up, SDIWBK
  stresso_stage<float, V3P, false>(tb, sSC, lane, false);
  WSYNC();
  if (lane < V3P) {
    float* q = sSC + lane * NSC;
    float UF = q[C_UFRIC], Z0 = q[C_Z0M], Z0Bv = q[C_Z0B], CH = q[C_CHRNCK];
    taut_z0_c(tb, 1, q[C_WSWAVE], q[C_COSWD] * q[C_TWCOS] + q[C_SINWD] * q[C_TWSIN], q[C_TAUW], UF, Z0, Z0Bv, CH);
    q[C_UFRIC] = UF; q[C_Z0M] = Z0; q[C_Z0B] = Z0Bv; q[C_CHRNCK] = CH;
    q[C_SIGN] = wsigstar(tb, q[C_WSWAVE], UF, Z0, q[C_WSTAR]);
    swell_setup_pt(tb, q);
    q[C_SDS] = sdiwbk_pt(tb, q[C_EMAXDPT], q[C_EMEAN], q[C_F1MEAN], q[C_DEPTH]);
  }
  WSYNC();
  UFRIC = c[C_UFRIC]; Z0M = c[C_Z0M];
  const float SDS = c[C_SDS];
  if (tb.DBG_SKIP == 104) return;
  // ---- second SINFLX call: wind-input coefficient (parked in the XLLWS rows), XLLWS, MIJ, wave stress, PHIWA integrals
  float* gx = xllws + (size_t)ij * N + 2 * j;   // this lane's pair in row 0 of the point's XLLWS block, tile order [M][18][2]
  v3_sinput<2, true>(tb, L, UFRIC, Z0M, RAORW, c[C_SIGN], c[C_TEMP2], c[C_PTURB], c[C_AIRDPVISC], sinwd, coswd, gx, xm0, xm1, wse, wslast, apl);
  femws_finish(wse, wslast, FMEANWS, EMW);
  MIJ = frcutindex3(FMEANWS, UFRIC);
  post_stress(MIJ, apl, true);
  WSYNC();
  if (tb.DBG_SKIP == 105) return;
  // ---- stage 3: STRESSO of the second call (TAUW, TAUWDIR, PHIWA)
  stresso_stage<float, V3P, false>(tb, sSC, lane, true);
  WSYNC();

  if (tb.DBG_SKIP == 106) return;
  // ---- SDISSIP + SNONLIN + update sweep (implsch.F90:262-392)
  V3F2 a_t = z2, a_x = z2;
  {
    float ENHFR = m_max(0.75f * DEPTH * AKMEAN, 0.5f);
    ENHFR = 1.0f + (5.5f / ENHFR) * (1.0f - 0.833f * ENHFR) * m_exp(-1.25f * ENHFR);
    const int MFR1STFR = -tb.MFRSTLW + 1;
    const int MFRLSTFR = NFRE - tb.KFRH + MFR1STFR;
    const float DAL1 = tb.DAL1, DAL2 = tb.DAL2;
    int a1[2], a2[2], a11[2], a21[2], ai1[2], ai2[2], ai1s[2], ai2s[2];
    unsigned sw = 0;
#define V3_SRC(dst, bit, K)                         \
  {                                                 \
    const int s_ = (K);                             \
    dst = 4 * (base + (s_ >= V3G ? s_ - V3G : s_)); \
    sw |= (s_ >= V3G ? 1u : 0u) << (bit);           \
  }
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      V3_SRC(a1[kh], 8 * kh + 0, tb.K1W[kh][j]);
      V3_SRC(a2[kh], 8 * kh + 1, tb.K2W[kh][j]);
      V3_SRC(a11[kh], 8 * kh + 2, tb.K11W[kh][j]);
      V3_SRC(a21[kh], 8 * kh + 3, tb.K21W[kh][j]);
      V3_SRC(ai1[kh], 8 * kh + 4, tb.IK1[kh][j]);
      V3_SRC(ai2[kh], 8 * kh + 5, tb.IK2[kh][j]);
      const int c1 = j - tb.D11[kh], c2 = j - tb.D21[kh];
      V3_SRC(ai1s[kh], 8 * kh + 6, tb.IK1[kh][c1 < 0 ? c1 + NANG : c1]);
      V3_SRC(ai2s[kh], 8 * kh + 7, tb.IK2[kh][c2 < 0 ? c2 + NANG : c2]);
    }
#undef V3_SRC
    // saturation filter of SDISSIP_ARD: rotated reads of the pair tile
    int o0[V3_MAXTAP], o1[V3_MAXTAP];
    V3F2 wgt[V3_MAXTAP];
    const int ntap = tb.NTAP;
#pragma unroll
    for (int t = 0; t < V3_MAXTAP; t++) {
      const int tt = t < ntap ? t : 0;
      const int k0 = tb.INDICESSAT[tt][j], k1 = tb.INDICESSAT[tt][j + V3G];
      o0[t] = 2 * (k0 % V3G) + k0 / V3G;
      o1[t] = 2 * (k1 % V3G) + k1 / V3G;
      wgt[t] = t < ntap ? V3F2{tb.SATWEIGHTS[tt][j], tb.SATWEIGHTS[tt][j + V3G]} : z2;
    }
    const float TMP03 = 1.0f / (tb.SDSBR * tb.MICHE), SSDSC4 = tb.SSDSC4;
    const float c2 = tb.SSDSC2 * tb.SSDSC6, c2m1 = tb.SSDSC2 * (1.0f - tb.SSDSC6);
    const bool turb = tb.SSDSC5 != 0.f;
    const float FACTURB = turb ? (2.0f * tb.SSDSC5 / tb.G) * RAORW * UFRIC * UFRIC : 0.f;
    const float DELT = (float)tb.IDELT, DELTM = 1.0f / DELT, DELT5 = tb.XIMP * DELT;
    const bool shallow_brk = tb.LBIWBK && (DEPTH < 50.0f);
    const float USFM = UFRIC * m_max(FMEANWS, FMEAN);
    const bool flux_snl = tb.LCFLX && tb.LWVFLX_SNL;

    V3F2 aS[8], aF[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { aS[i] = z2; aF[i] = z2; }
    // wind-input rows (parked in the XLLWS block by the second SINFLX call) come back through a ring of eight prefetched rows:
    // slot jj holds row MCb + jj - 4 while block MCb is processed and is refilled with row MCb + jj + 4 as soon as it is consumed
    V3F2 wiq[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int mr = i - 4 < 0 ? i + 4 : i - 4;   // rows 4..7 in slots 0..3 (block 0 updates rows -4..3: slots 4..7 hold rows 0..3)
      wiq[i] = (L.act && mr < NFRE) ? *reinterpret_cast<const V3F2*>(gx + (size_t)mr * NANG) : z2;
    }
    for (int MCb = 0; MCb < tb.MLSTHG + 4; MCb += 8) {
#pragma unroll
      for (int jj = 0; jj < 8; jj++) {
        const int MC = MCb + 1 + jj;
        const int c0 = (1 + jj) & 7, cm = (1 + jj + 4) & 7, cm1 = (1 + jj + 5) & 7, cp = (1 + jj + 2) & 7, cp1 = (1 + jj + 3) & 7;
        if (MC <= tb.MLSTHG) {
          const int IC = tb.INLCOEF[MC - 1][0], IP = tb.INLCOEF[MC - 1][1], IP1 = tb.INLCOEF[MC - 1][2];
          const int IM = tb.INLCOEF[MC - 1][3], IM1 = tb.INLCOEF[MC - 1][4];
          const float* R = tb.RNLCOEF[MC - 1];
          const float FTAIL = R[0], GW1 = R[1], GW2 = R[2], GW3 = R[3], GW4 = R[4];
          const float FKLAMPA = R[5], FKLAMPB = R[6], FKLAMP2 = R[7], FKLAMP1 = R[8];
          const float FKLAPA2 = R[9], FKLAPB2 = R[10], FKLAP12 = R[11], FKLAP22 = R[12];
          const float GW5 = R[13], GW6 = R[14], GW7 = R[15], GW8 = R[16];
          const float FKLAMMA = R[17], FKLAMMB = R[18], FKLAMM2 = R[19], FKLAMM1 = R[20];
          const float FKLAMA2 = R[21], FKLAMB2 = R[22], FKLAM12 = R[23], FKLAM22 = R[24];
          const float FTEMP = tb.AF11[MC - 1] * ENHFR;
          const bool mid = (MC > MFR1STFR && MC < MFRLSTFR);
          const V3F2 fIP = *reinterpret_cast<const V3F2*>(L.tF + IP * NANG), fIP1 = *reinterpret_cast<const V3F2*>(L.tF + IP1 * NANG);
          const V3F2 fIM = *reinterpret_cast<const V3F2*>(L.tF + IM * NANG), fIM1 = *reinterpret_cast<const V3F2*>(L.tF + IM1 * NANG);
          V3F2 FIJ = *reinterpret_cast<const V3F2*>(L.tF + IC * NANG);
          if (!mid) FIJ = FIJ * FTAIL;
          const V3F2 up = GW1 * fIP + GW3 * fIP1, vp = GW2 * fIP + GW4 * fIP1;
          const V3F2 um = GW5 * fIM + GW7 * fIM1, vm = GW6 * fIM + GW8 * fIM1;
#pragma unroll
          for (int kh = 0; kh < 2; kh++) {
            const unsigned b = sw >> (8 * kh);
            const V3F2 SAP = v3_pull(up, a1[kh], b & 1u) + v3_pull(vp, a11[kh], b & 4u);
            const V3F2 SAM = v3_pull(um, a2[kh], b & 2u) + v3_pull(vm, a21[kh], b & 8u);
            V3F2 FAD1 = FIJ * (SAP + SAM);
            const V3F2 FAD2 = FAD1 - 2.0f * SAP * SAM;
            FAD1 = FAD1 + FAD2;
            const V3F2 FCEN = FTEMP * FIJ;
            const V3F2 AD = FAD2 * FCEN;
            const V3F2 DELAD = FAD1 * FTEMP;
            const V3F2 DELAP = (FIJ - 2.0f * SAM) * DAL1 * FCEN;
            const V3F2 DELAM = (FIJ - 2.0f * SAP) * DAL2 * FCEN;
            const V3F2 A2 = v3_pull(AD, ai2[kh], b & 32u), D2 = v3_pull(DELAM, ai2[kh], b & 32u);
            const V3F2 A1 = v3_pull(AD, ai1[kh], b & 16u), P1 = v3_pull(DELAP, ai1[kh], b & 16u);
            const V3F2 A2s = v3_pull(AD, ai2s[kh], b & 128u), D2s = v3_pull(DELAM, ai2s[kh], b & 128u);
            const V3F2 A1s = v3_pull(AD, ai1s[kh], b & 64u), P1s = v3_pull(DELAP, ai1s[kh], b & 64u);
            aS[c0] -= 2.0f * AD;
            aF[c0] -= 2.0f * DELAD;
            aS[cm] += A2 * FKLAMM1 + A2s * FKLAMM2;
            aF[cm] += D2 * FKLAM12 + D2s * FKLAM22;
            aS[cm1] += A2 * FKLAMMA + A2s * FKLAMMB;
            aF[cm1] += D2 * FKLAMA2 + D2s * FKLAMB2;
            aS[cp] += A1 * FKLAMP1 + A1s * FKLAMP2;
            aF[cp] += P1 * FKLAP12 + P1s * FKLAP22;
            aS[cp1] += A1 * FKLAMPA + A1s * FKLAMPB;
            aF[cp1] += P1 * FKLAPA2 + P1s * FKLAPB2;
          }
        }
        const int m = MC - 5;  // 0-based row MC-4: no later interaction reads or feeds it
        if (m >= 0 && m < NFRE) {
          const float* row = L.tP + m * NANG;
          const float* fa = L.tFac + m * V3_NFAC;
          const V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
          // dissipation coefficient of the row (sdissip_ard.F90:117-314), from the not yet updated row
          V3F2 bsat = z2;
#pragma unroll
          for (int t = 0; t < V3_MAXTAP; t++) bsat = bsat + wgt[t] * V3F2{row[o0[t]], row[o1[t]]};
          bsat = bsat * (fa[FA_WAVNUM] * (1.0f / tb.ZPI) * fa[FA_XK2CG]);
          const float bm = v3_allmax(L.grp ? fmaxf(bsat.x, bsat.y) : 0.f, L.rot);
          const float sig = lane_get(L.rZPIFR, m);
          const float d0 = m_max(0.f, bm * TMP03 - SSDSC4);
          const V3F2 t1 = bsat * TMP03 - SSDSC4;
          const V3F2 d1 = {m_max(0.f, t1.x), m_max(0.f, t1.y)};
          V3F2 D = (c2 * sig) * (d0 * d0) + (c2m1 * sig) * (d1 * d1);
          if (turb) D = D - (sig * fa[FA_WAVNUM] * FACTURB) * coswdif;
          const V3F2 wi = wiq[jj];   // wind input of the second SINFLX call
          const V3F2 fldw = D + wi;
          V3F2 sl = fldw * f + aS[cm];
          V3F2 fld = fldw + aF[cm];
          V3F2 ss = z2;
          if (flux_snl) {
            const V3F2 den = {m_max(1.0f - DELT5 * fld.x, 1.0f), m_max(1.0f - DELT5 * fld.y, 1.0f)};
            ss = V3F2{f_div(sl.x, den.x), f_div(sl.y, den.y)};
          }
          if (shallow_brk && m < tb.NFRE_RED) { sl = sl - SDS * f; fld = fld - SDS; }
          if (m < tb.NFRE_RED) { const float sbo = fa[FA_SBO]; sl = sl + sbo * f; fld = fld + sbo; }
          const float lim = USFM * (lane_get(L.rCOFRM4, m) * DELT), flmax = lane_get(L.rFLMAX, m);
          V3F2 fn;
          {
            const float G0 = f_div(DELT * sl.x, m_max(1.0f - DELT5 * fld.x, 1.0f)), G1 = f_div(DELT * sl.y, m_max(1.0f - DELT5 * fld.y, 1.0f));
            fn.x = m_max(f.x + m_sign(m_min(m_abs(G0), lim), G0), FLM.x);
            fn.y = m_max(f.y + m_sign(m_min(m_abs(G1), lim), G1), FLM.y);
          }
          ss.x = ss.x + DELTM * m_min(flmax - fn.x, 0.f);
          ss.y = ss.y + DELTM * m_min(flmax - fn.y, 0.f);
          fn.x = m_min(fn.x, flmax); fn.y = m_min(fn.y, flmax);
          if (L.grp) *reinterpret_cast<V3F2*>(L.tFw + m * NANG) = fn;
          float rh = 0.f;   // RHOWGDFTH(M) (frcutindex.F90:98-107); MIJ differs between the three points
          if (m + 1 <= MIJ) { rh = lane_get(L.rRHOWG, m); if (m + 1 == MIJ && MIJ != NFRE) rh = 0.5f * rh; }
          a_t = a_t + rh * ss;
          a_x = a_x + (fa[FA_CINV] * rh) * ss;
        }
        aS[cm] = z2;
        aF[cm] = z2;
        {
          const int mn = MC - 5 + 8;
          if (mn >= 0 && mn < NFRE && L.act) wiq[jj] = *reinterpret_cast<const V3F2*>(gx + (size_t)mn * NANG);
        }
      }
    }
  }
  WSYNC();
  const float TAUW = c[C_TAUW], TAUWDIR = c[C_TAUWDIR], PHIWA = c[C_PHIWA], Z0B = c[C_Z0B], CHRNCK = c[C_CHRNCK];

  if (tb.DBG_SKIP == 107) return;
  // ---- WNFLUXES (wnfluxes.F90:147-330), LWNEMOCOU = F
  float TAUXD = 0.f, TAUYD = 0.f, TAUOCXD = 0.f, TAUOCYD = 0.f, TAUOC = 0.f, PHIOCD = 0.f, PHIEPS = 0.f, PHIAW = 0.f;
  if (tb.LCFLX) {
    const V3F2 sx = a_x * L.sinth, sy = a_x * L.costh;
    const V3F2 r0 = v3_allsum(L.grp ? V3F2{a_t.x + a_t.y, sx.x + sx.y} : z2, L.rot);
    const float YSTRESS = v3_allsum(L.grp ? V3F2{sy.x + sy.y, 0.f} : z2, L.rot).x;
    const float PHILF = r0.x, XSTRESS = r0.y;
    const float EPSUS3 = tb.EPSUS * m_sqrt(tb.EPSUS);
    float OOVAL = 1.0f, USTAR = UFRIC;
    if (tb.LICERUN && tb.LWAMRSETCI && CICOVER > tb.CIBLOCK) {
      OOVAL = m_exp(-m_min(m_pow4(CICOVER * (1.0f / m_max(tb.CITHRSH, 0.01f))), 10.0f));
      const float U10P = m_max(WSWAVE, tb.EPSU10);
      const float CD_BULK = m_min((1.03E-3f + 0.04E-3f * m_pow(U10P, 1.48f)) * m_pow(U10P, -0.21f), 0.003f);
      const float CD_WAVE = (UFRIC / U10P) * (UFRIC / U10P);
      const float CD_ICE = OOVAL * CD_WAVE + (1.0f - OOVAL) * CD_BULK;
      USTAR = m_max(m_sqrt(CD_ICE) * U10P, tb.EPSUS);
    }
    const float TAU = AIRD * m_max(USTAR * USTAR, tb.EPSUS);
    TAUXD = TAU * sinwd;
    TAUYD = TAU * coswd;
    TAUOCXD = TAUXD - OOVAL * XSTRESS;
    TAUOCYD = TAUYD - OOVAL * YSTRESS;
    const float TAUO = m_sqrt(TAUOCXD * TAUOCXD + TAUOCYD * TAUOCYD);
    TAUOC = m_min(m_max(TAUO / TAU, tb.TAUOCMIN), tb.TAUOCMAX);
    const float USTRA = ffa[(size_t)ij * ECWAM_HIP_NFF + 5], VSTRA = ffa[(size_t)ij * ECWAM_HIP_NFF + 6];
    if (tb.LWCOUAST && (USTRA != 0.f || VSTRA != 0.f)) { TAUXD = USTRA; TAUOCXD = USTRA * TAUOC; TAUYD = VSTRA; TAUOCYD = VSTRA * TAUOC; }
    const float XN = AIRD * m_max(USTAR * USTAR * USTAR, EPSUS3);
    PHIOCD = OOVAL * (PHILF - PHIWA) + (1.0f - OOVAL) * (-3.75f) * XN;
    PHIEPS = m_min(m_max(PHIOCD / XN, tb.PHIEPSMIN), tb.PHIEPSMAX);
    PHIOCD = PHIEPS * XN;
    PHIAW = OOVAL * PHIWA / XN + (1.0f - OOVAL) * 3.75f;
  }

  // ---- second FKMEAN / FEMEANWS, IMPHFTAIL, SETICE, STOKESDRIFT (implsch.F90:422-462)
  fkmean3(EMEAN, FMEAN, F1MEAN, AKMEAN, XKMEAN);
  float EMEANWS;
  {
    V3F2 we = z2, wl = z2;
    for (int m = 0; m < NFRE; m++) {
      const V3F2 f = *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
      const V3F2 x = {((xm0 >> m) & 1ull) ? f.x : 0.f, ((xm1 >> m) & 1ull) ? f.y : 0.f};
      const float* fa = L.tFac + m * V3_NFAC;
      we = we + V3F2{lane_get(L.rDFIM, m), lane_get(L.rDFIMOFR, m)} * (x.x + x.y);
      wl = x;
    }
    femws_finish(we, wl, FMEANWS, EMEANWS);
  }
  if (L.grp) {  // imphftail.F90
    auto rt = [&](int m) { const float* fa = L.tFac + m * V3_NFAC; return 1.0f / fa[FA_XK2CG] / fa[FA_WAVNUM]; };
    const float T1 = rt(MIJ - 1);
    const V3F2 tf = *reinterpret_cast<const V3F2*>(L.tF + (MIJ - 1) * NANG);
    for (int m = MIJ; m < NFRE; m++) {
      const float tm = rt(m) / T1;
      *reinterpret_cast<V3F2*>(L.tFw + m * NANG) = V3F2{m_max(tm * tf.x, FLM.x), m_max(tm * tf.y, FLM.y)};
    }
  }
  if (tb.LICERUN && tb.LMASKICE && L.grp) {  // setice.F90:67-86
    float CIREDUC, ICEFREE;
    if (CICOVER > tb.CITHRSH) { CIREDUC = m_max(tb.EPSMIN, 1.0f - CICOVER); ICEFREE = 0.f; }
    else { CIREDUC = 0.f; ICEFREE = 1.f; }
    const V3F2 add = (CIREDUC * tb.FLMIN) * (cpos * cpos);
    for (int m = 0; m < NFRE; m++) *reinterpret_cast<V3F2*>(L.tFw + m * NANG) = *reinterpret_cast<const V3F2*>(L.tF + m * NANG) * ICEFREE + add;
  }
  float USTOKES, VSTOKES;
  {  // stokesdrift.F90:89-142
    const int MO = tb.NFRE_ODD;
    const float fo = tb.FR[MO - 1];
    const float CONST = 2.0f * tb.DELTH * (tb.ZPI * tb.ZPI * tb.ZPI) / tb.G * m_pow4(fo);
    V3F2 a = z2;
    for (int m = 0; m < MO; m++) a = a + L.tFac[m * V3_NFAC + FA_STOK] * *reinterpret_cast<const V3F2*>(L.tF + m * NANG);
    a = a + CONST * *reinterpret_cast<const V3F2*>(L.tF + (MO - 1) * NANG);
    const V3F2 ax = a * L.sinth, ay = a * L.costh;
    const V3F2 s = v3_allsum(L.grp ? V3F2{ax.x + ax.y, ay.x + ay.y} : z2, L.rot);
    USTOKES = s.x; VSTOKES = s.y;
    if (tb.LICERUN && tb.LWAMRSETCI && CICOVER > tb.CITHRSH) {
      USTOKES = 0.016f * WSWAVE * sinwd * (1.0f - CICOVER);
      VSTOKES = 0.016f * WSWAVE * coswd * (1.0f - CICOVER);
    }
    USTOKES = m_min(m_max(USTOKES, -1.5f), 1.5f);
    VSTOKES = m_min(m_max(VSTOKES, -1.5f), 1.5f);
  }
  WSYNC();
  // ---- store FL1 (coalesced, from the pair tile), XLLWS from tile order to [K][M] in place is done by the caller's layout
  //      conversion below; per-point scalars
  {
    const float rnf = 1.0f / (float)NFRE;
    float* g = fl1 + (size_t)ij0 * N;
    for (int e = lane; e < n * N; e += 64) {
      const int kq = (int)(((float)e + 0.5f) * rnf);
      const int m = e - kq * NFRE;
      const int pq = (kq >= 2 * NANG) ? 2 : (kq >= NANG ? 1 : 0), k = kq - pq * NANG;
      g[e] = sF[pq * N + m * NANG + 2 * (k >= V3G ? k - V3G : k) + (k >= V3G ? 1 : 0)];
    }
  }
  if (L.act) {   // XLLWS(K,M) of the second SINFLX call: every wind-input row parked in this block has been read by now
    float* x0 = xllws + (size_t)ij * N + (size_t)j * NFRE;
    float* x1 = x0 + (size_t)V3G * NFRE;
    for (int m = 0; m < NFRE; m++) { x0[m] = ((xm0 >> m) & 1ull) ? 1.f : 0.f; x1[m] = ((xm1 >> m) & 1ull) ? 1.f : 0.f; }
  }
  if (L.act && j == 0) {
    float* fo = ffa + (size_t)ij * ECWAM_HIP_NFF;
    fo[7] = UFRIC; fo[8] = TAUW; fo[9] = TAUWDIR; fo[10] = Z0M; fo[11] = Z0B; fo[12] = CHRNCK;
    float* io = intfa + (size_t)ij * ECWAM_HIP_NINTF;
    io[2] = USTOKES; io[3] = VSTOKES;
    if (tb.LCFLX) {
      io[5] = TAUXD; io[6] = TAUYD; io[7] = TAUOCXD; io[8] = TAUOCYD; io[9] = TAUOC; io[10] = 0.f; io[11] = 0.f;
      io[12] = PHIOCD; io[13] = PHIEPS; io[14] = PHIAW;
    }
    if (tb.LWFLUX) {
      io[0] = (EMEANWS < tb.WSEMEAN_MIN) ? tb.WSEMEAN_MIN : EMEANWS;
      io[1] = (EMEANWS < tb.WSEMEAN_MIN) ? 2.0f * tb.FR[NFRE - 1] : FMEANWS;
    }
    mij_out[ij] = MIJ;
  }
}
