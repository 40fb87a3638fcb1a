// SNONLIN (discrete interaction approximation, snonlin.F90:100-330) in the lane layout planned for the next IMPLSCH kernel
// generation: THREE sea points per wavefront, 18 lanes per point, lane j of a point holds the direction
// pair (K = j, j + 18) as one packed-fp32 operand -- 54 of 64 lanes carry data and every arithmetic instruction works on 108
// spectral values (k_implsch2: 36 lanes, 36 or 72 values).  Single precision, NANG = 36 only.
//
// A rotation of the directions by r (K1W/K2W/K11W/K21W and the inverse maps IK1/IK2 of the pull-form DIA, dev.h) moves both
// directions of a lane to the SAME source lane (j + r mod 18 within the point's 18 lanes) and swaps the two halves when the
// rotation crosses K = 18 an odd number of times: one ds_bpermute per half plus a select, per-lane constants.
//
// This file is a self-contained entry point (the reference's SNONLIN seam, SL and FLD starting from zero) used to validate the
// layout (tests/test_gpu_parity.py) and to time the DIA in it against k_implsch2's (tools/time_snonlin3.py); IMPLSCH itself still
// runs k_implsch2.
#include <hip/hip_runtime.h>

#include "dev.h"

typedef float F2 __attribute__((ext_vector_type(2)));

#define S3_GROUP 18
#define S3_PTS 3

__device__ __forceinline__ float s3_bperm(int addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v)));
}
// value of the direction pair that the rotation maps onto this lane's pair
__device__ __forceinline__ F2 s3_pull(F2 v, int addr, bool swap) {
  const float a = s3_bperm(addr, v.x), b = s3_bperm(addr, v.y);
  F2 r;
  r.x = swap ? b : a;
  r.y = swap ? a : b;
  return r;
}

// mode 0: DIA; mode 1: load / store only (timing baseline: SL = F, FLD = 0)
template <int WPB, int MODE>
__global__ void __launch_bounds__(64 * WPB) k_snonlin3(const DevTab<float>* __restrict__ tp, int n, const float* __restrict__ fl1,
                                                       const float* __restrict__ depth, const float* __restrict__ akmean,
                                                       float* __restrict__ sl, float* __restrict__ fld) {
  extern __shared__ __align__(16) unsigned char s3_smem[];
  const DevTab<float>& tb = *tp;
  const int NANG = tb.NANG, NFRE = tb.NFRE, N = NANG * NFRE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = lane / S3_GROUP, j = lane - p * S3_GROUP;
  const int ij0 = (blockIdx.x * WPB + wave) * S3_PTS;
  if (ij0 >= n) return;  // no block-level synchronisation in this kernel
  const bool act = (p < S3_PTS) && (ij0 + p < n);
  const int ij = act ? ij0 + p : ij0;
  float* sF = reinterpret_cast<float*>(s3_smem) + (size_t)wave * S3_PTS * N;  // [point][M][18][2]: pair (K=j, K=j+18) interleaved
  // ---- F[ij][K][M] (coalesced) -> LDS
  for (int q = 0; q < S3_PTS; q++) {
    if (ij0 + q >= n) break;
    const float* g = fl1 + (size_t)(ij0 + q) * N;
    float* t = sF + q * N;
    for (int e = lane; e < N; e += 64) {
      const int k = e / NFRE, m = e - k * NFRE;
      t[m * NANG + 2 * (k % S3_GROUP) + k / S3_GROUP] = g[e];
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tile is written before it is read
  const float* tF = sF + (act ? p : 0) * N + 2 * j;
  const size_t orow = (size_t)ij * NFRE * NANG + 2 * j;   // outputs in the tile order [ij][M][18][2]
  if (MODE == 1) {
    if (act)
      for (int m = 0; m < NFRE; m++) {
        const F2 f = *reinterpret_cast<const F2*>(tF + m * NANG);
        *reinterpret_cast<F2*>(sl + orow + (size_t)m * NANG) = f;
        *reinterpret_cast<F2*>(fld + orow + (size_t)m * NANG) = F2{0.f, 0.f};
      }
    return;
  }
  // ---- per-lane rotation sources: lane address (x4 for ds_bpermute) and half swap
  const int base = (p < S3_PTS ? p : 0) * S3_GROUP;
  int a1[2], a2[2], a11[2], a21[2], ai1[2], ai2[2], ai1s[2], ai2s[2];
  unsigned sw = 0;
#define S3_SRC(dst, bit, K)                                 \
  {                                                         \
    const int s_ = (K);                                     \
    dst = 4 * (base + (s_ >= S3_GROUP ? s_ - S3_GROUP : s_)); \
    sw |= (s_ >= S3_GROUP ? 1u : 0u) << (bit);              \
  }
#pragma unroll
  for (int kh = 0; kh < 2; kh++) {
    S3_SRC(a1[kh], 8 * kh + 0, tb.K1W[kh][j]);
    S3_SRC(a2[kh], 8 * kh + 1, tb.K2W[kh][j]);
    S3_SRC(a11[kh], 8 * kh + 2, tb.K11W[kh][j]);
    S3_SRC(a21[kh], 8 * kh + 3, tb.K21W[kh][j]);
    S3_SRC(ai1[kh], 8 * kh + 4, tb.IK1[kh][j]);
    S3_SRC(ai2[kh], 8 * kh + 5, tb.IK2[kh][j]);
    const int c1 = j - tb.D11[kh], c2 = j - tb.D21[kh];
    S3_SRC(ai1s[kh], 8 * kh + 6, tb.IK1[kh][c1 < 0 ? c1 + NANG : c1]);
    S3_SRC(ai2s[kh], 8 * kh + 7, tb.IK2[kh][c2 < 0 ? c2 + NANG : c2]);
  }
#undef S3_SRC
  // shallow-water enhancement (snonlin.F90:127-136, ISNONLIN = 0)
  const float DEPTH = depth[ij], AKMEAN = akmean[ij];
  float ENHFR = fmaxf(0.75f * DEPTH * AKMEAN, 0.5f);
  ENHFR = 1.0f + (5.5f / ENHFR) * (1.0f - 0.833f * ENHFR) * __expf(-1.25f * ENHFR);
  const int MFR1STFR = -tb.MFRSTLW + 1;
  const int MFRLSTFR = NFRE - tb.KFRH + MFR1STFR;
  const float DAL1 = tb.DAL1, DAL2 = tb.DAL2;

  F2 aS[8], aF[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { aS[i] = F2{0.f, 0.f}; aF[i] = F2{0.f, 0.f}; }
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
        const F2 fIP = *reinterpret_cast<const F2*>(tF + IP * NANG), fIP1 = *reinterpret_cast<const F2*>(tF + IP1 * NANG);
        const F2 fIM = *reinterpret_cast<const F2*>(tF + IM * NANG), fIM1 = *reinterpret_cast<const F2*>(tF + IM1 * NANG);
        F2 FIJ = *reinterpret_cast<const F2*>(tF + IC * NANG);
        if (!mid) FIJ = FIJ * FTAIL;
        const F2 up = GW1 * fIP + GW3 * fIP1, vp = GW2 * fIP + GW4 * fIP1;
        const F2 um = GW5 * fIM + GW7 * fIM1, vm = GW6 * fIM + GW8 * fIM1;
#pragma unroll
        for (int kh = 0; kh < 2; kh++) {
          const unsigned b = sw >> (8 * kh);
          const F2 SAP = s3_pull(up, a1[kh], b & 1u) + s3_pull(vp, a11[kh], b & 4u);
          const F2 SAM = s3_pull(um, a2[kh], b & 2u) + s3_pull(vm, a21[kh], b & 8u);
          F2 FAD1 = FIJ * (SAP + SAM);
          const F2 FAD2 = FAD1 - 2.0f * SAP * SAM;
          FAD1 = FAD1 + FAD2;
          const F2 FCEN = FTEMP * FIJ;
          const F2 AD = FAD2 * FCEN;
          const F2 DELAD = FAD1 * FTEMP;
          const F2 DELAP = (FIJ - 2.0f * SAM) * DAL1 * FCEN;
          const F2 DELAM = (FIJ - 2.0f * SAP) * DAL2 * FCEN;
          const F2 A2 = s3_pull(AD, ai2[kh], b & 32u), D2 = s3_pull(DELAM, ai2[kh], b & 32u);
          const F2 A1 = s3_pull(AD, ai1[kh], b & 16u), P1 = s3_pull(DELAP, ai1[kh], b & 16u);
          const F2 A2s = s3_pull(AD, ai2s[kh], b & 128u), D2s = s3_pull(DELAM, ai2s[kh], b & 128u);
          const F2 A1s = s3_pull(AD, ai1s[kh], b & 64u), P1s = s3_pull(DELAP, ai1s[kh], b & 64u);
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
      const int m = MC - 5;  // 0-based row MC-4: no later interaction feeds it
      if (m >= 0 && m < NFRE && act) {
        *reinterpret_cast<F2*>(sl + orow + (size_t)m * NANG) = aS[cm];
        *reinterpret_cast<F2*>(fld + orow + (size_t)m * NANG) = aF[cm];
      }
      aS[cm] = F2{0.f, 0.f};
      aF[cm] = F2{0.f, 0.f};
    }
  }
}

// returns 0, or 1 when the configuration is outside what this layout covers (NANG != 36, tables without the rotation structure)
int launch_snonlin3(const void* tab, int nang, int nfre, int dia_pull, int n, const void* fl1, const void* depth, const void* akmean,
                    void* sl, void* fld, int mode, hipStream_t s) {
  if (nang != 2 * S3_GROUP || !dia_pull) return 1;
  if (n <= 0) return 0;
  constexpr int WPB = 2;
  const size_t shmem = (size_t)WPB * S3_PTS * nang * nfre * sizeof(float);
  const int blocks = (n + WPB * S3_PTS - 1) / (WPB * S3_PTS);
  if (mode == 1)
    hipLaunchKernelGGL((k_snonlin3<WPB, 1>), dim3(blocks), dim3(64 * WPB), shmem, s, (const DevTab<float>*)tab, n, (const float*)fl1,
                       (const float*)depth, (const float*)akmean, (float*)sl, (float*)fld);
  else
    hipLaunchKernelGGL((k_snonlin3<WPB, 0>), dim3(blocks), dim3(64 * WPB), shmem, s, (const DevTab<float>*)tab, n, (const float*)fl1,
                       (const float*)depth, (const float*)akmean, (float*)sl, (float*)fld);
  return 0;
}

// =====================================================================================================================
// SINPUT_ARD (sinput_ard.F90:153-520) in the same lane layout: the second SINFLX call (NGST = 2, LLSNEG = T) with the
// sheltering recurrence (TAUWSHELTER /= 0), LLNORMAGAM = F.  Per row M the stress of each gust state over all 36
// directions is an all-reduce within the point's 18 lanes (no DPP: the groups straddle the 16-lane rows), two
// quantities per packed operand.
// in : pt[n][12] = UFRIC, Z0M, RAORW, SIG_N, TEMP2, PTURB, AIRD_PVISC, SIN(WDWAVE), COS(WDWAVE), -, -, -
//      wvprpt[n][5][NFRE] (WAVNUM, CGROUP, CINV, XK2CG, STOKFAC: the rows IMPLSCH receives)
// out: fld, spos, xllws [n][NFRE][18][2]; xys[n][NFRE][4] = SUM_K SPOS*SINTH, SUM_K SPOS*COSTH, SUM_K SPOS, -
// =====================================================================================================================
__device__ __forceinline__ F2 s3_same(F2 v, int addr) {   // both halves from the same lane, no swap
  F2 r;
  r.x = s3_bperm(addr, v.x);
  r.y = s3_bperm(addr, v.y);
  return r;
}
// sums over the 18 lanes of a point, every lane gets them: x and y are two independent quantities
__device__ __forceinline__ F2 s3_allsum(F2 v, int a9, int a3, int a6, int a1, int a2) {
  v = v + s3_same(v, a9);
  v = v + (s3_same(v, a3) + s3_same(v, a6));
  v = v + (s3_same(v, a1) + s3_same(v, a2));
  return v;
}

#define S3_NFAC 8
template <int WPB, int MODE>
__global__ void __launch_bounds__(64 * WPB) k_sinput3(const DevTab<float>* __restrict__ tp, int n, const float* __restrict__ fl1,
                                                      const float* __restrict__ wvprpt, const float* __restrict__ pt,
                                                      float* __restrict__ fld, float* __restrict__ spos, float* __restrict__ xllws,
                                                      float* __restrict__ xys) {
  extern __shared__ __align__(16) unsigned char s3_smem[];
  const DevTab<float>& tb = *tp;
  const int NANG = tb.NANG, NFRE = tb.NFRE, N = NANG * NFRE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = lane / S3_GROUP, j = lane - p * S3_GROUP;
  const int ij0 = (blockIdx.x * WPB + wave) * S3_PTS;
  if (ij0 >= n) return;
  const bool grp = p < S3_PTS;
  const bool act = grp && (ij0 + p < n);
  const int ij = act ? ij0 + p : ij0;
  float* sF = reinterpret_cast<float*>(s3_smem) + (size_t)wave * S3_PTS * (N + S3_NFAC * NFRE);
  float* sFac = sF + S3_PTS * N;   // [point][M][S3_NFAC]: ZCN, CNSN, CONSTF, DSTAB1, TEMP1, CINV, RHOWG_DFIM, -
  for (int q = 0; q < S3_PTS; q++) {
    if (ij0 + q >= n) break;
    const float* g = fl1 + (size_t)(ij0 + q) * N;
    float* t = sF + q * N;
    for (int e = lane; e < N; e += 64) {
      const int k = e / NFRE, m = e - k * NFRE;
      t[m * NANG + 2 * (k % S3_GROUP) + k / S3_GROUP] = g[e];
    }
  }
  // ---- per-point scalars, replicated over the point's lanes
  const float* q = pt + (size_t)ij * 12;
  const float UFRIC = q[0], Z0M = q[1], RAORW = q[2], SIG_N = q[3], TEMP2 = q[4], PTURB = q[5], AIRD_PVISC = q[6];
  const float sinwd = q[7], coswd = q[8];
  const float CONST1 = tb.BETAMAXOXKAPPA2, ABS_TAUWSHELTER = fabsf(tb.TAUWSHELTER);
  const float FU = fabsf(tb.SWELLF3), FUD = tb.SWELLF2, ROGOROAIR = tb.G / RAORW;
  // ---- per-frequency factors of the point (sinput_ard.F90:340-354, 379-388): lane j evaluates M = j+1 and j+19
  if (act) {
    const float* wp = wvprpt + (size_t)ij * ECWAM_HIP_NWPR * NFRE;
    for (int m = j; m < NFRE; m += S3_GROUP) {
      const float SIG = tb.ZPIFR[m], WAVNUM = wp[m], CINV = wp[2 * NFRE + m];
      float* f = sFac + ((size_t)p * NFRE + m) * S3_NFAC;
      f[0] = __logf(WAVNUM * Z0M);
      f[1] = (SIG * CONST1) * RAORW;
      f[2] = ROGOROAIR * CINV * tb.DFIM[m];
      f[3] = (-tb.SWELLF5 * 2.0f * f_sqrt(2.0f * tb.RNU * SIG)) * AIRD_PVISC * WAVNUM;
      f[4] = (-tb.SWELLF * 16.0f * (SIG * SIG) / tb.G) * RAORW;
      f[5] = CINV;
      f[6] = tb.RHOWG_DFIM[m];
      f[7] = 0.f;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float* tF = sF + (grp ? p : 0) * N + 2 * j;
  const float* tFac = sFac + (size_t)(grp ? p : 0) * NFRE * S3_NFAC;
  const size_t orow = (size_t)ij * NFRE * NANG + 2 * j;
  if (MODE == 1) {   // load / store only
    if (act)
      for (int m = 0; m < NFRE; m++) {
        const F2 f = *reinterpret_cast<const F2*>(tF + m * NANG);
        *reinterpret_cast<F2*>(fld + orow + (size_t)m * NANG) = f;
        *reinterpret_cast<F2*>(spos + orow + (size_t)m * NANG) = f;
        *reinterpret_cast<F2*>(xllws + orow + (size_t)m * NANG) = F2{0.f, 0.f};
        if (j == 0) *reinterpret_cast<float4*>(xys + ((size_t)ij * NFRE + m) * 4) = float4{0.f, 0.f, 0.f, 0.f};
      }
    return;
  }
  const int base = (grp ? p : 0) * S3_GROUP;
#define S3_ROT(r) (4 * (base + ((j + (r)) >= S3_GROUP ? j + (r) - S3_GROUP : j + (r))))
  const int a9 = S3_ROT(9), a3 = S3_ROT(3), a6 = S3_ROT(6), a1 = S3_ROT(1), a2 = S3_ROT(2);
#undef S3_ROT
  const F2 sinth = {tb.SINTH[j], tb.SINTH[j + S3_GROUP]}, costh = {tb.COSTH[j], tb.COSTH[j + S3_GROUP]};
  const float XKAPPA = tb.XKAPPA, ZALP = tb.ZALP;
  float USTP[2], XSTRESS[2] = {0.f, 0.f}, YSTRESS[2] = {0.f, 0.f}, TAUX[2], TAUY[2];
  USTP[0] = UFRIC * (1.0f + SIG_N);
  USTP[1] = UFRIC * (1.0f - SIG_N);
#pragma unroll
  for (int ig = 0; ig < 2; ig++) {
    const float USG2 = USTP[ig] * USTP[ig];
    TAUX[ig] = USG2 * sinwd;
    TAUY[ig] = USG2 * coswd;
  }
  for (int m = 0; m < NFRE; m++) {
    const float4 fa = *reinterpret_cast<const float4*>(tFac + m * S3_NFAC);      // ZCN, CNSN, CONSTF, DSTAB1
    const float2 fb = *reinterpret_cast<const float2*>(tFac + m * S3_NFAC + 4);   // TEMP1, CINV
    const float ZCN = fa.x, CNSN = fa.y, CONSTF = fa.z, DSTAB1 = fa.w, TEMP1 = fb.x, cinv_m = fb.y;
    const F2 f = *reinterpret_cast<const F2*>(tF + m * NANG);
    F2 SLP[2], FLP[2];
    bool xl0 = false, xl1 = false;   // XLLWS of the two halves
#pragma unroll
    for (int ig = 0; ig < 2; ig++) {
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
      const F2 coslp = costh * COSU + sinth * SINU;
      F2 gam0 = {0.f, 0.f};
      {
        const bool c0 = coslp.x > 0.01f, c1 = coslp.y > 0.01f;
        const float Z0 = ZCN + UCNZALPD * f_rcp(coslp.x), Z1 = ZCN + UCNZALPD * f_rcp(coslp.y);
        const bool n0 = c0 && (Z0 < 0.f), n1 = c1 && (Z1 < 0.f);
        if (__builtin_amdgcn_ballot_w64(n0 || n1) != 0ull) {
          const F2 ZL = {Z0, Z1};
          const F2 Z2X = ZL * ZL * (coslp * UCN);
          const F2 ex = {f_exp(Z0), f_exp(Z1)};
          const F2 g = ex * Z2X * Z2X * CNSN;
          gam0.x = n0 ? g.x : 0.f;
          gam0.y = n1 ? g.y : 0.f;
          xl0 = xl0 || n0;
          xl1 = xl1 || n1;
        }
      }
      const F2 DSTAB2 = TEMP1 * (TEMP2 + (FU + FUD * coslp) * USTP[ig]);
      const F2 dstab = DSTAB1 + PTURB * DSTAB2;
      FLP[ig] = gam0 + dstab;
      SLP[ig] = gam0 * f;
    }
    const F2 sp = 0.5f * (SLP[0] + SLP[1]);
    const F2 fl = 0.5f * (FLP[0] + FLP[1]);
    // directional integrals of the row: (X, Y) per gust state and (SUM SLP) of both, every lane of the point gets them
    F2 xs[2], ss;
    const bool anygrow = __builtin_amdgcn_ballot_w64(xl0 || xl1) != 0ull;
    if (anygrow) {
#pragma unroll
      for (int ig = 0; ig < 2; ig++) {
        const F2 sx = grp ? SLP[ig] * sinth : F2{0.f, 0.f}, sy = grp ? SLP[ig] * costh : F2{0.f, 0.f};
        xs[ig] = s3_allsum(F2{sx.x + sx.y, sy.x + sy.y}, a9, a3, a6, a1, a2);
        XSTRESS[ig] = XSTRESS[ig] + CONSTF * xs[ig].x;
        YSTRESS[ig] = YSTRESS[ig] + CONSTF * xs[ig].y;
      }
      ss = s3_allsum(grp ? F2{SLP[0].x + SLP[0].y, SLP[1].x + SLP[1].y} : F2{0.f, 0.f}, a9, a3, a6, a1, a2);
    } else {
      xs[0] = xs[1] = ss = F2{0.f, 0.f};
    }
    if (act) {
      *reinterpret_cast<F2*>(fld + orow + (size_t)m * NANG) = fl;
      *reinterpret_cast<F2*>(spos + orow + (size_t)m * NANG) = sp;
      *reinterpret_cast<F2*>(xllws + orow + (size_t)m * NANG) = F2{xl0 ? 1.f : 0.f, xl1 ? 1.f : 0.f};
      if (j == 0)
        *reinterpret_cast<float4*>(xys + ((size_t)ij * NFRE + m) * 4) =
            float4{0.5f * (xs[0].x + xs[1].x), 0.5f * (xs[0].y + xs[1].y), 0.5f * (ss.x + ss.y), 0.f};
    }
  }
}

int launch_sinput3(const void* tab, int nang, int nfre, int n, const void* fl1, const void* wvprpt, const void* pt, void* fld,
                   void* spos, void* xllws, void* xys, int mode, hipStream_t s) {
  if (nang != 2 * S3_GROUP) return 1;
  if (n <= 0) return 0;
  constexpr int WPB = 2;
  const size_t shmem = (size_t)WPB * S3_PTS * (nang * nfre + S3_NFAC * nfre) * sizeof(float);
  const int blocks = (n + WPB * S3_PTS - 1) / (WPB * S3_PTS);
#define S3_ARGS (const DevTab<float>*)tab, n, (const float*)fl1, (const float*)wvprpt, (const float*)pt, (float*)fld, (float*)spos, (float*)xllws, (float*)xys
  if (mode == 1) hipLaunchKernelGGL((k_sinput3<WPB, 1>), dim3(blocks), dim3(64 * WPB), shmem, s, S3_ARGS);
  else hipLaunchKernelGGL((k_sinput3<WPB, 0>), dim3(blocks), dim3(64 * WPB), shmem, s, S3_ARGS);
#undef S3_ARGS
  return 0;
}

// =====================================================================================================================
// SDISSIP_ARD (sdissip_ard.F90:117-314, SSDSC3 = 0) in the same lane layout.  The saturation filter reads the 2*NSDSNTH+1
// neighbouring directions straight from the pair tile at rotated addresses (one ds_read_b32 per half and tap: no half
// swap needed, the address of each half is a per-lane constant); the directional maximum of a row is an 18-lane all-reduce.
// in : pt[n][12] (UFRIC at 0, RAORW at 2, COS/SIN of WDWAVE at 8/7), wvprpt as above.  out: fld [n][NFRE][18][2].
// =====================================================================================================================
#define S3_MAXTAP 17
template <int WPB, int MODE>
__global__ void __launch_bounds__(64 * WPB) k_sdissip3(const DevTab<float>* __restrict__ tp, int n, const float* __restrict__ fl1,
                                                       const float* __restrict__ wvprpt, const float* __restrict__ pt,
                                                       float* __restrict__ fld) {
  extern __shared__ __align__(16) unsigned char s3_smem[];
  const DevTab<float>& tb = *tp;
  const int NANG = tb.NANG, NFRE = tb.NFRE, N = NANG * NFRE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = lane / S3_GROUP, j = lane - p * S3_GROUP;
  const int ij0 = (blockIdx.x * WPB + wave) * S3_PTS;
  if (ij0 >= n) return;
  const bool grp = p < S3_PTS;
  const bool act = grp && (ij0 + p < n);
  const int ij = act ? ij0 + p : ij0;
  float* sF = reinterpret_cast<float*>(s3_smem) + (size_t)wave * S3_PTS * (N + 4 * NFRE);
  float* sFac = sF + S3_PTS * N;   // [point][M][4]: FACSAT, SIG, WAVNUM, -
  for (int q = 0; q < S3_PTS; q++) {
    if (ij0 + q >= n) break;
    const float* g = fl1 + (size_t)(ij0 + q) * N;
    float* t = sF + q * N;
    for (int e = lane; e < N; e += 64) {
      const int k = e / NFRE, m = e - k * NFRE;
      t[m * NANG + 2 * (k % S3_GROUP) + k / S3_GROUP] = g[e];
    }
  }
  const float* q = pt + (size_t)ij * 12;
  const float UFRIC = q[0], RAORW = q[2], sinwd = q[7], coswd = q[8];
  if (act) {
    const float* wp = wvprpt + (size_t)ij * ECWAM_HIP_NWPR * NFRE;
    const float TPIINV = 1.0f / tb.ZPI;
    for (int m = j; m < NFRE; m += S3_GROUP) {
      float* f = sFac + ((size_t)p * NFRE + m) * 4;
      f[0] = wp[m] * TPIINV * wp[3 * NFRE + m];
      f[1] = tb.ZPIFR[m];
      f[2] = wp[m];
      f[3] = 0.f;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const float* tP = sF + (grp ? p : 0) * N;
  const float* tF = tP + 2 * j;
  const float* tFac = sFac + (size_t)(grp ? p : 0) * NFRE * 4;
  const size_t orow = (size_t)ij * NFRE * NANG + 2 * j;
  if (MODE == 1) {
    if (act)
      for (int m = 0; m < NFRE; m++) *reinterpret_cast<F2*>(fld + orow + (size_t)m * NANG) = *reinterpret_cast<const F2*>(tF + m * NANG);
    return;
  }
  const int ntap = tb.NTAP;   // <= S3_MAXTAP (checked by the launcher)
  int o0[S3_MAXTAP], o1[S3_MAXTAP];
  F2 wgt[S3_MAXTAP];
#pragma unroll
  for (int t = 0; t < S3_MAXTAP; t++) {
    const int tt = t < ntap ? t : 0;
    const int k0 = tb.INDICESSAT[tt][j], k1 = tb.INDICESSAT[tt][j + S3_GROUP];
    o0[t] = 2 * (k0 % S3_GROUP) + k0 / S3_GROUP;
    o1[t] = 2 * (k1 % S3_GROUP) + k1 / S3_GROUP;
    wgt[t] = t < ntap ? F2{tb.SATWEIGHTS[tt][j], tb.SATWEIGHTS[tt][j + S3_GROUP]} : F2{0.f, 0.f};
  }
  const int base = (grp ? p : 0) * S3_GROUP;
#define S3_ROT(r) (4 * (base + ((j + (r)) >= S3_GROUP ? j + (r) - S3_GROUP : j + (r))))
  const int a9 = S3_ROT(9), a3 = S3_ROT(3), a6 = S3_ROT(6), a1 = S3_ROT(1), a2 = S3_ROT(2);
#undef S3_ROT
  const float TMP03 = 1.0f / (tb.SDSBR * tb.MICHE), SSDSC4 = tb.SSDSC4;
  const float c2 = tb.SSDSC2 * tb.SSDSC6, c2m1 = tb.SSDSC2 * (1.0f - tb.SSDSC6);
  const bool turb = tb.SSDSC5 != 0.f;
  const float FACTURB = turb ? (2.0f * tb.SSDSC5 / tb.G) * RAORW * UFRIC * UFRIC : 0.f;
  const F2 coswdif = F2{tb.COSTH[j], tb.COSTH[j + S3_GROUP]} * coswd + F2{tb.SINTH[j], tb.SINTH[j + S3_GROUP]} * sinwd;
  for (int m = 0; m < NFRE; m++) {
    const float* row = tP + m * NANG;
    const float4 fa = *reinterpret_cast<const float4*>(tFac + m * 4);
    F2 b = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < S3_MAXTAP; t++) b = b + wgt[t] * F2{row[o0[t]], row[o1[t]]};
    b = b * fa.x;
    float bm = grp ? fmaxf(b.x, b.y) : 0.f;
    bm = fmaxf(bm, s3_bperm(a9, bm));
    bm = fmaxf(bm, fmaxf(s3_bperm(a3, bm), s3_bperm(a6, bm)));
    bm = fmaxf(bm, fmaxf(s3_bperm(a1, bm), s3_bperm(a2, bm)));
    const float a0 = fmaxf(0.f, bm * TMP03 - SSDSC4);
    const F2 t1 = b * TMP03 - SSDSC4;
    const F2 a1v = {fmaxf(0.f, t1.x), fmaxf(0.f, t1.y)};
    F2 D = (c2 * fa.y) * (a0 * a0) + (c2m1 * fa.y) * (a1v * a1v);
    if (turb) D = D - (fa.y * fa.z * FACTURB) * coswdif;
    if (act) *reinterpret_cast<F2*>(fld + orow + (size_t)m * NANG) = D;
  }
}

int launch_sdissip3(const void* tab, int nang, int nfre, int ntap, int n, const void* fl1, const void* wvprpt, const void* pt, void* fld,
                    int mode, hipStream_t s) {
  if (nang != 2 * S3_GROUP || ntap > S3_MAXTAP) return 1;
  if (n <= 0) return 0;
  constexpr int WPB = 2;
  const size_t shmem = (size_t)WPB * S3_PTS * (nang * nfre + 4 * nfre) * sizeof(float);
  const int blocks = (n + WPB * S3_PTS - 1) / (WPB * S3_PTS);
  if (mode == 1)
    hipLaunchKernelGGL((k_sdissip3<WPB, 1>), dim3(blocks), dim3(64 * WPB), shmem, s, (const DevTab<float>*)tab, n, (const float*)fl1,
                       (const float*)wvprpt, (const float*)pt, (float*)fld);
  else
    hipLaunchKernelGGL((k_sdissip3<WPB, 0>), dim3(blocks), dim3(64 * WPB), shmem, s, (const DevTab<float>*)tab, n, (const float*)fl1,
                       (const float*)wvprpt, (const float*)pt, (float*)fld);
  return 0;
}
