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
