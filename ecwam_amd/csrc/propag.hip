// Advection kernels: CTU weight construction (ctuwini.F90 + ctuw.F90, IREFRA=0) and the PROPAGS2 stencil
// (propags2.F90:99-121) on the point-major device layout FL[ij][K][M] (see include/ecwam_hip.h).
//
// PROPAGS2 is HBM-bound (16 flop per (ij,K,M) against 10 words): one thread per (ij,K,M) element,
// consecutive threads on consecutive M then K, so the eight weight streams W[ij][w][K*NR+M] and the
// own spectrum are read/written fully coalesced; neighbour spectra are gathered as contiguous
// NFRE-long runs (one run per direction), which the XCD L2 / Infinity Cache serve (a point's spectrum
// is re-used by its ~8 neighbours within one or two latitude rows).
#include "dev.h"

#include "ctu.h"

template <typename T>
__global__ void __launch_bounds__(256) k_propags2(const DevTab<T>* __restrict__ tab, const T* __restrict__ f1, T* __restrict__ f3,
                                                  const int* __restrict__ klon, const int* __restrict__ klat,
                                                  const int* __restrict__ kcor, const T* __restrict__ w, int kijs, int kijl,
                                                  int m0, int m1, int copy_rest) {
  const int NANG = tab->NANG, NFRE = tab->NFRE, NR = tab->NFRE_RED;
  const int N = NANG * NFRE;
  const long long total = (long long)(kijl - kijs) * N;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = kijs + (int)(g / N);
    const int e = (int)(g - (long long)(ij - kijs) * N);
    const int k = e / NFRE, m = e - k * NFRE;
    const size_t own = (size_t)ij * N;
    if (m < m0 || m >= m1) {
      if (copy_rest) f3[own + e] = f1[own + e];
      continue;
    }
    const int jx = tab->JXO[k][0], jy = tab->JYO[k][0], kc = tab->KCR[k][0];
    const int km = tab->KPM[k][0], kp = tab->KPM[k][2];
    const int ilon = klon[ij * 2 + jx];
    const int ilat1 = klat[(ij * 2 + jy) * 2 + 0], ilat2 = klat[(ij * 2 + jy) * 2 + 1];
    const int icor1 = kcor[(ij * 4 + kc) * 2 + 0], icor2 = kcor[(ij * 4 + kc) * 2 + 1];
    const size_t wb = (size_t)ij * 8 * NANG * NR + (size_t)k * NR + m;
    const size_t ws = (size_t)NANG * NR;
    const int km_e = k * NFRE + m;
    const T r = ctu_stencil(w[wb], w[wb + ws], w[wb + 2 * ws], w[wb + 3 * ws], w[wb + 4 * ws], w[wb + 5 * ws], w[wb + 6 * ws],
                            w[wb + 7 * ws], f1[own + km_e], f1[(size_t)ilon * N + km_e], f1[(size_t)ilat1 * N + km_e],
                            f1[(size_t)ilat2 * N + km_e], f1[(size_t)icor1 * N + km_e], f1[(size_t)icor2 * N + km_e],
                            f1[own + km * NFRE + m], f1[own + kp * NFRE + m]);
    f3[own + e] = r;
  }
}

// Vectorised form (16 B per lane: 4 floats / 2 doubles along M) used when NFRE, NFRE_RED and the frequency range are
// multiples of the vector width: 4x fewer load/store instructions, 1 KiB per wave-instruction on every stream.
template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float4 type; enum { W = 4 }; };
template <> struct VecOf<double> { typedef double2 type; enum { W = 2 }; };
__device__ __forceinline__ float4 vmul_add(float4 acc, float4 w, float4 f) {
  acc.x = acc.x + w.x * f.x; acc.y = acc.y + w.y * f.y; acc.z = acc.z + w.z * f.z; acc.w = acc.w + w.w * f.w;
  return acc;
}
__device__ __forceinline__ double2 vmul_add(double2 acc, double2 w, double2 f) {
  acc.x = acc.x + w.x * f.x; acc.y = acc.y + w.y * f.y;
  return acc;
}
__device__ __forceinline__ float4 vfirst(float4 w, float4 f) { return make_float4((1.f - w.x) * f.x, (1.f - w.y) * f.y, (1.f - w.z) * f.z, (1.f - w.w) * f.w); }
__device__ __forceinline__ double2 vfirst(double2 w, double2 f) { return make_double2((1.0 - w.x) * f.x, (1.0 - w.y) * f.y); }

template <typename T>
__global__ void __launch_bounds__(256) k_propags2_vec(const DevTab<T>* __restrict__ tab, const T* __restrict__ f1, T* __restrict__ f3,
                                                      const int* __restrict__ klon, const int* __restrict__ klat,
                                                      const int* __restrict__ kcor, const T* __restrict__ w, int kijs, int kijl,
                                                      int m0, int m1, int copy_rest) {
  typedef typename VecOf<T>::type V;
  constexpr int W = VecOf<T>::W;
  const int NANG = tab->NANG, NFRE = tab->NFRE, NR = tab->NFRE_RED;
  const int N = NANG * NFRE, NV = N / W, FV = NFRE / W;
  const long long total = (long long)(kijl - kijs) * NV;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = kijs + (int)(g / NV);
    const int ev = (int)(g - (long long)(ij - kijs) * NV);
    const int k = ev / FV, m = (ev - k * FV) * W;
    const size_t own = (size_t)ij * N;
    const int e = k * NFRE + m;
    if (m < m0 || m >= m1) {
      if (copy_rest) *reinterpret_cast<V*>(f3 + own + e) = *reinterpret_cast<const V*>(f1 + own + e);
      continue;
    }
    const int jx = tab->JXO[k][0], jy = tab->JYO[k][0], kc = tab->KCR[k][0];
    const int km = tab->KPM[k][0], kp = tab->KPM[k][2];
    const int ilon = klon[ij * 2 + jx];
    const int ilat1 = klat[(ij * 2 + jy) * 2 + 0], ilat2 = klat[(ij * 2 + jy) * 2 + 1];
    const int icor1 = kcor[(ij * 4 + kc) * 2 + 0], icor2 = kcor[(ij * 4 + kc) * 2 + 1];
    const size_t ws = (size_t)NANG * NR;
    const T* wp = w + (size_t)ij * 8 * ws + (size_t)k * NR + m;
#define LDV(p) (*reinterpret_cast<const V*>(p))
    const V w0 = LDV(wp), w1 = LDV(wp + ws), w2 = LDV(wp + 2 * ws), w3 = LDV(wp + 3 * ws), w4 = LDV(wp + 4 * ws),
            w5 = LDV(wp + 5 * ws), w6 = LDV(wp + 6 * ws), w7 = LDV(wp + 7 * ws);
    const V g0 = LDV(f1 + own + e), g1 = LDV(f1 + (size_t)ilon * N + e), g2 = LDV(f1 + (size_t)ilat1 * N + e),
            g3 = LDV(f1 + (size_t)ilat2 * N + e), g4 = LDV(f1 + (size_t)icor1 * N + e), g5 = LDV(f1 + (size_t)icor2 * N + e),
            g6 = LDV(f1 + own + km * NFRE + m), g7 = LDV(f1 + own + kp * NFRE + m);
    V r;
    const T* pw[8] = {(const T*)&w0, (const T*)&w1, (const T*)&w2, (const T*)&w3, (const T*)&w4, (const T*)&w5, (const T*)&w6, (const T*)&w7};
    const T* pf[8] = {(const T*)&g0, (const T*)&g1, (const T*)&g2, (const T*)&g3, (const T*)&g4, (const T*)&g5, (const T*)&g6, (const T*)&g7};
#pragma unroll
    for (int c = 0; c < W; c++)
      ((T*)&r)[c] = ctu_stencil(pw[0][c], pw[1][c], pw[2][c], pw[3][c], pw[4][c], pw[5][c], pw[6][c], pw[7][c], pf[0][c], pf[1][c],
                                pf[2][c], pf[3][c], pf[4][c], pf[5][c], pf[6][c], pf[7][c]);
#undef LDV
    *reinterpret_cast<V*>(f3 + own + e) = r;
  }
}

// ctuwini.F90:58-99: snap WLAT/WCOR near land.  One thread per point.
template <typename T>
__global__ void k_ctuwini(int n, int nland, const int* __restrict__ klat, const int* __restrict__ kcor, T* __restrict__ wlat,
                          T* __restrict__ wcor) {
  int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= n) return;
  for (int ic = 0; ic < 2; ic++) {
    int k1 = klat[(ij * 2 + ic) * 2 + 0], k2 = klat[(ij * 2 + ic) * 2 + 1];
    T wv = wlat[ij * 2 + ic];
    if (k1 < nland && k2 < nland) {
    } else if (k1 == nland) {
      if (wv <= T(0.75)) wv = T(0);
    } else {
      if (wv >= T(0.5)) wv = T(1);
    }
    wlat[ij * 2 + ic] = wv;
  }
  for (int icr = 0; icr < 4; icr++) {
    int k1 = kcor[(ij * 4 + icr) * 2 + 0], k2 = kcor[(ij * 4 + icr) * 2 + 1];
    T wv = wcor[ij * 4 + icr];
    if (k1 < nland && k2 < nland) {
    } else if (k1 == nland) {
      if (wv <= T(0.75)) wv = T(0);
    } else {
      if (wv > T(0.5)) wv = T(1);
    }
    wcor[ij * 4 + icr] = wv;
  }
}


// One thread per (ij,K,M): stores the eight weights (w != nullptr) and raises the CFL flag of the point.
template <typename T>
__global__ void __launch_bounds__(256) k_ctuw(const DevTab<T>* __restrict__ tab, int n, int ngy, T delpro, int m0, int m1,
                                              const int* __restrict__ kxlt, const T* __restrict__ zdello, T xdella,
                                              const T* __restrict__ cosph, const T* __restrict__ sinph,
                                              const int* __restrict__ klon, const int* __restrict__ klat,
                                              const T* __restrict__ wlat, const T* __restrict__ wcor,
                                              const T* __restrict__ cg, const T* __restrict__ cosphm1, T* __restrict__ w,
                                              int* __restrict__ cflfail, const T* __restrict__ obs) {
  const int NANG = tab->NANG, NFRE = tab->NFRE, NR = tab->NFRE_RED;
  const int nm = m1 - m0;
  const long long total = (long long)n * NANG * nm;
  const T CMTODEG = T(360.0) / tab->CIRC;
  const T DELTH0 = T(0.25) * delpro / tab->DELTH;  // ctuw.F90:407
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = (int)(g / (NANG * nm));
    const int r = (int)(g - (long long)ij * NANG * nm);
    const int k = r / nm, m = m0 + (r - k * nm);
    const CtuPoint<T> p = ctu_point(ij, ngy, kxlt, zdello, xdella, cosph, sinph, wlat, wcor, cosphm1);
    T cgl[2], cgy0[2], cgy1[2];
    for (int ic = 0; ic < 2; ic++) {
      cgl[ic] = cg[(size_t)klon[ij * 2 + ic] * NFRE + m];
      cgy0[ic] = cg[(size_t)klat[(ij * 2 + ic) * 2 + 0] * NFRE + m];
      cgy1[ic] = cg[(size_t)klat[(ij * 2 + ic) * 2 + 1] * NFRE + m];
    }
    const CtuBase<T> b = ctu_base(cg[(size_t)ij * NFRE + m], cgl, cgy0, cgy1, p.wl, p.dp);
    T tsp, tsm, w8[8];
    ctu_dirfac(tab, k, DELTH0, p.tanph, tsp, tsm);
    const int jx0 = tab->JXO[k][0], jx1 = tab->JXO[k][1], jy0 = tab->JYO[k][0], jy1 = tab->JYO[k][1], kc = tab->KCR[k][0];
    const bool fail = ctu_w8(b, tab->SINTH[k], tab->COSTH[k], p.cpm1, p.zd, xdella, p.ga, delpro, CMTODEG, jx0, jx1, jy0, jy1,
                             p.wl[jy0], p.wc[kc], tsp, tsm, w8);
    if (fail) cflfail[ij] = 1;
    if (obs) ctu_obstruct8(w8, obs + (size_t)ij * 8 * NFRE + m, NFRE, jx0, jy0, kc);
    if (w) {
      const size_t ws = (size_t)NANG * NR;
      const size_t wb = (size_t)ij * 8 * ws + (size_t)k * NR + m;
      for (int i = 0; i < 8; i++) w[wb + i * ws] = w8[i];
    }
  }
}

// PROPAGS2 with the weights rebuilt on the fly: the eight weight streams (8/10 of the HBM traffic of k_propags2*) are
// replaced by ~45 flops per element on data that is already on chip.  A block walks tiles of OTF_TP points: (1) the
// point scalars, (2) the direction-independent halves of the weights per (point, frequency) -- seven CGROUP runs per
// point, shared by all NANG directions -- go to LDS, (3) every thread combines them with its direction's factors and
// applies the stencil to VW consecutive frequencies (16-byte loads/stores along M).
#ifndef OTF_TP
#define OTF_TP 16
#endif
template <typename T, int VW> struct VecIO {  // VW consecutive elements as one 16-byte access (VW == 1: scalar)
  static __device__ __forceinline__ void ld(const T* p, T* o) {
    typedef T V __attribute__((ext_vector_type(VW)));
    const V v = *reinterpret_cast<const V*>(p);
#pragma unroll
    for (int c = 0; c < VW; c++) o[c] = v[c];
  }
  static __device__ __forceinline__ void st(T* p, const T* o) {
    typedef T V __attribute__((ext_vector_type(VW)));
    V v;
#pragma unroll
    for (int c = 0; c < VW; c++) v[c] = o[c];
    *reinterpret_cast<V*>(p) = v;
  }
  // streaming store: the advected spectra are next read by IMPLSCH, a whole pass (2 GB at O320) later -- keeping them out of the
  // way of the neighbour rows the stencil gathers through L2 is worth 6 % of the kernel (1.79 -> 1.68 ms at O320)
  static __device__ __forceinline__ void st_stream(T* p, const T* o) {
    typedef T V __attribute__((ext_vector_type(VW)));
    V v;
#pragma unroll
    for (int c = 0; c < VW; c++) v[c] = o[c];
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
  }
};
template <typename T> struct VecIO<T, 1> {
  static __device__ __forceinline__ void ld(const T* p, T* o) { o[0] = p[0]; }
  static __device__ __forceinline__ void st(T* p, const T* o) { p[0] = o[0]; }
  static __device__ __forceinline__ void st_stream(T* p, const T* o) { __builtin_nontemporal_store(o[0], p); }
};
// (single precision without obstructions -- the benchmark's build -- needs its four waves per SIMD to hide the gathers' latency: it
// compiles to 128 VGPRs as it is; pinned to four waves with amdgpu_waves_per_eu the 2-D tile order lost a quarter of its speed)
template <typename T, int VW, bool OBS>
__global__ void __launch_bounds__(256) k_propags2_otf(const DevTab<T>* __restrict__ tab, const T* __restrict__ f1_rows, T* __restrict__ f3,
                                                      int n_geom, int ngy, T delpro, const int* __restrict__ kxlt,
                                                      const T* __restrict__ zdello, T xdella, const T* __restrict__ cosph,
                                                      const T* __restrict__ sinph, const int* __restrict__ klon,
                                                      const int* __restrict__ klat, const int* __restrict__ kcor,
                                                      const T* __restrict__ wlat, const T* __restrict__ wcor,
                                                      const T* __restrict__ cg, const T* __restrict__ cosphm1,
                                                      const int* __restrict__ order, int kijs, int kijl, int m0, int m1,
                                                      int copy_rest, int ntiles, const T* __restrict__ obs, int mlf, T delpro_lf, int in_k,
                                                      T* __restrict__ gout, int gout_k, const T* __restrict__ gin, int gin_k, int out_k) {
  // gin (optional, with full input rows only): the first gin_k frequencies of every direction are READ from the compact buffer
  // gin[ij][K][gin_k] instead of the input rows -- the fast waves after their sub-steps on compact rows; the other frequencies come from f1.
  // out_k: frequencies per direction in the OUTPUT rows (NFRE, or the width of a compact buffer: a fast-wave sub-step compact -> compact).
  // gout (optional): the first gout_k frequencies of every advected direction are ALSO written to the compact buffer
  // gout[ij][K][gout_k] (the fast waves the next sub-step starts from: saves extracting them from the FL3 rows afterwards)
  // in_k: frequencies per direction in the INPUT rows (NFRE, or the width of a compact fast-wave buffer [ij][K][in_k]);
  // the output rows always have the FL layout.
  // mlf > 0: frequencies [0, mlf) (the fast waves, IFRELFMAX) advance with delpro_lf, the others with delpro, in one pass
  // (propag_wam.F90:247-283 calls PROPAGS2 once per range)
  extern __shared__ __align__(16) unsigned char otf_smem[];
  const int NANG = tab->NANG, NFRE = tab->NFRE;
  const int N = NANG * NFRE, FV = in_k / VW, NV = NANG * FV;
  if (out_k <= 0) out_k = NFRE;
  const int NOUT = NANG * out_k;
  const T CMTODEG = T(360.0) / tab->CIRC;
  const T DELTH0 = T(0.25) * delpro / tab->DELTH;
  const T DELTH0_LF = T(0.25) * delpro_lf / tab->DELTH;
  // LDS: point scalars, neighbour indices, CtuBase per (point, frequency) as 5 planes, direction factors
  CtuPoint<T>* sP = reinterpret_cast<CtuPoint<T>*>(otf_smem);
  int* sI = reinterpret_cast<int*>(sP + OTF_TP);                    // [TP][16]: ij, ilon[2], ilat[2][2], icor[4][2]
  T* sB = reinterpret_cast<T*>(sI + OTF_TP * 16);                   // [TP][5][NFRE]
  T* sK = sB + (size_t)OTF_TP * 5 * NFRE;                           // [NANG][4]: (SINTH+SINTH(K+1))*DELTH0/R, same for K-1; both for DELTH0_LF
  T* sT = sK + 4 * NANG;                                            // [NANG][2]: SINTH, COSTH
  int* sD = reinterpret_cast<int*>(sT + 2 * NANG);                  // [NANG][8]: JXO(K,1:2), JYO(K,1:2), KCR(K,1), KPM(K,-1), KPM(K,1)
  T* sQ = reinterpret_cast<T*>(sD + 8 * NANG);                      // [TP][8]: ZDELLO GA, XDELLA GA, 1 - WLAT(1:2), 1 - WCOR(1:4) (the hoisted form of the weights, ctu.h)
  T* sO = sQ + 8 * OTF_TP;                                          // OBS: [TP][8][NFRE] transmission coefficients
  // the per-direction tables go to LDS once per block: read from the DevTab in global memory inside the stencil loop they
  // were seven more loads in front of every group of gathers, through the same L1 miss queue the gathers wait in
  for (int k = threadIdx.x; k < NANG; k += blockDim.x) {
    T a, b;
    ctu_dirfac(tab, k, DELTH0, T(1), a, b);  // TANPH applied per point below: TANPH*SP is formed as in k_ctuw
#if ECWAM_HIP_CTU_STRICT
    sK[4 * k] = a; sK[4 * k + 1] = b;
    ctu_dirfac(tab, k, DELTH0_LF, T(1), a, b);
    sK[4 * k + 2] = a; sK[4 * k + 3] = b;
    sT[2 * k] = tab->SINTH[k]; sT[2 * k + 1] = tab->COSTH[k];
#else      // doubled factors of the great-circle term, magnitudes of SINTH / COSTH (ctu.h: ctu_fast_dir, ctu_fast_w8)
    sK[4 * k] = T(2) * a; sK[4 * k + 1] = T(2) * b;
    ctu_dirfac(tab, k, DELTH0_LF, T(1), a, b);
    sK[4 * k + 2] = T(2) * a; sK[4 * k + 3] = T(2) * b;
    sT[2 * k] = m_abs(tab->SINTH[k]); sT[2 * k + 1] = m_abs(tab->COSTH[k]);
#endif
    int* d = sD + 8 * k;
    d[0] = tab->JXO[k][0]; d[1] = tab->JXO[k][1]; d[2] = tab->JYO[k][0]; d[3] = tab->JYO[k][1]; d[4] = tab->KCR[k][0];
    d[5] = tab->KPM[k][0]; d[6] = tab->KPM[k][2]; d[7] = 0;
  }
  // XCD-aware tile walk: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own 4 MiB L2.  XCD x
  // walks the contiguous tile range [x*tpx, (x+1)*tpx) so that a spectrum fetched as somebody's neighbour is still in that
  // L2 when its own tile (or the next neighbour) comes up; gridDim.x is a multiple of 8 (launch_propags2_otf).
  // natural work order: element e = threadIdx.x + i blockDim.x of a tile is (point t, direction k, vector mv) with e = (t NANG + k) FV + mv
  const int step_q = (int)blockDim.x / FV, step_r = (int)blockDim.x - step_q * FV;   // blockDim.x vectors = step_q directions + step_r vectors
  const int step_wrap = (step_q + 1 + NANG - 1) / NANG;                              // directions wrap into the next point at most this often per step
  const int t_first = (int)threadIdx.x / NV, k_first = ((int)threadIdx.x - t_first * NV) / FV,
            mv_first = (int)threadIdx.x - t_first * NV - k_first * FV;
  const bool xwalk = !(copy_rest & 2);  // copy_rest & 2: plain grid-stride walk (diagnostics)
  const int tpx = xwalk ? (ntiles + 7) / 8 : ntiles;
  const int xcd = xwalk ? (blockIdx.x & 7) : 0;
  const int tend = min((xcd + 1) * tpx, ntiles);
  const int tstep = xwalk ? (gridDim.x >> 3) : gridDim.x;
  for (int tile = xcd * tpx + (xwalk ? (blockIdx.x >> 3) : blockIdx.x); tile < tend; tile += tstep) {
    const int p0 = kijs + tile * OTF_TP;
    const int np = min(OTF_TP, kijl - p0);
    __syncthreads();  // previous tile fully consumed
    if (threadIdx.x < np) {
      const int t = threadIdx.x;
      const int ij = order ? order[p0 + t] : p0 + t;
      int* q = sI + t * 16;
      q[0] = ij;   // < 0: padding entry of a 2-D tile (skipped)
      if (ij >= 0) {
        const CtuPoint<T> cp = ctu_point(ij, ngy, kxlt, zdello, xdella, cosph, sinph, wlat, wcor, cosphm1);
        sP[t] = cp;
        T* e = sQ + 8 * t;
        e[0] = cp.zd * cp.ga; e[1] = xdella * cp.ga; e[2] = T(1) - cp.wl[0]; e[3] = T(1) - cp.wl[1];
        for (int i = 0; i < 4; i++) e[4 + i] = T(1) - cp.wc[i];
        q[1] = klon[ij * 2 + 0]; q[2] = klon[ij * 2 + 1];
        for (int i = 0; i < 4; i++) q[3 + i] = klat[ij * 4 + i];
        for (int i = 0; i < 8; i++) q[7 + i] = kcor[ij * 8 + i];
      }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < np * NFRE; it += blockDim.x) {
      const int t = it / NFRE, m = it - t * NFRE;
      const int* q = sI + t * 16;
      if (q[0] >= 0 && m >= (m0 / VW) * VW && m < ((m1 + VW - 1) / VW) * VW) {   // whole vectors around the range (their spare lanes are computed, not stored)
        T cgl[2], cgy0[2], cgy1[2];
        for (int ic = 0; ic < 2; ic++) {
          cgl[ic] = cg[(size_t)q[1 + ic] * NFRE + m];
          cgy0[ic] = cg[(size_t)q[3 + 2 * ic] * NFRE + m];
          cgy1[ic] = cg[(size_t)q[4 + 2 * ic] * NFRE + m];
        }
        const CtuBase<T> b = ctu_base(cg[(size_t)q[0] * NFRE + m], cgl, cgy0, cgy1, sP[t].wl, sP[t].dp);
        T* o = sB + (size_t)t * 5 * NFRE + m;
#if ECWAM_HIP_CTU_STRICT
        o[0] = b.h[0]; o[NFRE] = b.h[1]; o[2 * NFRE] = b.hy[0]; o[3 * NFRE] = b.hy[1]; o[4 * NFRE] = b.cg0;
#else      // |h| COSPHM1 DELPRO CMTODEG and |hy| DELPRO CMTODEG with the frequency's own time step
        ctu_fast_planes<T>(b, m_abs(sP[t].cpm1), (m < mlf ? delpro_lf : delpro) * CMTODEG, o, o + NFRE, o + 2 * NFRE, o + 3 * NFRE);
        o[4 * NFRE] = b.cg0;
#endif
        if (OBS) {
#pragma unroll
          for (int i = 0; i < 8; i++) sO[((size_t)t * 8 + i) * NFRE + m] = obs[((size_t)q[0] * 8 + i) * NFRE + m];
        }
      }
    }
    __syncthreads();
    typedef VecIO<T, VW> IO;
    auto element = [&](const int t, const int k, const int mv) {   // direction k, vector mv of the NFRE / VW of a direction
      const int m = mv * VW;
      const int* q = sI + t * 16;
      const bool fromg = gin && m < gin_k;                 // this vector's inputs (own and neighbours) live in the compact buffer
      const T* __restrict__ f1 = fromg ? gin : f1_rows;
      const int ik = fromg ? gin_k : in_k;
      const int NIN = NANG * ik;
      const size_t own = (size_t)q[0] * NOUT, own_in = (size_t)q[0] * NIN;
      const int el = k * out_k + m, el_in = k * ik + m;
      if (m + VW <= m0 || m >= m1) {   // no element of this vector is advected
        if (copy_rest & 1) {
          T v[VW];
          IO::ld(f1 + own_in + el_in, v);
          IO::st(f3 + own + el, v);
          // (a compact copy wider than the advected range, LFP > NFRE_RED: its last vector is carried over as well)
          if (gout && m < gout_k) IO::st(gout + ((size_t)q[0] * NANG + k) * gout_k + m, v);
        }
        return;
      }
      const bool partial = (VW > 1) && (m < m0 || m + VW > m1);   // the range boundary cuts this vector
      const CtuPoint<T>& p = sP[t];
      const int* dk = sD + 8 * k;
      const int jx0 = dk[0], jx1 = dk[1], jy0 = dk[2], jy1 = dk[3], kc = dk[4];
      const int km = dk[5], kp = dk[6];
      const T sink = sT[2 * k], cosk = sT[2 * k + 1];
      T tsp, tsm, tsp_lf, tsm_lf;
      {
#pragma clang fp contract(off)
        tsp = p.tanph * sK[4 * k];
        tsm = p.tanph * sK[4 * k + 1];
        tsp_lf = p.tanph * sK[4 * k + 2];
        tsm_lf = p.tanph * sK[4 * k + 3];
      }
      T fo[VW], flon[VW], fla1[VW], fla2[VW], fco1[VW], fco2[VW], fkm[VW], fkp[VW];
      IO::ld(f1 + own_in + el_in, fo);
      IO::ld(f1 + (size_t)q[1 + jx0] * NIN + el_in, flon);
      IO::ld(f1 + (size_t)q[3 + 2 * jy0] * NIN + el_in, fla1);
      IO::ld(f1 + (size_t)q[4 + 2 * jy0] * NIN + el_in, fla2);
      IO::ld(f1 + (size_t)q[7 + 2 * kc] * NIN + el_in, fco1);
      IO::ld(f1 + (size_t)q[8 + 2 * kc] * NIN + el_in, fco2);
      IO::ld(f1 + own_in + km * ik + m, fkm);
      IO::ld(f1 + own_in + kp * ik + m, fkp);
      const T* bb = sB + (size_t)t * 5 * NFRE + m;
      T r[VW];
#if ECWAM_HIP_CTU_STRICT
      T bh0[VW], bh1[VW], by0[VW], by1[VW], bc0[VW];
      IO::ld(bb, bh0); IO::ld(bb + NFRE, bh1); IO::ld(bb + 2 * NFRE, by0); IO::ld(bb + 3 * NFRE, by1); IO::ld(bb + 4 * NFRE, bc0);
      if constexpr (sizeof(T) == 4 && (VW % 2 == 0) && !OBS) {
        // two frequencies per packed-fp32 operand
#pragma unroll
        for (int c = 0; c < VW; c += 2) {
          const bool lf0 = (m + c) < mlf, lf1 = (m + c + 1) < mlf;
          const F2 dl = {lf0 ? (float)delpro_lf : (float)delpro, lf1 ? (float)delpro_lf : (float)delpro};
          const F2 sp2 = {lf0 ? (float)tsp_lf : (float)tsp, lf1 ? (float)tsp_lf : (float)tsp};
          const F2 sm2 = {lf0 ? (float)tsm_lf : (float)tsm, lf1 ? (float)tsm_lf : (float)tsm};
#define P2(a) F2{(float)a[c], (float)a[c + 1]}
          const F2 rr = ctu_w8_stencil_pk<float>(P2(bh0), P2(bh1), P2(by0), P2(by1), P2(bc0), (float)sink, (float)cosk, (float)p.cpm1, (float)p.zd,
                                          (float)xdella, (float)p.ga, dl, (float)CMTODEG, jx0, jy0, (float)p.wl[jy0], (float)p.wc[kc], sp2,
                                          sm2, P2(fo), P2(flon), P2(fla1), P2(fla2), P2(fco1), P2(fco2), P2(fkm), P2(fkp));
#undef P2
          r[c] = (T)rr.x; r[c + 1] = (T)rr.y;
        }
      } else
#pragma unroll
      for (int c = 0; c < VW; c++) {
        CtuBase<T> b;
        b.h[0] = bh0[c]; b.h[1] = bh1[c]; b.hy[0] = by0[c]; b.hy[1] = by1[c]; b.cg0 = bc0[c];
        const bool lf = (m + c) < mlf;
        T w8[8];
        (void)ctu_w8(b, sink, cosk, p.cpm1, p.zd, xdella, p.ga, lf ? delpro_lf : delpro, CMTODEG, jx0, jx1, jy0, jy1, p.wl[jy0], p.wc[kc],
                     lf ? tsp_lf : tsp, lf ? tsm_lf : tsm, w8);
        if (OBS) ctu_obstruct8(w8, sO + (size_t)t * 8 * NFRE + m + c, NFRE, jx0, jy0, kc);
        const T a = ctu_stencil(w8[0], w8[1], w8[2], w8[3], w8[4], w8[5], w8[6], w8[7], fo[c], flon[c], fla1[c], fla2[c], fco1[c],
                                fco2[c], fkm[c], fkp[c]);
        r[c] = a;
      }
#else
      {
        // the hoisted form (ctu.h: ctu_fast_w8): planes ordered by the direction's quadrant, two frequencies per packed operand
        T xa[VW], xb[VW], ya[VW], yb[VW], bc0[VW];
        IO::ld(bb + jx0 * NFRE, xa); IO::ld(bb + (1 - jx0) * NFRE, xb); IO::ld(bb + (2 + jy0) * NFRE, ya); IO::ld(bb + (3 - jy0) * NFRE, yb);
        IO::ld(bb + 4 * NFRE, bc0);
        const T* e = sQ + 8 * t;
        const T zdg = e[0], xdg = e[1], omwl = e[2 + jy0], omwc = e[4 + kc], wl = p.wl[jy0], wc = p.wc[kc];
        T ab2, p2, m2, ab2_lf, p2_lf, m2_lf;
        ctu_fast_dir<T>(p.tanph, sK[4 * k], sK[4 * k + 1], ab2, p2, m2);
        ctu_fast_dir<T>(p.tanph, sK[4 * k + 2], sK[4 * k + 3], ab2_lf, p2_lf, m2_lf);
        typedef CtuV2<T> F;
#pragma unroll
        for (int c = 0; c < VW; c += 2) {
          constexpr int c1 = VW > 1 ? 1 : 0;      // (scalar accesses: the pair holds the element twice)
          const bool lf0 = (m + c) < mlf, lf1 = (m + c + c1) < mlf;
#define P2(a) F{a[c], a[c + c1]}
          CtuFastW8<T> w = ctu_fast_w8<T>(P2(xa), P2(xb), P2(ya), P2(yb), P2(bc0), sink, cosk, p.zd, xdella, p.ga, zdg, xdg, wl, omwl, wc, omwc,
                                          F{lf0 ? ab2_lf : ab2, lf1 ? ab2_lf : ab2}, F{lf0 ? p2_lf : p2, lf1 ? p2_lf : p2},
                                          F{lf0 ? m2_lf : m2, lf1 ? m2_lf : m2});
          if (OBS) {      // ctuw.F90:703-733: the space weights of the neighbours scaled by the transmission coefficients
            const T* o = sO + (size_t)t * 8 * NFRE + m + c;
            const F olon = {o[(2 + jx0) * NFRE], o[(2 + jx0) * NFRE + c1]}, olat = {o[jy0 * NFRE], o[jy0 * NFRE + c1]},
                    ocor = {o[(4 + kc) * NFRE], o[(4 + kc) * NFRE + c1]};
            w.wlon = w.wlon * olon; w.wlat1 = w.wlat1 * olat; w.wlat2 = w.wlat2 * olat; w.wcor1 = w.wcor1 * ocor; w.wcor2 = w.wcor2 * ocor;
          }
          const F rr = ctu_fast_apply<T>(w, P2(fo), P2(flon), P2(fla1), P2(fla2), P2(fco1), P2(fco2), P2(fkm), P2(fkp));
#undef P2
          r[c] = rr.x;
          if (VW > 1) r[c + c1] = rr.y;
        }
      }
#endif
      if (partial) {  // elements outside [m0, m1): carried over from F1 (copy_rest) or left as they are in F3
        T keep[VW];
        if (copy_rest & 1) {
#pragma unroll
          for (int c = 0; c < VW; c++) keep[c] = fo[c];
        } else IO::ld(f3 + own + el, keep);
#pragma unroll
        for (int c = 0; c < VW; c++)
          if (m + c < m0 || m + c >= m1) r[c] = keep[c];
      }
      IO::st_stream(f3 + own + el, r);
      if (gout && m < gout_k) IO::st(gout + ((size_t)q[0] * NANG + k) * gout_k + m, r);
    };
    if (copy_rest & 4) {
      // 2-D tiles (decomp.tile2d_order: tile entry t = 4 g + w is the g-th point of a segment of latitude row w of a group of four
      // rows): wave w walks the points of row w, the four waves work on the same 1 KB chunk of (K, M) of four latitude neighbours at
      // the same time, chunk by chunk over the whole tile, so that a chunk fetched as a neighbour is served by the CU's L1 / the XCD's
      // L2 when its owner (or the next neighbour) asks for it a few hundred cycles later
      const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
      for (int c = 0; c < NV; c += 64) {
        const int ev = c + lane, k = ev / FV, mv = ev - k * FV;
        for (int g = 0; g < OTF_TP / 4; g++) {
          const int t = 4 * g + w;
          if (t < np && sI[t * 16] >= 0 && ev < NV) element(t, k, mv);
        }
      }
    } else {
      // (point, direction, vector) of the thread's element advance by blockDim.x vectors per iteration: kept incrementally -- two
      // divisions by run-time values per element (~ 55 of the ~ 330 instructions of an iteration) were the alternative
      int t = t_first, k = k_first, mv = mv_first;
      for (int e = threadIdx.x; e < np * NV; e += blockDim.x) {
        if (sI[t * 16] >= 0) element(t, k, mv);
        mv += step_r; k += step_q;
        if (mv >= FV) { mv -= FV; k += 1; }
        for (int i = 0; i < step_wrap; i++)
          if (k >= NANG) { k -= NANG; t += 1; }
      }
    }
  }
}

// =====================================================================================================================
// IREFRA = 1 (depth refraction), 2 (current refraction), 3 (depth + current refraction)
//
// The reference keeps THDD/THDC(IJ,K), SDOT(IJ,K,M) and 21 weight arrays per (IJ,K,M) (ctuwupdt.F90:163-177).  Here the
// per-point part of PROPDOT is stored -- REFR[ij][2*NANG+5] = THD(K) (THDD for IREFRA=1, THDC otherwise), S0(K) (the
// current-gradient factor of SDOT, propdot.F90:172-173), U, V, OMDD, CURMASK of the first / second frequency range
// (CTUWDRV is called per range, ctuwupdt.F90:220-256) -- and everything per (K,M) is rebuilt inside
// the stencil, as for IREFRA=0.
// =====================================================================================================================
#define REFR_U(NANG) (2 * (NANG))
#define REFR_V(NANG) (2 * (NANG) + 1)
#define REFR_OMDD(NANG) (2 * (NANG) + 2)
#define REFR_MASK(NANG) (2 * (NANG) + 3)
#define REFR_W(NANG) (2 * (NANG) + 5)

// gradi.F90:113-232 + propdot.F90:108-196, one thread per point
template <typename T>
__global__ void k_propdot(const DevTab<T>* __restrict__ tab, int n, int nland, int IREFRA, const int* __restrict__ kxlt,
                          const T* __restrict__ zdello, T xdella, const T* __restrict__ cosph, const int* __restrict__ klon,
                          const int* __restrict__ klat, const T* __restrict__ wlat, const T* __restrict__ cosphm1,
                          const T* __restrict__ depth, const T* __restrict__ ue, const T* __restrict__ ve, T* __restrict__ refr) {
#pragma clang fp contract(off)
  const int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= n) return;
  const int NANG = tab->NANG;
  const T CGM = T(0.00001);  // CURRENT_GRADIENT_MAX, yowcurr.F90:19
  const T DELPHI = xdella * tab->CIRC / T(360.0);
  const T ONEO2DELPHI = T(0.5) / DELPHI;
  const int KX = kxlt[ij];
  const T DELLAM = zdello[KX] * tab->CIRC / T(360.0);
  T DDPHI = T(0), DDLAM = T(0), DUPHI = T(0), DULAM = T(0), DVPHI = T(0), DVLAM = T(0);
  const T wl0 = wlat[ij * 2 + 0], wl1 = wlat[ij * 2 + 1];
  if (IREFRA == 1 || IREFRA == 3) {
    const int IPP = klat[(ij * 2 + 1) * 2 + 0], IPM = klat[(ij * 2 + 0) * 2 + 0];
    const int IPP2 = klat[(ij * 2 + 1) * 2 + 1], IPM2 = klat[(ij * 2 + 0) * 2 + 1];
    if (IPP != nland && IPM != nland && IPP2 != nland && IPM2 != nland) {
      const T DPTP = wl1 * depth[IPP] + (T(1) - wl1) * depth[IPP2];
      const T DPTM = wl0 * depth[IPM] + (T(1) - wl0) * depth[IPM2];
      DDPHI = (DPTP - DPTM) * ONEO2DELPHI;
    } else if (IPP != nland && IPM != nland) DDPHI = (depth[IPP] - depth[IPM]) * ONEO2DELPHI;
    else if (IPP2 != nland && IPM2 != nland) DDPHI = (depth[IPP2] - depth[IPM2]) * ONEO2DELPHI;
    const int ILP = klon[ij * 2 + 1], ILM = klon[ij * 2 + 0];
    if (ILP != nland && ILM != nland) DDLAM = (depth[ILP] - depth[ILM]) / (T(2) * DELLAM);
  }
  if (IREFRA == 2 || IREFRA == 3) {
    int IPP = klat[(ij * 2 + 1) * 2 + 0], IPM = klat[(ij * 2 + 0) * 2 + 0];
    int IPP2 = klat[(ij * 2 + 1) * 2 + 1], IPM2 = klat[(ij * 2 + 0) * 2 + 1];
    // an exact zero means "current not defined there": no gradient is extrapolated (gradi.F90:170-180)
    if (ue[IPP] == T(0) && ve[IPP] == T(0)) IPP = nland;
    if (ue[IPM] == T(0) && ve[IPM] == T(0)) IPM = nland;
    if (ue[IPP2] == T(0) && ve[IPP2] == T(0)) IPP2 = nland;
    if (ue[IPM2] == T(0) && ve[IPM2] == T(0)) IPM2 = nland;
    if (IPP != nland && IPM != nland && IPP2 != nland && IPM2 != nland) {
      const T UP = wl1 * ue[IPP] + (T(1) - wl1) * ue[IPP2], VP = wl1 * ve[IPP] + (T(1) - wl1) * ve[IPP2];
      const T UM = wl0 * ue[IPM] + (T(1) - wl0) * ue[IPM2], VM = wl0 * ve[IPM] + (T(1) - wl0) * ve[IPM2];
      DUPHI = (UP - UM) * ONEO2DELPHI;
      DVPHI = (VP - VM) * ONEO2DELPHI;
    } else if (IPP != nland && IPM != nland) {
      DUPHI = (ue[IPP] - ue[IPM]) * ONEO2DELPHI;
      DVPHI = (ve[IPP] - ve[IPM]) * ONEO2DELPHI;
    }
    int ILP = klon[ij * 2 + 1], ILM = klon[ij * 2 + 0];
    if (ue[ILP] == T(0) && ve[ILP] == T(0)) ILP = nland;
    if (ue[ILM] == T(0) && ve[ILM] == T(0)) ILM = nland;
    if (ILP != nland && ILM != nland) {
      DULAM = (ue[ILP] - ue[ILM]) / (T(2) * DELLAM);
      DVLAM = (ve[ILP] - ve[ILM]) / (T(2) * DELLAM);
    }
    const T CGMAX = CGM * cosph[KX];
    DUPHI = m_sign(m_min(m_abs(DUPHI), CGMAX), DUPHI);
    DVPHI = m_sign(m_min(m_abs(DVPHI), CGMAX), DVPHI);
    DULAM = m_sign(m_min(m_abs(DULAM), CGMAX), DULAM);
    DVLAM = m_sign(m_min(m_abs(DVLAM), CGMAX), DVLAM);
  }
  const T DCO = cosphm1[ij];
  T* o = refr + (size_t)ij * REFR_W(NANG);
  o[REFR_U(NANG)] = ue[ij];
  o[REFR_V(NANG)] = ve[ij];
  o[REFR_OMDD(NANG)] = (IREFRA == 3) ? ve[ij] * DDPHI + ue[ij] * DDLAM * DCO : T(0);
  o[REFR_MASK(NANG)] = T(1);
  o[REFR_MASK(NANG) + 1] = T(1);
  for (int k = 0; k < NANG; k++) {
    const T SD = tab->SINTH[k], CD = tab->COSTH[k];
    T thd = T(0), s0 = T(0);
    if (IREFRA == 1) thd = SD * DDPHI - CD * DDLAM * DCO;  // THDD enters CTUW for IREFRA = 1 only (ctuw.F90:434)
    if (IREFRA == 2 || IREFRA == 3) {
      const T SS = SD * SD, SC = SD * CD, CC = CD * CD;
      s0 = -SC * DUPHI - CC * DVPHI - (SS * DULAM + SC * DVLAM) * DCO;
      thd = SS * DUPHI + SC * DVPHI - (SC * DULAM + CC * DVLAM) * DCO;
    }
    o[k] = thd;
    o[NANG + k] = s0;
  }
}

// All weights of one (point, K, M) for any IREFRA (ctuw.F90:146-275, 403-527, 536-687); returns the CFL / range flag.
// jx0,jx1,jy0,jy1,kc[] are the 0-based JXO/JYO/KCR entries of direction K.  dthp/dthm: the complete theta-dot sums of the
// direction refraction, fdp/fdm: those of the frequency shift (formed by the caller), cur: IREFRA = 2 or 3.
template <typename T>
struct CtuGenW {
  T sumwn, wlon[2], wlat[2][2], wcor[4][2], wk[3], wm[3];
};
template <typename T>
__device__ __forceinline__ bool same_sign(T a, T b) { return __builtin_signbit(a) == __builtin_signbit(b); }
// No array is indexed with a run-time value (JXO/JYO/KCR entries select through conditionals): everything stays in registers.
template <typename T>
__device__ __forceinline__ T sel2(int i, T a0, T a1) { return i ? a1 : a0; }
template <typename T, bool CHECK>
__device__ __forceinline__ bool ctu_wgen(const CtuBase<T>& b, T sink, T cosk, const CtuPoint<T>& p, T xdella, T delpro, T cmtodeg,
                                         int jx0, int jx1, int jy0, int jy1, const int* kc, bool cur, T u, T v, T dthp, T dthm,
                                         T fdp, T fdm, T fratio, CtuGenW<T>& w, bool& flipped) {
  // flipped: the current turns the advection velocity against the group velocity somewhere (a downwind weight is non-zero)
#pragma clang fp contract(off)
  T adxp[2], adyp[2], dxup[2], dxdw[2], dyup[2], dydw[2];
  bool fail = false;
  flipped = false;
#pragma unroll
  for (int ic = 0; ic < 2; ic++) {
    const T cgx = b.h[ic] * sink * p.cpm1;
    const T cgy = b.hy[ic] * cosk;
    T urel = cgx, vrel = cgy;
    bool su = true, sv = true;
    if (cur) {
      const T uu = u * p.cpm1;
      urel = cgx + uu;
      su = same_sign(urel, cgx);
      const T vv = v * T(0.5) * (T(1) + p.dp[ic]);
      vrel = cgy + vv;
      sv = same_sign(vrel, cgy);
      if (!su || !sv) flipped = true;
    }
    adxp[ic] = m_abs(-delpro * urel * cmtodeg);
    adyp[ic] = m_abs(-delpro * vrel * cmtodeg);
    dxup[ic] = su ? adxp[ic] : T(0); dxdw[ic] = su ? T(0) : adxp[ic];
    dyup[ic] = sv ? adyp[ic] : T(0); dydw[ic] = sv ? T(0) : adyp[ic];
    if (CHECK && (adxp[ic] > p.zd || adyp[ic] > xdella)) fail = true;
  }
  // the JXO / JYO selections (jx1 = 1 - jx0, jy1 = 1 - jy0)
  const T dxup_0 = sel2(jx0, dxup[0], dxup[1]), dxup_1 = sel2(jx1, dxup[0], dxup[1]);   // DXUP(JXO(K,1)), DXUP(JXO(K,2))
  const T dxdw_0 = sel2(jx0, dxdw[0], dxdw[1]), dxdw_1 = sel2(jx1, dxdw[0], dxdw[1]);
  const T dyup_0 = sel2(jy0, dyup[0], dyup[1]), dyup_1 = sel2(jy1, dyup[0], dyup[1]);
  const T dydw_0 = sel2(jy0, dydw[0], dydw[1]), dydw_1 = sel2(jy1, dydw[0], dydw[1]);
  const T dxx = p.zd - dxup_1 - dxdw_0;
  const T dyy = xdella - dyup_1 - dydw_0;
  const T wgt_a = dxx * dyup_0 * p.ga;  // WEIGHT(JYO(K,1))
  const T wgt_b = dxx * dydw_1 * p.ga;  // WEIGHT(JYO(K,2))
  const T wgt0 = sel2(jy0, wgt_a, wgt_b), wgt1 = sel2(jy0, wgt_b, wgt_a);  // WEIGHT(1), WEIGHT(2)
  w.wlat[0][0] = p.wl[0] * wgt0;
  w.wlat[0][1] = (T(1) - p.wl[0]) * wgt0;
  w.wlat[1][0] = p.wl[1] * wgt1;
  w.wlat[1][1] = (T(1) - p.wl[1]) * wgt1;
  const T wlon_a = dyy * dxup_0 * p.ga;  // WLONN(JXO(K,1))
  const T wlon_b = dyy * dxdw_1 * p.ga;  // WLONN(JXO(K,2))
  w.wlon[0] = sel2(jx0, wlon_a, wlon_b);
  w.wlon[1] = sel2(jx0, wlon_b, wlon_a);
  T wc4[4];
  wc4[0] = dxup_0 * dyup_0 * p.ga;
  wc4[1] = dxdw_1 * dyup_0 * p.ga;
  wc4[2] = dxup_0 * dydw_1 * p.ga;
  wc4[3] = dxdw_1 * dydw_1 * p.ga;
#pragma unroll
  for (int icr = 0; icr < 4; icr++) {
    const int kk = kc[icr];
    const T wcv = kk == 0 ? p.wc[0] : (kk == 1 ? p.wc[1] : (kk == 2 ? p.wc[2] : p.wc[3]));
    w.wcor[icr][0] = wcv * wc4[icr];
    w.wcor[icr][1] = (T(1) - wcv) * wc4[icr];
  }
  T sumwn = (p.zd * (dydw_0 + dyup_1) + xdella * (dxup_1 + dxdw_0) - (dxdw_0 + dxup_1) * (dydw_0 + dyup_1)) * p.ga;
  w.wk[1] = (dthp + m_abs(dthp)) + (m_abs(dthm) - dthm);
  w.wk[2] = -dthp + m_abs(dthp);
  w.wk[0] = dthm + m_abs(dthm);
  w.wm[0] = w.wm[1] = w.wm[2] = T(0);
  if (cur) {
    w.wm[1] = (fdp + m_abs(fdp)) + (m_abs(fdm) - fdm);
    w.wm[2] = (-fdp + m_abs(fdp)) / fratio;
    w.wm[0] = (fdm + m_abs(fdm)) * fratio;
  }
  const T one = T(1), zero = T(0);
#define OUTR(x) ((x) > one || (x) < zero)
  if (CHECK) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (OUTR(w.wlon[i]) || OUTR(w.wlat[i][0]) || OUTR(w.wlat[i][1])) fail = true;
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (OUTR(w.wcor[i][0]) || OUTR(w.wcor[i][1])) fail = true;
#pragma unroll
    for (int i = 0; i < 3; i++)
      if (OUTR(w.wk[i]) || OUTR(w.wm[i])) fail = true;
  }
  sumwn = sumwn + w.wk[1];
  if (cur) sumwn = sumwn + w.wm[1];
  if (CHECK && OUTR(sumwn)) fail = true;
#undef OUTR
  w.sumwn = sumwn;
  return fail;
}

// PROPAGS2 for IREFRA = 1, 2, 3 with the weights rebuilt on the fly (and, with f1 == nullptr, the CFL / range checks of
// CTUW alone: cflfail[ij] = 1 where one fails).  Same tile structure as k_propags2_otf: a thread owns VW consecutive
// frequencies of one (point, direction) and moves 16 bytes per access.
// IREFRA = 1: the eight-term stencil of propags2.F90:107-116; IREFRA = 2, 3: every neighbour in the order of
// propags2.F90:130-186 -- a neighbour whose VW weights are all zero is not loaded (the reference's LLW* flags skip a term
// only when its weight is zero at every point: adding a zero product changes nothing).
// theta-dot (ctuw.F90:424-452, 471-493) and sigma-dot (ctuw.F90:506-520) sums are formed per thread from the REFR row.
#define GEN_TP 16
template <typename T, int VW, bool CHECK, bool OBS>
__global__ void __launch_bounds__(256) k_propags2_gen(const DevTab<T>* __restrict__ tab, int IREFRA, const T* __restrict__ f1,
                                                      T* __restrict__ f3, int ngy, T delpro, const int* __restrict__ kxlt,
                                                      const T* __restrict__ zdello, T xdella, const T* __restrict__ cosph,
                                                      const T* __restrict__ sinph, const int* __restrict__ klon,
                                                      const int* __restrict__ klat, const int* __restrict__ kcor,
                                                      const T* __restrict__ wlat, const T* __restrict__ wcor,
                                                      const T* __restrict__ cg, const T* __restrict__ om, const T* __restrict__ wn,
                                                      const T* __restrict__ cosphm1, const T* __restrict__ refr,
                                                      int* __restrict__ cflfail, int slot, int kijs, int kijl, int m0, int m1,
                                                      int copy_rest, int ntiles, const T* __restrict__ obs) {
  extern __shared__ __align__(16) unsigned char gen_smem[];
  const int NANG = tab->NANG, NFRE = tab->NFRE, NR = tab->NFRE_RED;
  const int N = NANG * NFRE, RW = REFR_W(NANG), NV = N / VW, FV = NFRE / VW;
  const T CMTODEG = T(360.0) / tab->CIRC;
  const T FRATIO = tab->FRATIO;
  T DELTH0, DELFR0;
  {
#pragma clang fp contract(off)
    DELTH0 = T(0.25) * delpro / tab->DELTH;
    DELFR0 = T(0.25) * delpro / ((tab->FRATIO - T(1)) * tab->ZPI);
  }
  const bool cur = (IREFRA == 2 || IREFRA == 3);
  CtuPoint<T>* sP = reinterpret_cast<CtuPoint<T>*>(gen_smem);
  int* sI = reinterpret_cast<int*>(sP + GEN_TP);                    // [TP][16]
  T* sB = reinterpret_cast<T*>(sI + GEN_TP * 16);                   // [TP][7][NFRE]: h0 h1 hy0 hy1 cg0 om wn
  T* sR = sB + (size_t)GEN_TP * 7 * NFRE;                           // [TP][RW]
  T* sK = sR + (size_t)GEN_TP * RW;                                 // [NANG][2]
  T* sDF = sK + 2 * NANG;                                           // [2][NFRE]: DELFR0/FR(M), DELFR0/FR(MAX(1,M-1))
  T* sT = sDF + 2 * NFRE;                                           // [NANG][2]: SINTH, COSTH
  int* sD = reinterpret_cast<int*>(sT + 2 * NANG);                  // [NANG][12]: JXO(K,1:2), JYO(K,1:2), KPM(K,-1), KPM(K,1), -, -, KCR(K,1:4)
  T* sQ = reinterpret_cast<T*>(sD + 12 * NANG);                     // [TP][12]: ZDELLO GA, XDELLA GA, 1 - WLAT(1:2), 1 - WCOR(1:4), the current's U / V terms of ADXP / ADYP(1:2)
  T* sBF = sQ + 12 * GEN_TP;                                        // [TP][4][NFRE]: the hoisted planes of the weights (ctu_fast_planes)
  T* sO = sBF + (size_t)GEN_TP * 4 * NFRE;                          // OBS: [TP][8][NFRE] transmission coefficients (LSUBGRID)
  for (int k = threadIdx.x; k < NANG; k += blockDim.x) {            // per-direction tables: LDS instead of global loads in the stencil loop
    T a, b;
    ctu_dirfac(tab, k, DELTH0, T(1), a, b);
    sK[2 * k] = a; sK[2 * k + 1] = b;
    sT[2 * k] = tab->SINTH[k]; sT[2 * k + 1] = tab->COSTH[k];
    int* d = sD + 12 * k;
    d[0] = tab->JXO[k][0]; d[1] = tab->JXO[k][1]; d[2] = tab->JYO[k][0]; d[3] = tab->JYO[k][1];
    d[4] = tab->KPM[k][0]; d[5] = tab->KPM[k][2]; d[6] = 0; d[7] = 0;
    for (int i = 0; i < 4; i++) d[8 + i] = tab->KCR[k][i];
  }
  for (int m = threadIdx.x; m < NFRE; m += blockDim.x) {
#pragma clang fp contract(off)
    sDF[m] = DELFR0 / tab->FR[m];
    sDF[NFRE + m] = DELFR0 / tab->FR[m > 0 ? m - 1 : 0];
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p0 = kijs + tile * GEN_TP;
    const int np = min(GEN_TP, kijl - p0);
    __syncthreads();
    if (threadIdx.x < np) {
      const int t = threadIdx.x, ij = p0 + t;
      const CtuPoint<T> cp = ctu_point(ij, ngy, kxlt, zdello, xdella, cosph, sinph, wlat, wcor, cosphm1);
      sP[t] = cp;
      {
#pragma clang fp contract(off)
        T* e = sQ + 12 * t;
        const T dc = delpro * CMTODEG;
        const T* rr = refr + (size_t)ij * RW;
        e[0] = cp.zd * cp.ga; e[1] = xdella * cp.ga; e[2] = T(1) - cp.wl[0]; e[3] = T(1) - cp.wl[1];
        for (int i = 0; i < 4; i++) e[4 + i] = T(1) - cp.wc[i];
        e[8] = cur ? (rr[REFR_U(NANG)] * cp.cpm1) * dc : T(0);
        e[9] = cur ? (rr[REFR_V(NANG)] * T(0.5) * (T(1) + cp.dp[0])) * dc : T(0);
        e[10] = cur ? (rr[REFR_V(NANG)] * T(0.5) * (T(1) + cp.dp[1])) * dc : T(0);
        e[11] = T(0);
      }
      int* q = sI + t * 16;
      q[0] = ij;
      q[1] = klon[ij * 2 + 0]; q[2] = klon[ij * 2 + 1];
      for (int i = 0; i < 4; i++) q[3 + i] = klat[ij * 4 + i];
      for (int i = 0; i < 8; i++) q[7 + i] = kcor[ij * 8 + i];
    }
    for (int it = threadIdx.x; it < np * RW; it += blockDim.x) sR[it] = refr[(size_t)p0 * RW + it];
    __syncthreads();
    for (int it = threadIdx.x; it < np * NFRE; it += blockDim.x) {
      const int t = it / NFRE, m = it - t * NFRE;
      const int* q = sI + t * 16;
      T cgl[2], cgy0[2], cgy1[2];
      for (int ic = 0; ic < 2; ic++) {
        cgl[ic] = cg[(size_t)q[1 + ic] * NFRE + m];
        cgy0[ic] = cg[(size_t)q[3 + 2 * ic] * NFRE + m];
        cgy1[ic] = cg[(size_t)q[4 + 2 * ic] * NFRE + m];
      }
      const CtuBase<T> b = ctu_base(cg[(size_t)q[0] * NFRE + m], cgl, cgy0, cgy1, sP[t].wl, sP[t].dp);
      T* o = sB + (size_t)t * 7 * NFRE + m;
      o[0] = b.h[0]; o[NFRE] = b.h[1]; o[2 * NFRE] = b.hy[0]; o[3 * NFRE] = b.hy[1]; o[4 * NFRE] = b.cg0;
      o[5 * NFRE] = om[(size_t)q[0] * NFRE + m];
      o[6 * NFRE] = wn[(size_t)q[0] * NFRE + m];
      {
        T* of = sBF + (size_t)t * 4 * NFRE + m;
        ctu_fast_planes<T>(b, m_abs(sP[t].cpm1), delpro * CMTODEG, of, of + NFRE, of + 2 * NFRE, of + 3 * NFRE);
      }
      if (OBS) {
#pragma unroll
        for (int i = 0; i < 8; i++) sO[((size_t)t * 8 + i) * NFRE + m] = obs[((size_t)q[0] * 8 + i) * NFRE + m];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < np * NV; e += blockDim.x) {
      const int t = e / NV, ev = e - t * NV;
      const int k = ev / FV, m = (ev - k * FV) * VW;
      const int* q = sI + t * 16;
      const size_t own = (size_t)q[0] * N;
      const int el = k * NFRE + m;
      typedef VecIO<T, VW> IO;
      if (m < m0 || m >= m1) {
        if (f1 && (copy_rest & 1)) {
          T v[VW];
          IO::ld(f1 + own + el, v);
          IO::st(f3 + own + el, v);
        }
        continue;
      }
      const CtuPoint<T>& p = sP[t];
      const int* dk = sD + 12 * k;
      const int jx0 = dk[0], jx1 = dk[1], jy0 = dk[2], jy1 = dk[3];
      const int* kc = dk + 8;
      const int km = dk[4], kp = dk[5];
      const T sink = sT[2 * k], cosk = sT[2 * k + 1];
      const T* bb = sB + (size_t)t * 7 * NFRE;
      const T* rr = sR + (size_t)t * RW;
      const T u = rr[REFR_U(NANG)], v = rr[REFR_V(NANG)];
      // direction refraction of (point, K): DRG*, DRD*, DRC* (ctuw.F90:424-452)
      T drgp, drgm, drdp = T(0), drdm = T(0), drcp = T(0), drcm = T(0);
      const T mask = rr[REFR_MASK(NANG) + slot];
      T s0 = T(0), omdd = T(0);
      {
#pragma clang fp contract(off)
        drgp = p.tanph * sK[2 * k];
        drgm = p.tanph * sK[2 * k + 1];
        if (IREFRA == 1) {
          drdp = (rr[k] + rr[kp]) * DELTH0;
          drdm = (rr[k] + rr[km]) * DELTH0;
        }
        if (cur) {
          drcp = mask * (rr[k] + rr[kp]) * DELTH0;
          drcm = mask * (rr[k] + rr[km]) * DELTH0;
          s0 = rr[NANG + k];
          omdd = rr[REFR_OMDD(NANG)];
        }
      }
      T bh0[VW], bh1[VW], by0[VW], by1[VW], bc0[VW], bom[VW], bwn[VW];
      IO::ld(bb + m, bh0); IO::ld(bb + NFRE + m, bh1); IO::ld(bb + 2 * NFRE + m, by0); IO::ld(bb + 3 * NFRE + m, by1);
      IO::ld(bb + 4 * NFRE + m, bc0); IO::ld(bb + 5 * NFRE + m, bom);
      // SDOT(K, M-1 .. M+VW) (propdot.F90:185-186) with the MPM clamps of ctuwupdt.F90:96-100
      T sd[VW + 2];
      if (cur) {
#pragma clang fp contract(off)
        IO::ld(bb + 6 * NFRE + m, bwn);
#pragma unroll
        for (int c = 0; c < VW; c++) sd[c + 1] = (s0 * bc0[c] + omdd * bom[c]) * bwn[c];
        const int ml = m > 0 ? m - 1 : 0, mh = m + VW < NR ? m + VW : NR - 1;
        sd[0] = (s0 * bb[4 * NFRE + ml] + omdd * bb[5 * NFRE + ml]) * bb[6 * NFRE + ml];
        sd[VW + 1] = (mh == m + VW) ? (s0 * bb[4 * NFRE + mh] + omdd * bb[5 * NFRE + mh]) * bb[6 * NFRE + mh] : sd[VW];
        if (m + VW > NR) {  // vector straddles NFRE_RED (only when NFRE_RED is not a multiple of VW): clamp inside too
#pragma unroll
          for (int c = 0; c < VW; c++)
            if (m + c >= NR) sd[c + 1] = sd[NR - m];
        }
      }
      T dthp_[VW], dthm_[VW], fdp_[VW], fdm_[VW];
#pragma unroll
      for (int c = 0; c < VW; c++) {
#pragma clang fp contract(off)
        fdp_[c] = fdm_[c] = T(0);
        if (IREFRA == 0) {
          dthp_[c] = drgp * bc0[c] + drcp;
          dthm_[c] = drgm * bc0[c] + drcm;
        } else {
          dthp_[c] = drgp * bc0[c] + bom[c] * drdp + drcp;
          dthm_[c] = drgm * bc0[c] + bom[c] * drdm + drcm;
        }
        if (cur) {
          const int mc = m + c;
          fdp_[c] = mask * (sd[c + 1] + (mc + 1 < NR ? sd[c + 2] : sd[c + 1])) * sDF[mc];
          fdm_[c] = mask * (sd[c + 1] + sd[c]) * sDF[NFRE + mc];
        }
      }
#if !ECWAM_HIP_CTU_STRICT
      if constexpr (!CHECK && !OBS) {
        // The lanes whose advection velocity keeps the sign of the group velocity (all of them unless the current exceeds it somewhere): the
        // upwind weights in the hoisted form (ctu.h: ctu_fast_wgen), two frequencies per packed operand.  Decided per LANE from the lane's own
        // numbers, so that a point's arithmetic does not depend on which points share its wavefront (decompositions stay bit-identical).
        typedef CtuV2<T> F;
        const T* bf = sBF + (size_t)t * 4 * NFRE + m;
        T xa[VW], xb[VW], ya[VW], yb[VW];
        IO::ld(bf + jx0 * NFRE, xa); IO::ld(bf + (1 - jx0) * NFRE, xb); IO::ld(bf + (2 + jy0) * NFRE, ya); IO::ld(bf + (3 - jy0) * NFRE, yb);
        const T* e = sQ + 12 * t;
        const T su = sink < T(0) ? -e[8] : e[8];
        const T sva = cosk < T(0) ? -e[9 + jy0] : e[9 + jy0], svb = cosk < T(0) ? -e[10 - jy0] : e[10 - jy0];
        const T rfr = T(1) / FRATIO;
        CtuFastGenW<T> fw[(VW + 1) / 2];
        bool neg = false;
        constexpr int c1 = VW > 1 ? 1 : 0;
#pragma unroll
        for (int c = 0; c < VW; c += 2) {
#define P2(a) F{a[c], a[c + c1]}
          bool n_;
          fw[c / 2] = ctu_fast_wgen<T>(P2(xa), P2(xb), P2(ya), P2(yb), m_abs(sink), m_abs(cosk), su, sva, svb, p.zd, xdella, p.ga, e[0], e[1], p.wl[jy0],
                                       e[2 + jy0], p.wc[kc[0]], e[4 + kc[0]], P2(dthp_), P2(dthm_), P2(fdp_), P2(fdm_), FRATIO, rfr, n_);
          neg |= n_;
        }
        if (!neg) {
          const T* fo = f1 + own;
          T fown[VW], flon[VW], fla1[VW], fla2[VW], fco1[VW], fco2[VW], fkm[VW], fkp[VW], fmm[VW], fmp[VW], r[VW];
          IO::ld(fo + el, fown);
          IO::ld(f1 + (size_t)q[1 + jx0] * N + el, flon);
          IO::ld(f1 + (size_t)q[3 + 2 * jy0] * N + el, fla1);
          IO::ld(f1 + (size_t)q[4 + 2 * jy0] * N + el, fla2);
          IO::ld(f1 + (size_t)q[7 + 2 * kc[0]] * N + el, fco1);
          IO::ld(f1 + (size_t)q[8 + 2 * kc[0]] * N + el, fco2);
          IO::ld(fo + km * NFRE + m, fkm);
          IO::ld(fo + kp * NFRE + m, fkp);
          if (cur) {      // the frequency neighbours of the own spectrum, MPM clamps of ctuwupdt.F90:96-100
            const T flo = fo[k * NFRE + (m > 0 ? m - 1 : 0)];
            const int mh = m + VW < NR ? m + VW : NR - 1;
            const T fhi = fo[k * NFRE + mh];
#pragma unroll
            for (int c = 0; c < VW; c++) {
              fmm[c] = (c == 0) ? flo : fown[c - 1];
              fmp[c] = (c == VW - 1) ? fhi : fown[c + 1];
              if (m + c + 1 > NR - 1) fmp[c] = fown[c];
            }
          } else {
#pragma unroll
            for (int c = 0; c < VW; c++) fmm[c] = fmp[c] = T(0);
          }
#pragma unroll
          for (int c = 0; c < VW; c += 2) {
            const F rr2 = ctu_fast_apply_gen<T>(fw[c / 2], P2(fown), P2(flon), P2(fla1), P2(fla2), P2(fco1), P2(fco2), P2(fkm), P2(fkp), P2(fmm), P2(fmp));
#undef P2
            r[c] = rr2.x;
            if (VW > 1) r[c + c1] = rr2.y;
          }
          if (m + VW > m1) {
#pragma unroll
            for (int c = 0; c < VW; c++)
              if (m + c >= m1) r[c] = (copy_rest & 1) ? fown[c] : f3[own + el + c];
          }
          IO::st_stream(f3 + own + el, r);
          continue;
        }
      }
#endif
      CtuGenW<T> w[VW];
      bool fail = false, flipped = false;
#pragma unroll
      for (int c = 0; c < VW; c++) {
        CtuBase<T> b;
        b.h[0] = bh0[c]; b.h[1] = bh1[c]; b.hy[0] = by0[c]; b.hy[1] = by1[c]; b.cg0 = bc0[c];
        const T dthp = dthp_[c], dthm = dthm_[c], fdp = fdp_[c], fdm = fdm_[c];
        {
          bool fl_;
          const bool f_ = ctu_wgen<T, CHECK>(b, sink, cosk, p, xdella, delpro, CMTODEG, jx0, jx1, jy0, jy1, kc, cur, u, v, dthp, dthm, fdp, fdm, FRATIO, w[c], fl_);
          if (m + c < m1) { fail |= f_; flipped |= fl_; }
        }
        if (OBS && !CHECK) {  // ctuw.F90:703-733, after the checks
#pragma clang fp contract(off)
          const T* o = sO + (size_t)t * 8 * NFRE + m + c;
#pragma unroll
          for (int ic = 0; ic < 2; ic++) {
            w[c].wlon[ic] = w[c].wlon[ic] * o[(2 + ic) * NFRE];
            w[c].wlat[ic][0] = w[c].wlat[ic][0] * o[ic * NFRE];
            w[c].wlat[ic][1] = w[c].wlat[ic][1] * o[ic * NFRE];
          }
#pragma unroll
          for (int icr = 0; icr < 4; icr++) {
            const T oc = o[(4 + kc[icr]) * NFRE];
            w[c].wcor[icr][0] = w[c].wcor[icr][0] * oc;
            w[c].wcor[icr][1] = w[c].wcor[icr][1] * oc;
          }
        }
      }
      if (CHECK) {
        if (fail) cflfail[q[0]] = 1;
        continue;
      }
      const T* fo = f1 + own;
      T fown[VW], r[VW];
      IO::ld(fo + el, fown);
      if (!cur) {
        T flon[VW], fla1[VW], fla2[VW], fco1[VW], fco2[VW], fkm[VW], fkp[VW];
        IO::ld(f1 + (size_t)q[1 + jx0] * N + el, flon);
        IO::ld(f1 + (size_t)q[3 + 2 * jy0] * N + el, fla1);
        IO::ld(f1 + (size_t)q[4 + 2 * jy0] * N + el, fla2);
        IO::ld(f1 + (size_t)q[7 + 2 * kc[0]] * N + el, fco1);
        IO::ld(f1 + (size_t)q[8 + 2 * kc[0]] * N + el, fco2);
        IO::ld(fo + km * NFRE + m, fkm);
        IO::ld(fo + kp * NFRE + m, fkp);
#pragma unroll
        for (int c = 0; c < VW; c++)
          r[c] = ctu_stencil(w[c].sumwn, sel2(jx0, w[c].wlon[0], w[c].wlon[1]), sel2(jy0, w[c].wlat[0][0], w[c].wlat[1][0]),
                             sel2(jy0, w[c].wlat[0][1], w[c].wlat[1][1]), w[c].wcor[0][0], w[c].wcor[0][1], w[c].wk[0], w[c].wk[2],
                             fown[c], flon[c], fla1[c], fla2[c], fco1[c], fco2[c], fkm[c], fkp[c]);
      } else if (__builtin_amdgcn_ballot_w64(flipped) == 0ull) {
        // no lane of this wave has an upwind switch: the downwind weights are all zero and only the five upwind space
        // neighbours can contribute, in the order the general sequence below visits them (adding a zero product is harmless)
#pragma clang fp contract(off)
        T flon[VW], fla1[VW], fla2[VW], fco1[VW], fco2[VW], fkm[VW], fkp[VW];
        IO::ld(f1 + (size_t)q[1 + jx0] * N + el, flon);
        IO::ld(f1 + (size_t)q[3 + 2 * jy0] * N + el, fla1);
        IO::ld(f1 + (size_t)q[4 + 2 * jy0] * N + el, fla2);
        IO::ld(f1 + (size_t)q[7 + 2 * kc[0]] * N + el, fco1);
        IO::ld(f1 + (size_t)q[8 + 2 * kc[0]] * N + el, fco2);
        IO::ld(fo + km * NFRE + m, fkm);
        IO::ld(fo + kp * NFRE + m, fkp);
        const T flo = fo[k * NFRE + (m > 0 ? m - 1 : 0)];
        const int mh = m + VW < NR ? m + VW : NR - 1;
        const T fhi = fo[k * NFRE + mh];
#pragma unroll
        for (int c = 0; c < VW; c++) {
          const T fm1 = (c == 0) ? flo : fown[c - 1];
          T fp1 = (c == VW - 1) ? fhi : fown[c + 1];
          if (m + c + 1 > NR - 1) fp1 = fown[c];
          T a = (T(1) - w[c].sumwn) * fown[c];
          a = a + sel2(jx0, w[c].wlon[0], w[c].wlon[1]) * flon[c];
          a = a + sel2(jy0, w[c].wlat[0][0], w[c].wlat[1][0]) * fla1[c];
          a = a + w[c].wcor[0][0] * fco1[c];
          a = a + sel2(jy0, w[c].wlat[0][1], w[c].wlat[1][1]) * fla2[c];
          a = a + w[c].wcor[0][1] * fco2[c];
          a = a + w[c].wk[0] * fkm[c];
          a = a + w[c].wm[0] * fm1;
          a = a + w[c].wk[2] * fkp[c];
          a = a + w[c].wm[2] * fp1;
          r[c] = a;
        }
      } else {
#pragma clang fp contract(off)
#pragma unroll
        for (int c = 0; c < VW; c++) r[c] = (T(1) - w[c].sumwn) * fown[c];
        // one neighbour: loaded when any of the VW weights is non-zero
#define TERMV(WEXPR, PTR)                                                     \
  {                                                                           \
    bool any_ = false;                                                        \
    _Pragma("unroll") for (int c = 0; c < VW; c++) { any_ |= ((WEXPR) != T(0)); } \
    if (any_) {                                                               \
      T fn_[VW];                                                              \
      IO::ld((PTR), fn_);                                                     \
      _Pragma("unroll") for (int c = 0; c < VW; c++) { const T wv_ = (WEXPR); if (wv_ != T(0)) r[c] = r[c] + wv_ * fn_[c]; } \
    }                                                                         \
  }
#pragma unroll
        for (int ic = 0; ic < 2; ic++) TERMV(w[c].wlon[ic], f1 + (size_t)q[1 + ic] * N + el);
#pragma unroll
        for (int icl = 0; icl < 2; icl++) {
#pragma unroll
          for (int ic = 0; ic < 2; ic++) TERMV(w[c].wlat[ic][icl], f1 + (size_t)q[3 + 2 * ic + icl] * N + el);
#pragma unroll
          for (int icr = 0; icr < 4; icr++) TERMV(w[c].wcor[icr][icl], f1 + (size_t)q[7 + 2 * kc[icr] + icl] * N + el);
        }
#undef TERMV
        // direction and frequency neighbours of the own spectrum: IC = -1 then +1, WKPMN before WMPMN (propags2.F90:170-186)
        T fkm[VW], fkp[VW];
        IO::ld(fo + km * NFRE + m, fkm);
        IO::ld(fo + kp * NFRE + m, fkp);
        const T flo = fo[k * NFRE + (m > 0 ? m - 1 : 0)];
        const int mh = m + VW < NR ? m + VW : NR - 1;
        const T fhi = fo[k * NFRE + mh];
#pragma unroll
        for (int c = 0; c < VW; c++) {
          const T fm1 = (c == 0) ? flo : fown[c - 1];
          T fp1 = (c == VW - 1) ? fhi : fown[c + 1];
          if (m + c + 1 > NR - 1) fp1 = fown[c];            // MPM(M,+1) = MIN(NFRE_RED, M+1)
          if (w[c].wk[0] != T(0)) r[c] = r[c] + w[c].wk[0] * fkm[c];
          if (w[c].wm[0] != T(0)) r[c] = r[c] + w[c].wm[0] * fm1;
          if (w[c].wk[2] != T(0)) r[c] = r[c] + w[c].wk[2] * fkp[c];
          if (w[c].wm[2] != T(0)) r[c] = r[c] + w[c].wm[2] * fp1;
        }
      }
      if (m + VW > m1) {  // partial vector at the end of the range (VW == 1 never gets here)
#pragma unroll
        for (int c = 0; c < VW; c++)
          if (m + c >= m1) r[c] = (copy_rest & 1) ? fown[c] : f3[own + el + c];
      }
      IO::st_stream(f3 + own + el, r);
    }
  }
}

// CURMASK of the second CTUW call (ctuw.F90:117-131): 0 where the first call failed
template <typename T>
__global__ void k_curmask(int n, int NANG, int slot, const int* __restrict__ cflfail, T* __restrict__ refr) {
  const int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij < n) refr[(size_t)ij * REFR_W(NANG) + REFR_MASK(NANG) + slot] = (cflfail && cflfail[ij]) ? T(0) : T(1);
}

// NEWWIND (newwind.F90:126-161)
template <typename T>
__global__ void k_newwind(const DevTab<T>* __restrict__ tab, int n, T* __restrict__ ff, const T* __restrict__ ffn, int icode_wnd) {
  int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= n) return;
  T* f = ff + (size_t)ij * ECWAM_HIP_NFF;
  const T* g = ffn + (size_t)ij * ECWAM_HIP_NFF;
  if (icode_wnd == 3) {   // ICODE_WND = ICODE_CPL when LWCOU, ICODE otherwise (newwind.F90:120-124)
    const T wght = T(1) / m_max(tab->WSPMIN_RESET_TAUW, tab->EPSMIN);
    T u = g[3];
    f[3] = u;
    if (u < tab->WSPMIN_RESET_TAUW) {
      T tl = wght * (tab->ACD + tab->BCD * u) * (u * u * u);
      f[8] = m_min(f[8], tl);
    }
  } else {  // friction-velocity forcing (newwind.F90:141-149), USTMIN_RESET_TAUW = 0.08 (yowwind.F90:20)
    const T us = g[7];
    f[7] = us;
    const T r = tab->ALPHA / f[12];
    T tw = (us * us) * (T(1) - r * r);
    if (us < T(0.08)) tw = T(0);
    f[8] = tw;
  }
  f[1] = g[1]; f[0] = g[0]; f[4] = g[4]; f[2] = g[2]; f[13] = g[13]; f[5] = g[5]; f[6] = g[6];
}

// chunked FL1(NPROMA,N2,N3,NCHNK) (Fortran order) -> points[ij][n2][n3] and back
template <typename T>
__global__ void k_chunks_to_points(const T* __restrict__ ch, T* __restrict__ pt, int nproma, int nchnk, int npts, int n2, int n3) {
  const long long total = (long long)npts * n2 * n3;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int ij = (int)(g / (n2 * n3));
    int e = (int)(g - (long long)ij * n2 * n3);
    int a = e / n3, b = e - a * n3;
    int ichnk = ij / nproma, iprm = ij - ichnk * nproma;
    pt[g] = ch[(((size_t)ichnk * n3 + b) * n2 + a) * nproma + iprm];
  }
}
template <typename T>
__global__ void k_points_to_chunks(const T* __restrict__ pt, T* __restrict__ ch, int nproma, int nchnk, int npts, int n2, int n3) {
  const long long total = (long long)nchnk * nproma * n2 * n3;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int iprm = (int)(g % nproma);
    long long r = g / nproma;
    int a = (int)(r % n2);
    r /= n2;
    int b = (int)(r % n3);
    int ichnk = (int)(r / n3);
    int ij = ichnk * nproma + iprm;
    if (ij >= npts) ij = ichnk * nproma;  // pad lanes replicate lane 1 (propag_wam.F90:388-398)
    ch[g] = pt[((size_t)ij * n2 + a) * n3 + b];
  }
}

// F(:,:,m0:m1-1) of rows [0,n) from src to dst: FL1_EXT(:,:,1:IFRELFMAX) <- FL3_EXT between the fast-wave sub-steps
// (propag_wam.F90:287-291); one thread per (row, K, m)
template <typename T>
__global__ void k_copy_freq_range(const T* __restrict__ src, T* __restrict__ dst, int n, int NANG, int NFRE, int m0, int m1,
                                  int dst_nfre) {
  // dst rows have dst_nfre frequencies per direction (NFRE: same layout; less: a compact fast-wave buffer)
  const int nm = m1 - m0;
  const long long total = (long long)n * NANG * nm;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const long long run = g / nm;
    const int m = m0 + (int)(g - run * nm);
    dst[run * dst_nfre + m] = src[run * NFRE + m];
  }
}

template <typename T>
__global__ void k_pack_rows(const T* __restrict__ fl, const int* __restrict__ idx, int n, int rowlen, T* __restrict__ buf) {
  const long long total = (long long)n * rowlen;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int i = (int)(g / rowlen);
    int e = (int)(g - (long long)i * rowlen);
    buf[g] = fl[(size_t)idx[i] * rowlen + e];
  }
}

// PROENVHALO (proenvhalo.F90:69-107) on the device.  BUFFER_EXT[row][3 NFRE + 3]: WAVNUM(1:NFRE), CGROUP, OMOSNH2KD, DEPTH, UCUR, VCUR
// of every local row (DELLAM1 / COSPHM1 of the reference's buffer are static geometry the context already holds per row).
// k_proenv_pack fills the owned rows from the device-resident fields; the halo rows arrive through the library's MPEXCHNG on rows of
// 3 NFRE + 3 reals; k_proenv_unpack spreads all rows over the extended arrays the advection kernels read and fills the land slot
// (WVPRPT_LAND, BATHYMAX, U = V = 0: proenvhalo.F90:99-106).
template <typename T>
__global__ void k_proenv_pack(int n, int NFRE, const T* __restrict__ wvprpt, const T* __restrict__ omosnh2kd, const T* __restrict__ depth,
                              const T* __restrict__ ucur, const T* __restrict__ vcur, T* __restrict__ buf) {
  const int RL = 3 * NFRE + 3;
  const long long total = (long long)n * RL;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = (int)(g / RL), e = (int)(g - (long long)ij * RL);
    T v;
    if (e < 2 * NFRE) v = wvprpt[(size_t)ij * ECWAM_HIP_NWPR * NFRE + e];     // WAVNUM and CGROUP are members 0 and 1 of the WVPRPT row
    else if (e < 3 * NFRE) v = omosnh2kd[(size_t)ij * NFRE + (e - 2 * NFRE)];
    else v = (e == 3 * NFRE) ? depth[ij] : (e == 3 * NFRE + 1 ? ucur[ij] : vcur[ij]);
    buf[g] = v;
  }
}
template <typename T>
__global__ void k_proenv_unpack(int nrows, int NFRE, const T* __restrict__ buf, const T* __restrict__ land, T* __restrict__ wavnum_ext,
                                T* __restrict__ cgroup_ext, T* __restrict__ omosnh2kd_ext, T* __restrict__ depth_ext, T* __restrict__ u_ext,
                                T* __restrict__ v_ext) {
  const int RL = 3 * NFRE + 3;
  const long long total = (long long)(nrows + 1) * RL;      // + the land slot, row index nrows
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = (int)(g / RL), e = (int)(g - (long long)ij * RL);
    const T v = ij < nrows ? buf[g] : land[e];
    if (e < NFRE) wavnum_ext[(size_t)ij * NFRE + e] = v;
    else if (e < 2 * NFRE) cgroup_ext[(size_t)ij * NFRE + (e - NFRE)] = v;
    else if (e < 3 * NFRE) omosnh2kd_ext[(size_t)ij * NFRE + (e - 2 * NFRE)] = v;
    else if (e == 3 * NFRE) depth_ext[ij] = v;
    else if (e == 3 * NFRE + 1) u_ext[ij] = v;
    else v_ext[ij] = v;
  }
}

// ---- host launchers (called from capi.hip) ---------------------------------------------------------
// dynamic LDS beyond the default 64 KB limit (double precision, 48 frequencies, obstructions: 80 KB of the CU's 160 KB)
template <typename K>
static inline void allow_lds(K kfn, size_t shmem) {
  if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
}
static inline int grid_for(long long total, int block = 256) {
  long long b = (total + block - 1) / block;
  const long long cap = 256LL * 16;  // 256 CUs x 16 blocks, grid-stride beyond (guide G11)
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

template <typename T>
void launch_propags2(const void* tab, const void* f1, void* f3, const int* klon, const int* klat, const int* kcor, const void* w,
                     int kijs, int kijl, int m0, int m1, int copy_rest, int dims, hipStream_t s) {
  const int NANG = dims >> 16, NFRE = (dims >> 8) & 0xFF, NR = dims & 0xFF, N = NANG * NFRE;  // packed by capi.hip
  long long total = (long long)(kijl - kijs) * N;
  if (total <= 0) return;
  constexpr int W = VecOf<T>::W;
  const bool aligned = ((uintptr_t)f1 % 16 == 0) && ((uintptr_t)f3 % 16 == 0) && ((uintptr_t)w % 16 == 0);
  if (aligned && NFRE % W == 0 && NR % W == 0 && m0 % W == 0 && m1 % W == 0) {
    hipLaunchKernelGGL(k_propags2_vec<T>, dim3(grid_for(total / W)), dim3(256), 0, s, (const DevTab<T>*)tab, (const T*)f1, (T*)f3,
                       klon, klat, kcor, (const T*)w, kijs, kijl, m0, m1, copy_rest);
    return;
  }
  hipLaunchKernelGGL(k_propags2<T>, dim3(grid_for(total)), dim3(256), 0, s, (const DevTab<T>*)tab, (const T*)f1, (T*)f3, klon,
                     klat, kcor, (const T*)w, kijs, kijl, m0, m1, copy_rest);
}
template <typename T>
void launch_ctuw(const void* tab, int n, int nland, int ngy, double delpro, int m0, int m1, const int* kxlt, const void* zdello,
                 double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat, const int* kcor,
                 void* wlat, void* wcor, const void* cg, const void* cosphm1, void* w, int* cflfail, int NANG, const void* obs,
                 hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_ctuwini<T>, dim3((n + 255) / 256), dim3(256), 0, s, n, nland, klat, kcor, (T*)wlat, (T*)wcor);
  long long total = (long long)n * NANG * (m1 - m0);
  hipLaunchKernelGGL(k_ctuw<T>, dim3(grid_for(total)), dim3(256), 0, s, (const DevTab<T>*)tab, n, ngy, (T)delpro, m0, m1, kxlt,
                     (const T*)zdello, (T)xdella, (const T*)cosph, (const T*)sinph, klon, klat, (const T*)wlat, (const T*)wcor,
                     (const T*)cg, (const T*)cosphm1, (T*)w, cflfail, (const T*)obs);
}
template <typename T>
void launch_propags2_otf(const void* tab, const void* f1, void* f3, int n_geom, int ngy, double delpro, const int* kxlt,
                         const void* zdello, double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat,
                         const int* kcor, const void* wlat, const void* wcor, const void* cg, const void* cosphm1, const int* order,
                         int kijs, int kijl, int m0, int m1, int copy_rest, int dims, const void* obs, int mlf, double delpro_lf, int in_k,
                         void* gout, int gout_k, const void* gin, int gin_k, int out_k, hipStream_t s) {
  const int NANG = dims >> 16, NFRE = (dims >> 8) & 0xFF;
  if (in_k <= 0) in_k = NFRE;
  const int n = kijl - kijs;
  if (n <= 0) return;
  const int ntiles = (n + OTF_TP - 1) / OTF_TP;
  const size_t shmem = OTF_TP * (sizeof(CtuPoint<T>) + 16 * sizeof(int)) +
                       ((size_t)OTF_TP * (obs ? 13 : 5) * NFRE + 6 * NANG + 8 * OTF_TP) * sizeof(T) + 8 * NANG * sizeof(int) + 16;
  int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
#ifdef ECWAM_HIP_DIAGNOSTICS
  { const char* e_ = getenv("ECWAM_HIP_OTF_GRID"); if (e_ && atoi(e_) > 0 && atoi(e_) < grid) grid = atoi(e_); }
#endif
  grid = (grid + 7) & ~7;  // whole rounds of the 8 XCDs
  constexpr int W = VecOf<T>::W;
  const bool aligned = ((uintptr_t)f1 % 16 == 0) && ((uintptr_t)f3 % 16 == 0);
#define OTF_ARGS                                                                                                              \
  (const DevTab<T>*)tab, (const T*)f1, (T*)f3, n_geom, ngy, (T)delpro, kxlt, (const T*)zdello, (T)xdella, (const T*)cosph,     \
      (const T*)sinph, klon, klat, kcor, (const T*)wlat, (const T*)wcor, (const T*)cg, (const T*)cosphm1, order, kijs, kijl, m0, \
      m1, copy_rest, ntiles, (const T*)obs, mlf, (T)delpro_lf, in_k, (T*)gout, gout_k, (const T*)gin, gin_k, out_k
  int vw = W;
#ifdef ECWAM_HIP_DIAGNOSTICS
  { const char* e_ = getenv("ECWAM_HIP_OTF_VW"); if (e_) vw = atoi(e_); }
#endif
  if (gout && (gout_k % W != 0 || (uintptr_t)gout % 16 != 0)) gout = nullptr;   // checked by the caller; never taken
  if (out_k <= 0) out_k = NFRE;
  const bool vec = vw >= W && aligned && NFRE % W == 0 && in_k % W == 0 && out_k % W == 0 && (!gin || (gin_k % W == 0 && (uintptr_t)gin % 16 == 0));   // a range boundary inside a vector is handled by the kernel
  if (obs) {  // LSUBGRID
    if (vec) { allow_lds(k_propags2_otf<T, W, true>, shmem); hipLaunchKernelGGL((k_propags2_otf<T, W, true>), dim3(grid), dim3(256), shmem, s, OTF_ARGS); }
    else { allow_lds(k_propags2_otf<T, 1, true>, shmem); hipLaunchKernelGGL((k_propags2_otf<T, 1, true>), dim3(grid), dim3(256), shmem, s, OTF_ARGS); }
  } else if (vec)
    { allow_lds(k_propags2_otf<T, W, false>, shmem); hipLaunchKernelGGL((k_propags2_otf<T, W, false>), dim3(grid), dim3(256), shmem, s, OTF_ARGS); }
  else if (vw >= 2 && aligned && NFRE % 2 == 0 && in_k % 2 == 0 && out_k % 2 == 0 && (!gin || gin_k % 2 == 0))
    { allow_lds(k_propags2_otf<T, 2, false>, shmem); hipLaunchKernelGGL((k_propags2_otf<T, 2, false>), dim3(grid), dim3(256), shmem, s, OTF_ARGS); }
  else
    { allow_lds(k_propags2_otf<T, 1, false>, shmem); hipLaunchKernelGGL((k_propags2_otf<T, 1, false>), dim3(grid), dim3(256), shmem, s, OTF_ARGS); }
#undef OTF_ARGS
}
template <typename T>
void launch_ctuwini_only(int n, int nland, const int* klat, const int* kcor, void* wlat, void* wcor, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_ctuwini<T>, dim3((n + 255) / 256), dim3(256), 0, s, n, nland, klat, kcor, (T*)wlat, (T*)wcor);
}
template <typename T>
void launch_propdot(const void* tab, int n, int nland, int irefra, const int* kxlt, const void* zdello, double xdella, const void* cosph,
                    const int* klon, const int* klat, const void* wlat, const void* cosphm1, const void* depth, const void* ue,
                    const void* ve, void* refr, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_propdot<T>, dim3((n + 127) / 128), dim3(128), 0, s, (const DevTab<T>*)tab, n, nland, irefra, kxlt,
                     (const T*)zdello, (T)xdella, (const T*)cosph, klon, klat, (const T*)wlat, (const T*)cosphm1, (const T*)depth,
                     (const T*)ue, (const T*)ve, (T*)refr);
}
template <typename T>
void launch_curmask(int n, int NANG, int slot, const int* cflfail, void* refr, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_curmask<T>, dim3((n + 255) / 256), dim3(256), 0, s, n, NANG, slot, cflfail, (T*)refr);
}
// f1 == nullptr: CFL / range checks only (cflfail); else the stencil
template <typename T>
void launch_propags2_gen(const void* tab, int irefra, const void* f1, void* f3, int ngy, double delpro, const int* kxlt,
                         const void* zdello, double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat,
                         const int* kcor, const void* wlat, const void* wcor, const void* cg, const void* om, const void* wn,
                         const void* cosphm1, const void* refr, int* cflfail, int slot, int kijs, int kijl, int m0, int m1,
                         int copy_rest, int dims, const void* obs, hipStream_t s) {
  const int NANG = dims >> 16, NFRE = (dims >> 8) & 0xFF, NR = dims & 0xFF;
  const int n = kijl - kijs;
  if (n <= 0) return;
  const int ntiles = (n + GEN_TP - 1) / GEN_TP;
  const size_t shmem = GEN_TP * (sizeof(CtuPoint<T>) + 16 * sizeof(int)) +
                       ((size_t)GEN_TP * (obs && f1 ? 19 : 11) * NFRE + (size_t)GEN_TP * (REFR_W(NANG) + 12) + 4 * NANG + 2 * NFRE) * sizeof(T) + 12 * NANG * sizeof(int) + 16;
  const int grid = ntiles < 256 * 16 ? ntiles : 256 * 16;
  constexpr int W = VecOf<T>::W;
  const bool aligned = ((uintptr_t)f1 % 16 == 0) && ((uintptr_t)f3 % 16 == 0);
#define GEN_ARGS                                                                                                               \
  (const DevTab<T>*)tab, irefra, (const T*)f1, (T*)f3, ngy, (T)delpro, kxlt, (const T*)zdello, (T)xdella, (const T*)cosph,       \
      (const T*)sinph, klon, klat, kcor, (const T*)wlat, (const T*)wcor, (const T*)cg, (const T*)om, (const T*)wn,             \
      (const T*)cosphm1, (const T*)refr, cflfail, slot, kijs, kijl, m0, m1, copy_rest, ntiles, (const T*)obs
  // 8 bytes per lane is the fastest width here (measured at O320 sp: 7.9 ms, against 9.2 ms at 16 bytes and 9.0 ms scalar):
  // the VW sets of 21 weights a thread keeps live cost more occupancy than the wider accesses save
  int vw = 2;
#ifdef ECWAM_HIP_DIAGNOSTICS
  { const char* e_ = getenv("ECWAM_HIP_GEN_VW"); if (e_) vw = atoi(e_); }
#endif
  const bool v4 = vw >= W && aligned && NFRE % W == 0 && NR % W == 0 && m0 % W == 0 && m1 % W == 0;
  const bool v2 = vw >= 2 && aligned && NFRE % 2 == 0 && NR % 2 == 0 && m0 % 2 == 0 && m1 % 2 == 0;
  if (!f1)
    { allow_lds(k_propags2_gen<T, 1, true, false>, shmem); hipLaunchKernelGGL((k_propags2_gen<T, 1, true, false>), dim3(grid), dim3(256), shmem, s, GEN_ARGS); }
  else if (obs) {
    if (v2) { allow_lds(k_propags2_gen<T, 2, false, true>, shmem); hipLaunchKernelGGL((k_propags2_gen<T, 2, false, true>), dim3(grid), dim3(256), shmem, s, GEN_ARGS); }
    else { allow_lds(k_propags2_gen<T, 1, false, true>, shmem); hipLaunchKernelGGL((k_propags2_gen<T, 1, false, true>), dim3(grid), dim3(256), shmem, s, GEN_ARGS); }
  } else if (v4)
    { allow_lds(k_propags2_gen<T, W, false, false>, shmem); hipLaunchKernelGGL((k_propags2_gen<T, W, false, false>), dim3(grid), dim3(256), shmem, s, GEN_ARGS); }
  else if (v2)
    { allow_lds(k_propags2_gen<T, 2, false, false>, shmem); hipLaunchKernelGGL((k_propags2_gen<T, 2, false, false>), dim3(grid), dim3(256), shmem, s, GEN_ARGS); }
  else
    { allow_lds(k_propags2_gen<T, 1, false, false>, shmem); hipLaunchKernelGGL((k_propags2_gen<T, 1, false, false>), dim3(grid), dim3(256), shmem, s, GEN_ARGS); }
#undef GEN_ARGS
}
template <typename T>
void launch_newwind(const void* tab, int n, void* ff, const void* ffn, int icode_wnd, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_newwind<T>, dim3((n + 255) / 256), dim3(256), 0, s, (const DevTab<T>*)tab, n, (T*)ff, (const T*)ffn, icode_wnd);
}
// The scalars of the CTU weights that the advecting tile load of k_implsch4 (implsch_v4.h::v4_advect_tile) takes ready-made: per point
// PT[ij][12] = ctu_point (ZDELLO, COSPHM1, 1 / (ZDELLO XDELLA), TAN(lat), DP(1:2), WLAT(1:2), WCOR(1:4)), per direction the factors of
// ctu_dirfac with TANPH = 1, SINTH, COSTH, then CMTODEG, and the index words JXO(K,1) | JYO(K,1) << 1 | KCR(K,1) << 2, KPM(K,-1), KPM(K,1).
// Computed HERE because the divisions in them are correctly rounded in this translation unit (the IMPLSCH units are built with the
// hardware reciprocal): the one-kernel step then forms the same weights, bit for bit, as k_propags2_otf.
// (delpro_lf > 0: the factors of ctu_dirfac for the fast waves' time step follow CMTODEG as [NANG][2].)
template <typename T>
__global__ void k_ctu_prep(const DevTab<T>* __restrict__ tab, int kijs, int kijl, int ngy, T delpro, T delpro_lf, const int* __restrict__ kxlt,
                           const T* __restrict__ zdello, T xdella, const T* __restrict__ cosph, const T* __restrict__ sinph,
                           const T* __restrict__ wlat, const T* __restrict__ wcor, const T* __restrict__ cosphm1, T* __restrict__ pt,
                           T* __restrict__ dirT, int* __restrict__ dirI) {
  const int NANG = tab->NANG;
  if (blockIdx.x == 0) {
    const T DELTH0 = T(0.25) * delpro / tab->DELTH;
    for (int k = threadIdx.x; k < NANG; k += blockDim.x) {
      T a, b;
      ctu_dirfac(tab, k, DELTH0, T(1), a, b);
      dirT[4 * k] = a; dirT[4 * k + 1] = b; dirT[4 * k + 2] = tab->SINTH[k]; dirT[4 * k + 3] = tab->COSTH[k];
      dirI[4 * k] = tab->JXO[k][0] | (tab->JYO[k][0] << 1) | (tab->KCR[k][0] << 2);
      dirI[4 * k + 1] = tab->KPM[k][0]; dirI[4 * k + 2] = tab->KPM[k][2]; dirI[4 * k + 3] = 0;
      if (delpro_lf > T(0)) {
        ctu_dirfac(tab, k, T(0.25) * delpro_lf / tab->DELTH, T(1), a, b);
        dirT[4 * NANG + 4 + 2 * k] = a; dirT[4 * NANG + 4 + 2 * k + 1] = b;
      }
    }
    if (threadIdx.x == 0) {
      dirT[4 * NANG] = T(360.0) / tab->CIRC;
      dirT[4 * NANG + 1] = dirT[4 * NANG + 2] = dirT[4 * NANG + 3] = T(0);
    }
  }
  for (int ij = kijs + blockIdx.x * blockDim.x + threadIdx.x; ij < kijl; ij += gridDim.x * blockDim.x) {
    const CtuPoint<T> p = ctu_point(ij, ngy, kxlt, zdello, xdella, cosph, sinph, wlat, wcor, cosphm1);
    T* o = pt + (size_t)ij * 12;
    o[0] = p.zd; o[1] = p.cpm1; o[2] = p.ga; o[3] = p.tanph; o[4] = p.dp[0]; o[5] = p.dp[1]; o[6] = p.wl[0]; o[7] = p.wl[1];
    o[8] = p.wc[0]; o[9] = p.wc[1]; o[10] = p.wc[2]; o[11] = p.wc[3];
  }
}
template <typename T>
void launch_ctu_prep(const void* tab, int kijs, int kijl, int ngy, double delpro, double delpro_lf, const int* kxlt, const void* zdello, double xdella,
                     const void* cosph, const void* sinph, const void* wlat, const void* wcor, const void* cosphm1, void* pt, void* dirT, int* dirI,
                     hipStream_t s) {
  const int n = kijl - kijs;
  hipLaunchKernelGGL(k_ctu_prep<T>, dim3(n > 0 ? grid_for(n) : 1), dim3(256), 0, s, (const DevTab<T>*)tab, kijs, kijl, ngy, (T)delpro, (T)delpro_lf, kxlt,
                     (const T*)zdello, (T)xdella, (const T*)cosph, (const T*)sinph, (const T*)wlat, (const T*)wcor, (const T*)cosphm1, (T*)pt,
                     (T*)dirT, dirI);
}
// NO SOURCE TERM CONTRIBUTION (wamintgr.F90:152-160): FL1 = MAX(FL1, EPSMIN), MIJ = NFRE, XLLWS = 0 on rows [kijs, kijl)
template <typename T>
__global__ void k_nosource(const DevTab<T>* __restrict__ tab, long long e0, long long e1, T* __restrict__ fl1, T* __restrict__ xllws, int kijs,
                           int kijl, int* __restrict__ mij) {
  const T eps = tab->EPSMIN;
  for (long long g = e0 + (long long)blockIdx.x * blockDim.x + threadIdx.x; g < e1; g += (long long)gridDim.x * blockDim.x) {
    if (fl1) fl1[g] = m_max(fl1[g], eps);   // fl1 == nullptr: the call before the source-term date (no clamp)
    xllws[g] = T(0);
  }
  for (int ij = kijs + blockIdx.x * blockDim.x + threadIdx.x; ij < kijl; ij += gridDim.x * blockDim.x) mij[ij] = tab->NFRE;
}
template <typename T>
void launch_nosource(const void* tab, int kijs, int kijl, int rowlen, void* fl1, void* xllws, int* mij, hipStream_t s) {
  if (kijl <= kijs) return;
  const long long e0 = (long long)kijs * rowlen, e1 = (long long)kijl * rowlen;
  hipLaunchKernelGGL(k_nosource<T>, dim3(grid_for(e1 - e0)), dim3(256), 0, s, (const DevTab<T>*)tab, e0, e1, (T*)fl1, (T*)xllws, kijs, kijl, mij);
}
template <typename T>
void launch_c2p(const void* ch, void* pt, int nproma, int nchnk, int npts, int n2, int n3, hipStream_t s) {
  long long total = (long long)npts * n2 * n3;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_chunks_to_points<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)ch, (T*)pt, nproma, nchnk, npts, n2, n3);
}
template <typename T>
void launch_p2c(const void* pt, void* ch, int nproma, int nchnk, int npts, int n2, int n3, hipStream_t s) {
  long long total = (long long)nchnk * nproma * n2 * n3;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_points_to_chunks<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)pt, (T*)ch, nproma, nchnk, npts, n2, n3);
}
template <typename T>
void launch_copy_freq_range(const void* src, void* dst, int n, int NANG, int NFRE, int m0, int m1, int dst_nfre, hipStream_t s) {
  long long total = (long long)n * NANG * (m1 - m0);
  if (total <= 0) return;
  hipLaunchKernelGGL(k_copy_freq_range<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)src, (T*)dst, n, NANG, NFRE, m0, m1,
                     dst_nfre);
}
template <typename T>
void launch_pack(const void* fl, const int* idx, int n, int rowlen, void* buf, hipStream_t s) {
  long long total = (long long)n * rowlen;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_pack_rows<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)fl, idx, n, rowlen, (T*)buf);
}

template <typename T>
void launch_proenv_pack(int n, int NFRE, const void* wvprpt, const void* om, const void* depth, const void* u, const void* v, void* buf, hipStream_t s) {
  const long long total = (long long)n * (3 * NFRE + 3);
  if (total <= 0) return;
  hipLaunchKernelGGL(k_proenv_pack<T>, dim3(grid_for(total)), dim3(256), 0, s, n, NFRE, (const T*)wvprpt, (const T*)om, (const T*)depth, (const T*)u,
                     (const T*)v, (T*)buf);
}
template <typename T>
void launch_proenv_unpack(int nrows, int NFRE, const void* buf, const void* land, void* wn, void* cg, void* om, void* dep, void* u, void* v, hipStream_t s) {
  const long long total = (long long)(nrows + 1) * (3 * NFRE + 3);
  hipLaunchKernelGGL(k_proenv_unpack<T>, dim3(grid_for(total)), dim3(256), 0, s, nrows, NFRE, (const T*)buf, (const T*)land, (T*)wn, (T*)cg, (T*)om,
                     (T*)dep, (T*)u, (T*)v);
}

#define INST(T)                                                                                                                   \
  template void launch_proenv_pack<T>(int, int, const void*, const void*, const void*, const void*, const void*, void*, hipStream_t); \
  template void launch_proenv_unpack<T>(int, int, const void*, const void*, void*, void*, void*, void*, void*, void*, hipStream_t); \
  template void launch_propags2<T>(const void*, const void*, void*, const int*, const int*, const int*, const void*, int, int,   \
                                   int, int, int, int, hipStream_t);                                                              \
  template void launch_ctuw<T>(const void*, int, int, int, double, int, int, const int*, const void*, double, const void*,       \
                               const void*, const int*, const int*, const int*, void*, void*, const void*, const void*, void*,   \
                               int*, int, const void*, hipStream_t);                                                              \
  template void launch_newwind<T>(const void*, int, void*, const void*, int, hipStream_t);                                        \
  template void launch_ctu_prep<T>(const void*, int, int, int, double, double, const int*, const void*, double, const void*, const void*, const void*, const void*, const void*, void*, void*, int*, hipStream_t);\
  template void launch_nosource<T>(const void*, int, int, int, void*, void*, int*, hipStream_t);                                  \
  template void launch_propdot<T>(const void*, int, int, int, const int*, const void*, double, const void*, const int*,         \
                                  const int*, const void*, const void*, const void*, const void*, const void*, void*, hipStream_t); \
  template void launch_ctuwini_only<T>(int, int, const int*, const int*, void*, void*, hipStream_t);                               \
  template void launch_curmask<T>(int, int, int, const int*, void*, hipStream_t);                                                      \
  template void launch_propags2_gen<T>(const void*, int, const void*, void*, int, double, const int*, const void*, double,        \
                                       const void*, const void*, const int*, const int*, const int*, const void*, const void*,   \
                                       const void*, const void*, const void*, const void*, const void*, int*, int, int, int, int, \
                                       int, int, int, const void*, hipStream_t);                                                                    \
  template void launch_propags2_otf<T>(const void*, const void*, void*, int, int, double, const int*, const void*, double,        \
                                       const void*, const void*, const int*, const int*, const int*, const void*, const void*,   \
                                       const void*, const void*, const int*, int, int, int, int, int, int, const void*, int, double,   \
                                       int, void*, int, const void*, int, int, hipStream_t);                                                                              \
  template void launch_c2p<T>(const void*, void*, int, int, int, int, int, hipStream_t);                                          \
  template void launch_p2c<T>(const void*, void*, int, int, int, int, int, hipStream_t);                                          \
  template void launch_copy_freq_range<T>(const void*, void*, int, int, int, int, int, int, hipStream_t);                             \
  template void launch_pack<T>(const void*, const int*, int, int, void*, hipStream_t);
INST(float)
INST(double)
