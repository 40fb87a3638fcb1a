// Advection kernels: CTU weight construction (ctuwini.F90 + ctuw.F90, IREFRA=0) and the PROPAGS2 stencil
// (propags2.F90:99-121) on the point-major device layout FL[ij][K][M] (see include/ecwam_hip.h).
//
// PROPAGS2 is HBM-bound (16 flop per (ij,K,M) against 10 words): one thread per (ij,K,M) element,
// consecutive threads on consecutive M then K, so the eight weight streams W[ij][w][K*NR+M] and the
// own spectrum are read/written fully coalesced; neighbour spectra are gathered as contiguous
// NFRE-long runs (one run per direction), which the XCD L2 / Infinity Cache serve (a point's spectrum
// is re-used by its ~8 neighbours within one or two latitude rows).
#include "dev.h"

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
    // same association order as propags2.F90:107-116
    T r = (T(1) - w[wb]) * f1[own + km_e];
    r = r + w[wb + ws] * f1[(size_t)ilon * N + km_e];
    r = r + w[wb + 2 * ws] * f1[(size_t)ilat1 * N + km_e];
    r = r + w[wb + 3 * ws] * f1[(size_t)ilat2 * N + km_e];
    r = r + w[wb + 4 * ws] * f1[(size_t)icor1 * N + km_e];
    r = r + w[wb + 5 * ws] * f1[(size_t)icor2 * N + km_e];
    r = r + w[wb + 6 * ws] * f1[own + km * NFRE + m];
    r = r + w[wb + 7 * ws] * f1[own + kp * NFRE + m];
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
    V r = vfirst(LDV(wp), LDV(f1 + own + e));
    r = vmul_add(r, LDV(wp + ws), LDV(f1 + (size_t)ilon * N + e));
    r = vmul_add(r, LDV(wp + 2 * ws), LDV(f1 + (size_t)ilat1 * N + e));
    r = vmul_add(r, LDV(wp + 3 * ws), LDV(f1 + (size_t)ilat2 * N + e));
    r = vmul_add(r, LDV(wp + 4 * ws), LDV(f1 + (size_t)icor1 * N + e));
    r = vmul_add(r, LDV(wp + 5 * ws), LDV(f1 + (size_t)icor2 * N + e));
    r = vmul_add(r, LDV(wp + 6 * ws), LDV(f1 + own + km * NFRE + m));
    r = vmul_add(r, LDV(wp + 7 * ws), LDV(f1 + own + kp * NFRE + m));
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

// ctuw.F90:146-275 (space weights), :407-484 (great-circle refraction WKPMN), :536-608 (range checks, SUMWN)
// restricted to what PROPAGS2 reads for IREFRA=0 (ISSU=ISSV=1 => DXDW=DYDW=0).  One thread per (ij,K,M).
template <typename T>
__global__ void __launch_bounds__(256) k_ctuw(const DevTab<T>* __restrict__ tab, int n, int ngy, T delpro, int m0, int m1,
                                              const int* __restrict__ kxlt, const T* __restrict__ zdello, T xdella,
                                              const T* __restrict__ cosph, const T* __restrict__ sinph,
                                              const int* __restrict__ klon, const int* __restrict__ klat,
                                              const T* __restrict__ wlat, const T* __restrict__ wcor,
                                              const T* __restrict__ cg, const T* __restrict__ cosphm1, T* __restrict__ w,
                                              int* __restrict__ cflfail) {
  const int NANG = tab->NANG, NFRE = tab->NFRE, NR = tab->NFRE_RED;
  const int nm = m1 - m0;
  const long long total = (long long)n * NANG * nm;
  const T CMTODEG = T(360.0) / tab->CIRC;
  const T DELTH0 = T(0.25) * delpro / tab->DELTH;  // ctuw.F90:407
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = (int)(g / (NANG * nm));
    const int r = (int)(g - (long long)ij * NANG * nm);
    const int k = r / nm, m = m0 + (r - k * nm);
    const int ky = kxlt[ij];
    const T zd = zdello[ky];
    const T cg0 = cg[(size_t)ij * NFRE + m];
    const T cpm1 = cosphm1[ij];
    T adxp[2], adyp[2];
    bool fail = false;
    for (int ic = 0; ic < 2; ic++) {
      const T cgl = cg[(size_t)klon[ij * 2 + ic] * NFRE + m];
      const T cgx = T(0.5) * (cg0 + cgl) * tab->SINTH[k] * cpm1;
      const T wl = wlat[ij * 2 + ic];
      const T cgyp = wl * cg[(size_t)klat[(ij * 2 + ic) * 2 + 0] * NFRE + m] +
                     (T(1) - wl) * cg[(size_t)klat[(ij * 2 + ic) * 2 + 1] * NFRE + m];
      int kk = ky + 1 + 2 * (ic + 1) - 3;  // 1-based row of the neighbour latitude, ctuwini.F90:159-162
      kk = kk < 1 ? 1 : (kk > ngy ? ngy : kk);
      const T dp = cosph[kk - 1] * cpm1;
      const T cgy = T(0.5) * (cg0 + dp * cgyp) * tab->COSTH[k];
      adxp[ic] = m_abs(-delpro * cgx * CMTODEG);
      adyp[ic] = m_abs(-delpro * cgy * CMTODEG);
      if (adxp[ic] > zd || adyp[ic] > xdella) fail = true;
    }
    const int jx0 = tab->JXO[k][0], jx1 = tab->JXO[k][1], jy0 = tab->JYO[k][0], jy1 = tab->JYO[k][1];
    const T dxx = zd - adxp[jx1] - T(0);
    const T dyy = xdella - adyp[jy1] - T(0);
    const T ga = T(1) / (zd * xdella);
    const T wgt_lat = dxx * adyp[jy0] * ga;   // WEIGHT(JYO(K,1))
    const T wlatn1 = wlat[ij * 2 + jy0] * wgt_lat;
    const T wlatn2 = (T(1) - wlat[ij * 2 + jy0]) * wgt_lat;
    const T wlonn = dyy * adxp[jx0] * ga;
    const T wgt_cor = adxp[jx0] * adyp[jy0] * ga;  // WEIGHT(1)
    const int kc = tab->KCR[k][0];
    const T wcorn1 = wcor[ij * 4 + kc] * wgt_cor;
    const T wcorn2 = (T(1) - wcor[ij * 4 + kc]) * wgt_cor;
    T sumwn = (zd * (T(0) + adyp[jy1]) + xdella * (adxp[jx1] + T(0)) - (T(0) + adxp[jx1]) * (T(0) + adyp[jy1])) * ga;
    // direction weights
    const int kp1 = tab->KPM[k][2], km1 = tab->KPM[k][0];
    const T sp = DELTH0 * (tab->SINTH[k] + tab->SINTH[kp1]) / tab->R;
    const T sm = DELTH0 * (tab->SINTH[k] + tab->SINTH[km1]) / tab->R;
    const T tanph = sinph[ky] / cosph[ky];
    const T dthp = tanph * sp * cg0 + T(0);
    const T dthm = tanph * sm * cg0 + T(0);
    const T wk0 = (dthp + m_abs(dthp)) + (m_abs(dthm) - dthm);
    const T wkp = -dthp + m_abs(dthp);
    const T wkm = dthm + m_abs(dthm);
    sumwn = sumwn + wk0;
    const T one = T(1), zero = T(0);
    if (wlatn1 > one || wlatn1 < zero || wlatn2 > one || wlatn2 < zero || wlonn > one || wlonn < zero || wcorn1 > one ||
        wcorn1 < zero || wcorn2 > one || wcorn2 < zero || wk0 > one || wk0 < zero || wkp > one || wkp < zero || wkm > one ||
        wkm < zero || sumwn > one || sumwn < zero)
      fail = true;
    if (fail) cflfail[ij] = 1;
    const size_t ws = (size_t)NANG * NR;
    const size_t wb = (size_t)ij * 8 * ws + (size_t)k * NR + m;
    w[wb] = sumwn;
    w[wb + ws] = wlonn;
    w[wb + 2 * ws] = wlatn1;
    w[wb + 3 * ws] = wlatn2;
    w[wb + 4 * ws] = wcorn1;
    w[wb + 5 * ws] = wcorn2;
    w[wb + 6 * ws] = wkm;
    w[wb + 7 * ws] = wkp;
  }
}

// NEWWIND (newwind.F90:126-161, ICODE_WND == 3)
template <typename T>
__global__ void k_newwind(const DevTab<T>* __restrict__ tab, int n, T* __restrict__ ff, const T* __restrict__ ffn) {
  int ij = blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= n) return;
  T* f = ff + (size_t)ij * ECWAM_HIP_NFF;
  const T* g = ffn + (size_t)ij * ECWAM_HIP_NFF;
  const T wght = T(1) / m_max(tab->WSPMIN_RESET_TAUW, tab->EPSMIN);
  T u = g[3];
  f[3] = u;
  if (u < tab->WSPMIN_RESET_TAUW) {
    T tl = wght * (tab->ACD + tab->BCD * u) * (u * u * u);
    f[8] = m_min(f[8], tl);
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

template <typename T>
__global__ void k_pack_rows(const T* __restrict__ fl, const int* __restrict__ idx, int n, int rowlen, T* __restrict__ buf) {
  const long long total = (long long)n * rowlen;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int i = (int)(g / rowlen);
    int e = (int)(g - (long long)i * rowlen);
    buf[g] = fl[(size_t)idx[i] * rowlen + e];
  }
}

// ---- host launchers (called from capi.hip) ---------------------------------------------------------
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
                 void* wlat, void* wcor, const void* cg, const void* cosphm1, void* w, int* cflfail, int NANG, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_ctuwini<T>, dim3((n + 255) / 256), dim3(256), 0, s, n, nland, klat, kcor, (T*)wlat, (T*)wcor);
  long long total = (long long)n * NANG * (m1 - m0);
  hipLaunchKernelGGL(k_ctuw<T>, dim3(grid_for(total)), dim3(256), 0, s, (const DevTab<T>*)tab, n, ngy, (T)delpro, m0, m1, kxlt,
                     (const T*)zdello, (T)xdella, (const T*)cosph, (const T*)sinph, klon, klat, (const T*)wlat, (const T*)wcor,
                     (const T*)cg, (const T*)cosphm1, (T*)w, cflfail);
}
template <typename T>
void launch_newwind(const void* tab, int n, void* ff, const void* ffn, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_newwind<T>, dim3((n + 255) / 256), dim3(256), 0, s, (const DevTab<T>*)tab, n, (T*)ff, (const T*)ffn);
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
void launch_pack(const void* fl, const int* idx, int n, int rowlen, void* buf, hipStream_t s) {
  long long total = (long long)n * rowlen;
  if (total <= 0) return;
  hipLaunchKernelGGL(k_pack_rows<T>, dim3(grid_for(total)), dim3(256), 0, s, (const T*)fl, idx, n, rowlen, (T*)buf);
}

#define INST(T)                                                                                                                   \
  template void launch_propags2<T>(const void*, const void*, void*, const int*, const int*, const int*, const void*, int, int,   \
                                   int, int, int, int, hipStream_t);                                                              \
  template void launch_ctuw<T>(const void*, int, int, int, double, int, int, const int*, const void*, double, const void*,       \
                               const void*, const int*, const int*, const int*, void*, void*, const void*, const void*, void*,   \
                               int*, int, hipStream_t);                                                                           \
  template void launch_newwind<T>(const void*, int, void*, const void*, hipStream_t);                                             \
  template void launch_c2p<T>(const void*, void*, int, int, int, int, int, hipStream_t);                                          \
  template void launch_p2c<T>(const void*, void*, int, int, int, int, int, hipStream_t);                                          \
  template void launch_pack<T>(const void*, const int*, int, int, void*, hipStream_t);
INST(float)
INST(double)
