// Integrated output parameters on the device: the OUTBLOCK parameters the reference validates a run
// with -- significant wave height, mean direction, mean period, peak period (outblock.F90:204,223-263 with FEMEAN
// femean.F90:84-121, STHQ sthq.F90:75-120 and DOMINANT_PERIOD dominant_period.F90:76-112) -- and the OUTWNORM statistics (average / minimum / maximum / non-missing count, outwnorm.F90), so
// that a device-resident run can be checked without copying the spectra back.
#include "dev.h"

// One wavefront per point.  The spectrum goes through an LDS tile [M][NANG|1]: lane = M sums MAX(F,EPSMIN) over K in the
// reference's order (FEMEAN), lane = K sums F*DFIM over M in the reference's order (STHQ).
template <typename T>
__global__ void __launch_bounds__(256) k_outbs(const DevTab<T>* __restrict__ tp, int kijs, int kijl, const T* __restrict__ fl1,
                                               T zmiss, T* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char ob_smem[];
  const DevTab<T>& tb = *tp;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ij = kijs + blockIdx.x * 4 + wave;
  if (ij >= kijl) return;  // wave-uniform, no block barrier below
  const int NANG = tb.NANG, NFRE = tb.NFRE, NAP = NANG | 1, N = NANG * NFRE;
  T* sF = reinterpret_cast<T*>(ob_smem) + (size_t)wave * NFRE * NAP;
  const T* g = fl1 + (size_t)ij * N;
  for (int e = lane; e < N; e += 64) {
    const int kk = e / NFRE, mm = e - kk * NFRE;
    sF[mm * NAP + kk] = g[e];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const bool actm = lane < NFRE, actk = lane < NANG;
  // FEMEAN
  T t2 = T(0);
  if (actm) {
    const T* p = sF + lane * NAP;
    t2 = m_max(p[0], tb.EPSMIN);
    for (int kk = 1; kk < NANG; kk++) t2 = t2 + m_max(p[kk], tb.EPSMIN);
  }
  T EM, FM;
  usum2(actm ? t2 * tb.DFIM[lane] : T(0), actm ? tb.DFIMOFR[lane] * t2 : T(0), EM, FM);
  const T tl = lane_get(t2, NFRE - 1);
  EM = EM + tb.WETAIL * tb.FR[NFRE - 1] * tb.DELTH * tl;
  FM = FM + tb.FRTAIL * tb.DELTH * tl;
  FM = EM / FM;
  FM = m_max(FM, tb.FR[0]);
  // STHQ
  T temp = T(0);
  if (actk)
    for (int m = 0; m < NFRE; m++) temp = temp + sF[m * NAP + lane] * tb.DFIM[m];
  T SI, CI;
  usum2(actk ? tb.SINTH[lane] * temp : T(0), actk ? tb.COSTH[lane] * temp : T(0), SI, CI);
  if (CI == T(0)) CI = tb.EPSMIN;
  T THQ = m_atan2(SI, CI);
  if (THQ < T(0)) THQ = THQ + tb.ZPI;
  // DOMINANT_PERIOD: lane = K finds its maximum, lane = M sums the cropped directions in the reference's order
  T fmx = T(0);
  if (actk)
    for (int m = 0; m < NFRE; m++) fmx = m_max(fmx, sF[m * NAP + lane]);
  const T FCROP = T(0.1) * umax(actk ? fmx : T(0));
  T f1d = T(0);
  if (actm) {
    const T* p = sF + lane * NAP;
    for (int kk = 0; kk < NANG; kk++)
      if (p[kk] > FCROP) f1d = f1d + p[kk] * tb.DELTH;
    f1d = (f1d * f1d) * (f1d * f1d);
  }
  T EM4, DP;
  usum2(actm ? tb.DFIM[lane] * f1d : T(0), actm ? tb.DFIMFR[lane] * f1d : T(0), EM4, DP);
  DP = (EM4 > T(0) && DP > tb.EPSMIN) ? EM4 / DP : T(0);
  if (lane == 0) {
    T* o = out + (size_t)ij * 5;
    const T DEG = T(180.0) / tb.PI;
    o[0] = T(4) * m_sqrt(m_max(EM, T(0)));
    T d = DEG * THQ + T(180.0);
    d = d - T(360.0) * T((int)(d / T(360.0)));  // MOD(.,360) for d >= 0
    o[1] = d;
    o[2] = (FM > T(0)) ? T(1) / FM : zmiss;
    o[3] = EM;
    o[4] = (DP > T(0)) ? DP : zmiss;
  }
}

// OUTWNORM: partial sums in double (the reference accumulates WNORM in JWRU), two deterministic stages
template <typename T>
__global__ void __launch_bounds__(256) k_norm_partial(const T* __restrict__ f, int stride, int n, T zmiss, double* __restrict__ part) {
  __shared__ double s_sum[256], s_min[256], s_max[256], s_cnt[256];
  double sum = 0.0, mn = 1e300, mx = -1e300, cnt = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const T v = f[(size_t)i * stride];
    if (v != zmiss) { const double d = (double)v; sum += d; mn = d < mn ? d : mn; mx = d > mx ? d : mx; cnt += 1.0; }
  }
  const int t = threadIdx.x;
  s_sum[t] = sum; s_min[t] = mn; s_max[t] = mx; s_cnt[t] = cnt;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if (t < h) {
      s_sum[t] += s_sum[t + h]; s_cnt[t] += s_cnt[t + h];
      s_min[t] = s_min[t + h] < s_min[t] ? s_min[t + h] : s_min[t];
      s_max[t] = s_max[t + h] > s_max[t] ? s_max[t + h] : s_max[t];
    }
    __syncthreads();
  }
  if (t == 0) { double* p = part + (size_t)blockIdx.x * 4; p[0] = s_sum[0]; p[1] = s_min[0]; p[2] = s_max[0]; p[3] = s_cnt[0]; }
}
__global__ void k_norm_final(const double* __restrict__ part, int nb, double* __restrict__ res) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double sum = 0.0, mn = 1e300, mx = -1e300, cnt = 0.0;
  for (int b = 0; b < nb; b++) {
    const double* p = part + (size_t)b * 4;
    sum += p[0]; cnt += p[3];
    mn = p[1] < mn ? p[1] : mn; mx = p[2] > mx ? p[2] : mx;
  }
  res[0] = cnt > 0.0 ? sum / cnt : 0.0; res[1] = cnt > 0.0 ? mn : 0.0; res[2] = cnt > 0.0 ? mx : 0.0; res[3] = cnt;
}

template <typename T>
int launch_outbs(const void* tab, int kijs, int kijl, const void* fl1, double zmiss, void* out, int NANG, int NFRE, hipStream_t s) {
  const int n = kijl - kijs;
  if (n <= 0) return 0;
  const size_t shmem = (size_t)4 * NFRE * (NANG | 1) * sizeof(T);
  if (shmem > 64 * 1024) return 1;
  hipLaunchKernelGGL(k_outbs<T>, dim3((n + 3) / 4), dim3(256), shmem, s, (const DevTab<T>*)tab, kijs, kijl, (const T*)fl1, (T)zmiss, (T*)out);
  return 0;
}
template <typename T>
void launch_norm(const void* f, int stride, int n, double zmiss, double* scratch, int nb, hipStream_t s) {
  hipLaunchKernelGGL(k_norm_partial<T>, dim3(nb), dim3(256), 0, s, (const T*)f, stride, n, (T)zmiss, scratch + 4);
  hipLaunchKernelGGL(k_norm_final, dim3(1), dim3(64), 0, s, scratch + 4, nb, scratch);
}
template int launch_outbs<float>(const void*, int, int, const void*, double, void*, int, int, hipStream_t);
template int launch_outbs<double>(const void*, int, int, const void*, double, void*, int, int, hipStream_t);
template void launch_norm<float>(const void*, int, int, double, double*, int, hipStream_t);
template void launch_norm<double>(const void*, int, int, double, double*, int, hipStream_t);
