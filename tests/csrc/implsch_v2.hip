// TEST INFRASTRUCTURE: launcher of the one-point-per-wavefront IMPLSCH kernel (implsch_v2.h: lane = direction, two LDS tiles, fused sweep --
// every configuration), the second implementation the GPU tests compare k_implsch4 with.  Built by tests/v2lib.py into
// tests/csrc/libecwam_v2.so; takes the device tables of a product context (ecwam_hip_device_tables).
#include <hip/hip_runtime.h>
#include "implsch_common.h"
#include "implsch_v2.h"

template <typename T>
int launch_implsch(const void* tab, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                   double* w2n, void* dbg, int NANG, int NFRE, int variant, hipStream_t s) {
  const int n = kijl - kijs;
  if (n <= 0) return 0;
  const bool norma = (variant & 16) != 0;  // LLNORMAGAM, packed by capi.hip
  const bool rare = (variant & 32) != 0;   // any of LLGCBZ0 / LCIWA2 / LCIWA3 / LCISCAL / LWNEMOCOU: the build that carries those branches
  const size_t per_wave = (size_t)(2 * NFRE * NANG + NSC) * sizeof(T);
  // waves (= points) per block: the choice that fits the most waves into the 160 KiB of LDS of a CU; ties go to the larger
  // block, which amortises the lane-per-point scalar stages over more points
  static const int cands[] = {3, 1};
  const int ncand = 2;
  // register-limited residency: compiled for 3 (sp) / 2 (dp) waves per SIMD
  const int capw = 4 * ((sizeof(T) == 4) ? 4 : 2);
  int wpb = 1, best = 0;
  for (int i = 0; i < ncand; i++) {
    const int cand = cands[i];
    int nb = (int)((160 * 1024) / (per_wave * cand));
    if (nb * cand > capw) nb = capw / cand;
    const int waves = nb * cand;
    if (waves > best) { best = waves; wpb = cand; }
  }
  if (best == 0) return 1;
  size_t shmem = per_wave * wpb;
#ifdef ECWAM_HIP_DIAGNOSTICS   // timing builds only (tools/build_diag.sh)
  { const char* e_ = getenv("ECWAM_HIP_IMPLSCH_WPB"); if (e_) { const int w = atoi(e_); if (w == 3 || w == 1) { wpb = w; shmem = per_wave * wpb; } } }
  { const char* e_ = getenv("ECWAM_HIP_IMPLSCH_PADLDS"); if (e_) { const int pad = atoi(e_); if (pad > 0 && shmem + (size_t)pad <= 160 * 1024) shmem += (size_t)pad; } }
#endif
  const int blocks = (n + wpb - 1) / wpb;
#define LAUNCHK(KFN)                                                                                                         \
  do {                                                                                                                       \
    if (shmem > 64 * 1024) (void)hipFuncSetAttribute((const void*)KFN, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
    hipLaunchKernelGGL(KFN, dim3(blocks), dim3(64 * wpb), shmem, s, (const DevTab<T>*)tab, kijs, kijl, (T*)fl1,              \
                       (const T*)wvprpt, (T*)ff, (T*)intf, mij, (T*)xllws, w2n, (T*)dbg);                                         \
  } while (0)
#define LAUNCH2(W)                                                                                                           \
  do {                                                                                                                       \
    if (norma && rare) LAUNCHK((k_implsch2<T, W, true, true>));                                                              \
    else if (norma) LAUNCHK((k_implsch2<T, W, true, false>));                                                                \
    else if (rare) LAUNCHK((k_implsch2<T, W, false, true>));                                                                 \
    else LAUNCHK((k_implsch2<T, W, false, false>));                                                                          \
  } while (0)
  if (wpb == 3) LAUNCH2(3);
  else LAUNCH2(1);
#undef LAUNCH2
#undef LAUNCHK
  return 0;
}
template int launch_implsch<float>(const void*, int, int, void*, const void*, void*, void*, int*, void*, double*, void*, int, int, int, hipStream_t);
template int launch_implsch<double>(const void*, int, int, void*, const void*, void*, void*, int*, void*, double*, void*, int, int, int, hipStream_t);

// variant: 16 = LLNORMAGAM, 32 = the build with the rare branches (LLGCBZ0, sea ice, NEMO coupling, IPHYS 0, ISNONLIN, ICODE 1 / 2, LWVFLX_SNL = F).
// Returns 0 when launched, 1 when the spectral size does not fit the LDS tiling, 2 on a launch error.
extern "C" int ecwam_v2_implsch(const void* dtab, int dp, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij,
                                void* xllws, double* w2n, void* dbg, int nang, int nfre, int variant, void* stream) {
  const int rc = dp ? launch_implsch<double>(dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, w2n, dbg, nang, nfre, variant, (hipStream_t)stream)
                    : launch_implsch<float>(dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, w2n, dbg, nang, nfre, variant, (hipStream_t)stream);
  if (rc) return rc;
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
