// C ABI of libecwam_hip.so (include/ecwam_hip.h): context management and kernel launch entry points.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is loaded with dlopen when a communicator is asked for

#include "dev.h"
#include "implsch_adv_args.h"

static thread_local std::string g_err;
static int fail(const std::string& m) {
  g_err = m;
  return 1;
}
#define HIPCHK(x)                                                                        \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_));   \
  } while (0)

struct ecwam_hip_ctx {
  int real_bytes;
  int device;
  int NANG, NFRE, NFRE_RED;
  void* dtab = nullptr;  // DevTab<T> in device memory
  double* norm_scratch = nullptr;   // ecwam_hip_outwnorm: per-context reduction scratch (4 + 4 x 256 doubles)
  // fourth kernel generation (implsch_v4.h): DIA rotations K1 = K -+ r1, K2 = K +- r2, NSDSNTH = nh; ok = the tables have that structure
  int v4_ok = 0, v4_r1 = 0, v4_r2 = 0, v4_nh = 0, v4_shelter = 0;
  int implsch_last = 0; // generation the last ecwam_hip_implsch call launched (4; 0 before the first call)
  void* fin = nullptr;  // rows of scalars k_implsch4 hands to its finishing kernel, indexed by the point number; grown on demand
  size_t fin_bytes = 0;
  void* wi = nullptr;   // rows [ij][M][K] for the wind-input coefficient between the two kernels of the split k_implsch4 (only allocated
  size_t wi_bytes = 0;  // where the split runs: the double precision RARE builds, the "split" build variant of the library)
  // advection halo exchange (MPEXCHNG): peers, the owned rows each of them needs (concatenated in peer order) and where their rows
  // land; RCCL communicator + a stream of its own so that the exchange runs beside the interior stencil
  int rank = 0, nranks = 1;
  std::vector<int> peer, send_cnt, send_off, recv_dst0, recv_cnt;
  int n_send = 0, n_recv = 0;
  int* d_send_idx = nullptr;
  // one send buffer + event pair per row length in use (full rows, compact fast-wave rows, PROENVHALO rows): two exchanges of
  // different rows may be outstanding at once (propag_wam.F90:166,293) without the second pack waiting for the first one's sends
  struct HaloSlot {
    int rowlen = 0;
    void* buf = nullptr;
    size_t bytes = 0;
    hipEvent_t ev_packed = nullptr, ev_done = nullptr;
    bool used = false;      // an exchange has been posted from this buffer: a later pack waits for its sends (ev_done)
    bool pending = false;   // ... and no ecwam_hip_halo_finish has made a stream wait for it yet
    unsigned long long age = 0;
  };
  static constexpr int NSLOT = 4;
  HaloSlot slot[NSLOT];
  unsigned long long slot_clock = 0;
  ncclComm_t comm = nullptr;
  hipStream_t comm_stream = nullptr;
  ecwam_hip_params p;
  void* fast_g = nullptr;     // ecwam_hip_set_fastwave_copy: compact rows [ij][K][fast_gk] IMPLSCH / NOSOURCE also leave the new fast waves in
  int fast_gk = 0;
  // one-kernel step (ecwam_hip_propags2_implsch): per-point and per-direction scalars of the CTU weights (propag.hip::k_ctu_prep)
  std::string implsch_why;    // non-empty: no build of k_implsch4 covers the configuration (ecwam_hip_implsch refuses with this text)
  void* adv_pt = nullptr;     // [npts][12]
  size_t adv_pt_bytes = 0;
  void* adv_dir = nullptr;    // reals [4 NANG + 4], then ints [4 NANG]
  const void* obs = nullptr;  // LSUBGRID: device OBS[n_obs][8][NFRE] (ecwam_hip_set_obstructions), read by CTUW / PROPAGS2
  int n_obs = 0;
};

// launchers implemented in propag.hip / implsch.hip
template <typename T> void launch_propags2(const void*, const void*, void*, const int*, const int*, const int*, const void*, int, int, int, int, int, int, hipStream_t);
template <typename T> void launch_ctuw(const void*, int, int, int, double, int, int, const int*, const void*, double, const void*, const void*, const int*, const int*, const int*, void*, void*, const void*, const void*, void*, int*, int, const void*, hipStream_t);
template <typename T> void launch_ctuwini_only(int, int, const int*, const int*, void*, void*, hipStream_t);
template <typename T> void launch_propdot(const void*, int, int, int, const int*, const void*, double, const void*, const int*, const int*, const void*, const void*, const void*, const void*, const void*, void*, hipStream_t);
template <typename T> void launch_curmask(int, int, int, const int*, void*, hipStream_t);
template <typename T> void launch_propags2_gen(const void*, int, const void*, void*, int, double, const int*, const void*, double, const void*, const void*, const int*, const int*, const int*, const void*, const void*, const void*, const void*, const void*, const void*, const void*, int*, int, int, int, int, int, int, int, const void*, hipStream_t);
template <typename T> void launch_propags2_otf(const void*, const void*, void*, int, int, double, const int*, const void*, double, const void*, const void*, const int*, const int*, const int*, const void*, const void*, const void*, const void*, const int*, int, int, int, int, int, int, const void*, int, double, int, void*, int, const void*, int, int, hipStream_t);
template <typename T> void launch_copy_freq_range(const void*, void*, int, int, int, int, int, int, hipStream_t);
template <typename T> int launch_outbs(const void*, int, int, const void*, double, void*, int, int, hipStream_t);
template <typename T> void launch_norm(const void*, int, int, double, double*, int, hipStream_t);
template <typename T> void launch_newwind(const void*, int, void*, const void*, int, hipStream_t);
template <typename T> void launch_nosource(const void*, int, int, int, void*, void*, int*, hipStream_t);
template <typename T> void launch_c2p(const void*, void*, int, int, int, int, int, hipStream_t);
template <typename T> void launch_p2c(const void*, void*, int, int, int, int, int, hipStream_t);
template <typename T> void launch_pack(const void*, const int*, int, int, void*, hipStream_t);
template <typename T> void launch_proenv_pack(int, int, const void*, const void*, const void*, const void*, const void*, void*, hipStream_t);
template <typename T> void launch_proenv_unpack(int, int, const void*, const void*, void*, void*, void*, void*, void*, void*, hipStream_t);
template <typename T> int launch_implsch4(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
template <typename T> int launch_implsch4x(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
template <typename T> int launch_implsch4r(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, void*, int, void*, int, int, int, int, int, int, hipStream_t);
template <typename T> void launch_ctu_prep(const void*, int, int, int, double, double, const int*, const void*, double, const void*, const void*, const void*, const void*, const void*, void*, void*, int*, hipStream_t);
template <typename T> int launch_implsch4_adv(const void*, int, int, void*, const void*, void*, void*, int*, void*, void*, double*, const Implsch4AdvArgs*, int, int, int, int, int, int, hipStream_t);
int implsch4_adv_forms(int, int);
int implsch4_fin_row();
int implsch4_split_all();
int implsch4r_dp_split();

// Does the fourth kernel generation cover these tables?  It needs the pull-form DIA structure with K1W = K -+ r1, K11W = K1W -+ 1,
// K2W = K +- r2, K21W = K2W +- 1 (kh = 1 / 2) and saturation weights that depend on the tap only (init_sdiss_ardh.F90:88-94: they
// are COS**2 of a multiple of DELTH, equal for all K up to rounding).
template <typename T>
static void v4_probe(const DevTab<T>& h, ecwam_hip_ctx* c) {
  c->v4_ok = 0;
  const int NANG = h.NANG;
  if (!h.DIA_PULL || !h.V4_ROWS || (NANG & 1) || h.NFRE != 36) return;
  const int r1 = (NANG - h.K1W[0][0]) % NANG, r2 = h.K2W[0][0];
  if (h.D11[0] != -1 || h.D21[0] != 1 || h.D11[1] != 1 || h.D21[1] != -1) return;
  if (h.K1W[1][0] != r1 % NANG || h.K2W[1][0] != (NANG - r2) % NANG) return;
  T wmax = T(0);
  for (int t = 0; t < h.NTAP; t++) wmax = wmax > h.SATWEIGHTS[t][NANG / 2] ? wmax : h.SATWEIGHTS[t][NANG / 2];
  const T eps = sizeof(T) == 4 ? T(1.2e-7) : T(2.3e-16);
  for (int t = 0; t < h.NTAP; t++)
    for (int k = 0; k < NANG; k++) {
      const T d = h.SATWEIGHTS[t][k] - h.SATWEIGHTS[t][NANG / 2];
      if ((d < 0 ? -d : d) > T(16) * eps * wmax) return;
      const T e = h.SATWEIGHTS[t][NANG / 2] - h.SATWEIGHTS[h.NTAP - 1 - t][NANG / 2];   // symmetric about the centre tap
      if ((e < 0 ? -e : e) > T(16) * eps * wmax) return;
      if (h.INDICESSAT[t][k] != ((k - h.NSDSNTH + t) % NANG + NANG) % NANG) return;
    }
  c->v4_r1 = r1; c->v4_r2 = r2; c->v4_nh = h.NSDSNTH; c->v4_shelter = (h.TAUWSHELTER != T(0)); c->v4_ok = 1;
}

template <typename T>
static void cpv(T* dst, const void* src, int n) {
  if (src) for (int i = 0; i < n; i++) dst[i] = ((const T*)src)[i];
}

template <typename T>
static int build_tab(const ecwam_hip_params* p, const ecwam_hip_tables* t, DevTab<T>* d) {
  memset(d, 0, sizeof(*d));
  const int NANG = p->nang, NFRE = p->nfre, ML = p->mlsthg;
  d->NANG = NANG; d->NFRE = NFRE; d->NFRE_RED = p->nfre_red; d->NFRE_ODD = p->nfre_odd; d->IDELT = p->idelt;
  d->LLGCBZ0 = p->llgcbz0; d->LLNORMAGAM = p->llnormagam; d->LLCAPCHNK = p->llcapchnk; d->LBIWBK = p->lbiwbk;
  d->LICERUN = p->licerun; d->LMASKICE = p->lmaskice; d->LWAMRSETCI = p->lwamrsetci; d->LWVFLX_SNL = p->lwvflx_snl;
  d->LWFLUX = p->lwflux; d->LCFLX = (p->lwflux || p->lwfluxout || p->lwnemocou); d->LWNEMOCOU = p->lwnemocou; d->LWCOU = p->lwcou;
  d->LWCOUAST = p->lwcouast; d->LWNEMOCOUWRS = p->lwnemocouwrs;
  d->LWNEMOTAUOC = p->lwnemotauoc; d->LWNEMOCOUSEND = p->lwnemocousend; d->LWNEMOCOUSTK = p->lwnemocoustk;
  d->ICODE = p->icode; d->ISNONLIN = p->isnonlin; d->IPHYS = p->iphys; d->IDAMPING = p->idamping;
  d->LCISCAL = p->lciscal; d->LCIWA2 = p->lciwa2; d->LCIWA3 = p->lciwa3;
  d->LCIWA1 = p->lciwa1; d->LWNEMOCOUIBR = p->lwnemocouibr; d->LWNEMOCOUSTRN = p->lwnemocoustrn; d->NICT = p->nict; d->NICH = p->nich;
  for (int i = 0; i < 36 * 16; i++) d->CIDEAC[i] = T(0);
  if (p->lciwa1 && t->cideac) cpv(d->CIDEAC, t->cideac, p->nict * p->nich);
  d->DBG_SKIP = 0;
#ifdef ECWAM_HIP_DIAGNOSTICS   // timing builds only (tools/): phases of IMPLSCH skipped / early returns; never in the product library
  { const char* e_ = getenv("ECWAM_HIP_DEBUG_SKIP"); d->DBG_SKIP = e_ ? atoi(e_) : 0; }
#endif
  d->NSDSNTH = p->nsdsnth; d->NTAP = 2 * p->nsdsnth + 1; d->MFRSTLW = p->mfrstlw; d->MLSTHG = ML; d->KFRH = p->kfrh; d->NWAV_GC = p->nwav_gc;
#define S_(dst, src) d->dst = (T)p->src
  S_(ZIBRW_THRSH, zibrw_thrsh); S_(TICMIN, ticmin); S_(DTIC, dtic); S_(DHIC, dhic); S_(HICMIN, hicmin);
  S_(XIMP, ximp); S_(G, g); S_(GM1, gm1); S_(PI, pi); S_(ZPI, zpi); S_(ZPI4GM1, zpi4gm1); S_(ZPI4GM2, zpi4gm2); S_(EPSMIN, epsmin);
  S_(ROWATER, rowater); S_(ROWATERM1, rowaterm1); S_(EPSUS, epsus); S_(EPSU10, epsu10); S_(ACD, acd); S_(BCD, bcd);
  S_(ACDLIN, acdlin); S_(BCDLIN, bcdlin); S_(CDMAX, cdmax); S_(TAUOCMIN, tauocmin); S_(TAUOCMAX, tauocmax);
  S_(PHIEPSMIN, phiepsmin); S_(PHIEPSMAX, phiepsmax); S_(WSEMEAN_MIN, wsemean_min); S_(CIRC, circ); S_(R, r_earth);
  S_(FRATIO, fratio); S_(WETAIL, wetail); S_(FRTAIL, frtail); S_(WP1TAIL, wp1tail); S_(FRIC, fric); S_(DELTH, delth);
  S_(FLOGSPRDM1, flogsprdm1); S_(XKAPPA, xkappa); S_(XNLEV, xnlev); S_(RNU, rnu); S_(RNUM, rnum);
  S_(BETAMAXOXKAPPA2, betamaxoxkappa2); S_(BMAXOKAP, bmaxokap); S_(GAMNCONST, gamnconst); S_(ZALP, zalp); S_(ALPHA, alpha);
  S_(ALPHAMIN, alphamin); S_(ALPHAMAX, alphamax); S_(CHNKMIN_U, chnkmin_u); S_(TAUWSHELTER, tauwshelter); S_(DTHRN_A, dthrn_a);
  S_(DTHRN_U, dthrn_u); S_(TAILFACTOR, tailfactor); S_(TAILFACTOR_PM, tailfactor_pm); S_(ANG_GC_A, ang_gc_a);
  S_(ANG_GC_B, ang_gc_b); S_(ANG_GC_C, ang_gc_c); S_(RN1_RN, rn1_rn); S_(ALPHAPMAX, alphapmax); S_(CDIS, cdis); S_(DELTA_SDIS, delta_sdis); S_(CDISVIS, cdisvis); S_(CDICWA, cdicwa); S_(ZALPFACB, zalpfacb); S_(ZALPFACX, zalpfacx); S_(SWELLF, swellf); S_(SWELLF2, swellf2);
  S_(SWELLF3, swellf3); S_(SWELLF4, swellf4); S_(SWELLF5, swellf5); S_(SWELLF6, swellf6); S_(SWELLF7, swellf7);
  S_(SWELLF7M1, swellf7m1); S_(Z0RAT, z0rat); S_(Z0TUBMAX, z0tubmax); S_(ABMIN, abmin); S_(ABMAX, abmax); S_(SDSBR, sdsbr);
  S_(SSDSC2, ssdsc2); S_(SSDSC3, ssdsc3); S_(SSDSC4, ssdsc4); S_(SSDSC5, ssdsc5); S_(SSDSC6, ssdsc6); S_(MICHE, miche);
  S_(EGRCRV, egrcrv); S_(AFCRV, afcrv); S_(BFCRV, bfcrv); S_(X0TAUHF, x0tauhf); S_(EPS1, eps1); S_(FLMIN, flmin);
  S_(CITHRSH, cithrsh); S_(CIBLOCK, ciblock); S_(CITHRSH_TAIL, cithrsh_tail); S_(ZALPWRS, zalpwrs); S_(BATHYMAX, bathymax);
  S_(WSPMIN, wspmin); S_(WSPMIN_RESET_TAUW, wspmin_reset_tauw); S_(DAL1, dal1); S_(DAL2, dal2);
  S_(XLOGKRATIOM1_GC, xlogkratiom1_gc); S_(SQRTGOSURFT, sqrtgosurft);
#undef S_
  cpv(d->FR, t->fr, NFRE); cpv(d->DFIM, t->dfim, NFRE); cpv(d->DFIMOFR, t->dfimofr, NFRE); cpv(d->DFIMFR, t->dfimfr, NFRE);
  cpv(d->DFIM_SIM, t->dfim_sim, NFRE); cpv(d->RHOWG_DFIM, t->rhowg_dfim, NFRE); cpv(d->ZPIFR, t->zpifr, NFRE);
  cpv(d->FR5, t->fr5, NFRE); cpv(d->COFRM4, t->cofrm4, NFRE); cpv(d->FLMAX, t->flmax, NFRE);
  cpv(d->TH, t->th, NANG); cpv(d->COSTH, t->costh, NANG); cpv(d->SINTH, t->sinth, NANG);
  cpv(d->WTAUHF, t->wtauhf, JTOT);
  cpv(d->SWELLFT + 1, t->swellft, ECWAM_HIP_IAB);
  for (int i = 0; i < ML; i++) { d->IKP[i] = t->ikp[i]; d->IKP1[i] = t->ikp1[i]; d->IKM[i] = t->ikm[i]; d->IKM1[i] = t->ikm1[i]; }
  cpv(d->AF11, t->af11, ML);
  for (int k = 0; k < NANG; k++)
    for (int kh = 0; kh < 2; kh++) {
      d->K1W[kh][k] = t->k1w[k * 2 + kh] - 1; d->K2W[kh][k] = t->k2w[k * 2 + kh] - 1;
      d->K11W[kh][k] = t->k11w[k * 2 + kh] - 1; d->K21W[kh][k] = t->k21w[k * 2 + kh] - 1;
    }
  // structure check for the pull-form DIA kernel: K1W/K2W rotations, K11W/K21W = K1W/K2W +-1, row offsets -4,-3,+2,+3
  d->DIA_PULL = 1;
  for (int kh = 0; kh < 2 && d->DIA_PULL; kh++) {
    const int s1 = ((d->K1W[kh][0] - 0) % NANG + NANG) % NANG, s2 = ((d->K2W[kh][0] - 0) % NANG + NANG) % NANG;
    const int e1 = ((d->K11W[kh][0] - d->K1W[kh][0]) % NANG + NANG) % NANG, e2 = ((d->K21W[kh][0] - d->K2W[kh][0]) % NANG + NANG) % NANG;
    if (!((e1 == 1 || e1 == NANG - 1) && (e2 == 1 || e2 == NANG - 1))) { d->DIA_PULL = 0; break; }
    d->D11[kh] = (e1 == 1) ? 1 : -1;
    d->D21[kh] = (e2 == 1) ? 1 : -1;
    for (int k = 0; k < NANG; k++) {
      if (d->K1W[kh][k] != (k + s1) % NANG || d->K2W[kh][k] != (k + s2) % NANG || d->K11W[kh][k] != (k + s1 + e1) % NANG ||
          d->K21W[kh][k] != (k + s2 + e2) % NANG) { d->DIA_PULL = 0; break; }
      d->IK1[kh][(k + s1) % NANG] = k;
      d->IK2[kh][(k + s2) % NANG] = k;
    }
  }
  for (int i = 0; i < ML && d->DIA_PULL; i++) {
    const int MC = i + 1;
    if (t->ikp[i] != MC + 2 || t->ikp1[i] != MC + 3) d->DIA_PULL = 0;
    if (t->ikm[i] != MC - 4 || t->ikm1[i] != MC - 3) d->DIA_PULL = 0;
  }
  if (ML != NFRE + 4 || p->kfrh != 8 || p->mfrstlw != -3) d->DIA_PULL = 0;
  bool sep_ok = true;      // RNLCOEF factors as frequency factor x angular weight (checked below)
  for (int i = 0; i < ML; i++) {
    for (int j = 0; j < 5; j++) d->INLCOEF[i][j] = t->inlcoef[i * 5 + j] - 1;
    for (int j = 0; j < 25; j++) d->RNLCOEF[i][j] = ((const T*)t->rnlcoef)[i * 25 + j];
  }
  {
    // separable form of the interaction coefficients: CL11 + ACL1 = 1 and CL21 + ACL2 = 1 (nlweigt.F90:166-169), so the frequency factor of a
    // coefficient pair is its sum and the angular weights are the ratios, taken where the factors are largest
    double cl11 = 1.0, acl1 = 0.0, cl21 = 1.0, acl2 = 0.0, bp = 0.0, bm = 0.0;
    for (int i = 0; i < ML; i++) {
      const double pa = (double)d->RNLCOEF[i][5], pb = (double)d->RNLCOEF[i][6], ma = (double)d->RNLCOEF[i][17], mb = (double)d->RNLCOEF[i][18];
      if (fabs(pa + pb) > bp) { bp = fabs(pa + pb); cl11 = pa / (pa + pb); acl1 = pb / (pa + pb); }
      if (fabs(ma + mb) > bm) { bm = fabs(ma + mb); cl21 = ma / (ma + mb); acl2 = mb / (ma + mb); }
    }
    const double ang[8] = {cl11, acl1, cl21, acl2, cl11 * cl11, acl1 * acl1, cl21 * cl21, acl2 * acl2};
    for (int j = 0; j < 8; j++) d->DIAANG[j] = (T)ang[j];
    for (int i = 0; i < ML; i++) {
      const T* R = d->RNLCOEF[i];
      const double gp = (double)R[1] + R[2], gp1 = (double)R[3] + R[4], gm = (double)R[13] + R[14], gm1 = (double)R[15] + R[16];
      const double fp = (double)R[5] + R[6], fp1 = (double)R[8] + R[7], fm = (double)R[17] + R[18], fm1 = (double)R[20] + R[19];
      const double w[12] = {gp, gp1, gm, gm1, fp, fp1, fp * fp, fp1 * fp1, fm, fm1, fm * fm, fm1 * fm1};
      for (int j = 0; j < 12; j++) d->DIAW[i][j] = (T)w[j];
    }
    for (int i = 0; i < ML; i++)
      for (int j = 0; j < 4; j++) d->DIAW[i][12 + j] = (i + 1 < ML) ? d->DIAW[i + 1][j] : T(0);
    // The kernel never reads RNLCOEF(2:25): it forms the coefficients as frequency factor x angular weight (inisnonlin.F90:186-241).  Tables
    // that do not factor that way (not from INISNONLIN, or modified) would silently give another Snl: rebuild every word and compare;
    // V4_ROWS = 0 makes ecwam_hip_create refuse the tables with the reason.
    const double tol = (sizeof(T) == 4 ? 1.2e-7 : 2.3e-16) * 64.0;
    for (int i = 0; i < ML && sep_ok; i++) {
      const T* R = d->RNLCOEF[i];
      const T* W = d->DIAW[i];
      const double gp = W[0], gp1 = W[1], gm = W[2], gm1 = W[3], fp = W[4], fp1 = W[5], fm = W[8], fm1 = W[9];
      const double want[24] = {gp * cl11, gp * acl1, gp1 * cl11, gp1 * acl1, fp * cl11, fp * acl1, fp1 * acl1, fp1 * cl11,
                               (fp * cl11) * (fp * cl11), (fp * acl1) * (fp * acl1), (fp1 * cl11) * (fp1 * cl11), (fp1 * acl1) * (fp1 * acl1),
                               gm * cl21, gm * acl2, gm1 * cl21, gm1 * acl2, fm * cl21, fm * acl2, fm1 * acl2, fm1 * cl21,
                               (fm * cl21) * (fm * cl21), (fm * acl2) * (fm * acl2), (fm1 * cl21) * (fm1 * cl21), (fm1 * acl2) * (fm1 * acl2)};
      for (int j = 0; j < 24; j++) {
        const double scale = fabs(want[j]) > fabs((double)R[1 + j]) ? fabs(want[j]) : fabs((double)R[1 + j]);
        if (fabs(want[j] - (double)R[1 + j]) > tol * scale) sep_ok = false;
      }
    }
  }
  d->V4_ROWS = sep_ok ? 1 : 0;
  for (int i = 0; i < ML; i++) {
    // gather set (words 0..11): FTAIL, GW1..GW8, AF11; scatter set (words 12..27): FKLAMPA .. FKLAP22, FKLAMMA .. FKLAM22
    for (int j = 0; j < 32; j++) d->DIACF[i][j] = T(0);
    d->DIACF[i][0] = d->RNLCOEF[i][0];
    for (int j = 0; j < 4; j++) { d->DIACF[i][1 + j] = d->RNLCOEF[i][1 + j]; d->DIACF[i][5 + j] = d->RNLCOEF[i][13 + j]; }
    d->DIACF[i][9] = d->AF11[i];
    for (int j = 0; j < 8; j++) { d->DIACF[i][12 + j] = d->RNLCOEF[i][5 + j]; d->DIACF[i][20 + j] = d->RNLCOEF[i][17 + j]; }
    const int MC = i + 1;
    auto cl = [&](int r) { return (r < 1 ? 1 : (r > NFRE ? NFRE : r)) - 1; };
    {  // snonlin.F90:236-240: the tail factor applies outside MFR1STFR < MC < MFRLSTFR
      const int MFR1STFR = -p->mfrstlw + 1, MFRLSTFR = NFRE - p->kfrh + MFR1STFR;
      d->DIACF[i][31] = (MC > MFR1STFR && MC < MFRLSTFR) ? T(1) : d->RNLCOEF[i][0];
    }
    d->DIACF[i][10] = d->ZPIFR[cl(MC - 3)];
    const int mu = cl(MC - 5);
    d->DIACF[i][28] = d->COFRM4[mu]; d->DIACF[i][29] = d->FLMAX[mu]; d->DIACF[i][30] = d->RHOWG_DFIM[mu]; d->DIACF[i][11] = d->ZPIFR[mu];
    const int want[5] = {cl(MC), cl(MC + 2), cl(MC + 3), cl(MC - 4), cl(MC - 3)};
    for (int j = 0; j < 5; j++) if (d->INLCOEF[i][j] != want[j]) d->V4_ROWS = 0;
  }
  for (int i = 0; i <= ML; i++) {
    const int k = i < ML ? i : ML - 1;
    static const int w7[7] = {9, 10, 11, 28, 29, 30, 31};
    for (int j = 0; j < 7; j++) d->DIAREC[i][j] = d->DIACF[k][w7[j]];
    d->DIAREC[i][7] = T(0);
    for (int j = 0; j < 12; j++) d->DIAREC[i][8 + j] = d->DIAW[k][4 + j];
  }
  for (int m = 0; m < NFRE; m++) {
    T* r = d->SINROW[m];
    r[0] = d->ZPIFR[m]; r[1] = d->DFIM[m];
    r[2] = -d->SWELLF5 * T(2) * std::sqrt(T(2) * d->RNU * d->ZPIFR[m]);
    r[3] = -d->SWELLF * T(16) * (d->ZPIFR[m] * d->ZPIFR[m]) / d->G;
    r[4] = d->RHOWG_DFIM[m]; r[5] = d->DFIMOFR[m]; r[6] = d->DFIMFR[m]; r[7] = T(0);
  }
  const int ntap = 2 * p->nsdsnth + 1;
  for (int k = 0; k < NANG; k++)
    for (int j = 0; j < ntap; j++) {
      d->INDICESSAT[j][k] = t->indicessat[k * ntap + j] - 1;
      d->SATWEIGHTS[j][k] = ((const T*)t->satweights)[k * ntap + j];
    }
  for (int k = 0; k < NANG; k++) {
    for (int j = 0; j < 3; j++) d->KPM[k][j] = t->kpm[k * 3 + j] - 1;
    for (int j = 0; j < 2; j++) { d->JXO[k][j] = t->jxo[k * 2 + j] - 1; d->JYO[k][j] = t->jyo[k * 2 + j] - 1; }
    for (int j = 0; j < 4; j++) d->KCR[k][j] = t->kcr[k * 4 + j] - 1;
  }
  const int ng = p->nwav_gc;
  cpv(d->XK_GC + 1, t->xk_gc, ng); cpv(d->XKM_GC + 1, t->xkm_gc, ng); cpv(d->OMEGA_GC + 1, t->omega_gc, ng);
  cpv(d->OMXKM3_GC + 1, t->omxkm3_gc, ng); cpv(d->CM_GC + 1, t->cm_gc, ng); cpv(d->C2OSQRTVG_GC + 1, t->c2osqrtvg_gc, ng);
  cpv(d->XKMSQRTVGOC2_GC + 1, t->xkmsqrtvgoc2_gc, ng); cpv(d->OM3GMKM_GC + 1, t->om3gmkm_gc, ng);
  cpv(d->DELKCC_GC_NS + 1, t->delkcc_gc_ns, ng); cpv(d->DELKCC_OMXKM3_GC + 1, t->delkcc_omxkm3_gc, ng);
  return 0;
}

// ---- lane-primitive self test (one wave): every cross-lane helper of dev.h against serial sums over LDS ----------------
template <typename T>
__global__ void k_selftest(int* nbad) {
  __shared__ T v[4][64];
  const int lane = threadIdx.x;
  T q[4];
  for (int j = 0; j < 4; j++) {
    q[j] = (lane < 36) ? T((lane * (7 + 2 * j) + 3 * j) % 23 + j) : T(0);  // small integers: every summation order is exact
    v[j][lane] = q[j];
  }
  __syncthreads();
  T ref[4], mx[4];
  for (int j = 0; j < 4; j++) {
    ref[j] = T(0); mx[j] = T(0);
    for (int l = 0; l < 64; l++) { ref[j] += v[j][l]; mx[j] = m_max(mx[j], v[j][l]); }
  }
  int bad = 0;
  T a, b, c, d;
  bad += (usum(q[0]) != ref[0]) + (umax(q[1]) != mx[1]);
  usum2(q[0], q[1], a, b); bad += (a != ref[0]) + (b != ref[1]);
  usum4(q[0], q[1], q[2], q[3], a, b, c, d); bad += (a != ref[0]) + (b != ref[1]) + (c != ref[2]) + (d != ref[3]);
  umax2(q[2], q[3], a, b); bad += (a != mx[2]) + (b != mx[3]);
  bad += (lane_get(q[0], 5) != v[0][5]);
  bad += (lane_pull(q[1], (lane + 9) & 63) != v[1][(lane + 9) & 63]);
  bad += (rot_up(q[2], lane, 36) != ((lane < 36) ? v[2][(lane + 35) % 36] : (lane == 36 ? v[2][35] : T(0))));
  // the hot-loop replacements of the library's EXP / SQRT / 1 / SQRT (dev.h) against the library over their argument ranges
  for (int i = 0; i < 256; i++) {
    const T u = T(lane * 256 + i) / T(16384);                      // 0 .. 1
    const T xe = T(-700) * u * u + T(3) * (T(1) - u) - T(1.5);     // EXP: -701.5 .. 1.5, dense near zero
    const T tol = sizeof(T) == 8 ? T(1e-15) : T(1e-4);             // sp: v_exp_f32 of x log2(e), |x| 1.2e-7 relative
    const T ee = m_exp(xe), eg = f_exp(xe);
    bad += !(m_abs(eg - ee) <= tol * ee + (sizeof(T) == 8 ? T(1e-320) : T(1e-37)));
    const T xs = m_exp(T(sizeof(T) == 8 ? 600 : 80) * (T(2) * u - T(1)));   // SQRT, 1 / SQRT: 1e-260 .. 1e260 (sp 1e-35 .. 1e35)
    const T ts = sizeof(T) == 8 ? T(9e-16) : T(3e-7);
    const T sr = m_sqrt(xs);
    bad += !(m_abs(f_sqrt(xs) - sr) <= ts * sr) + !(m_abs(f_rsq(xs) * sr - T(1)) <= T(2) * ts);
  }
  bad += (f_sqrt(T(0)) != T(0)) + (f_exp(T(-1000)) != T(0));
  if (bad) atomicAdd(nbad, bad);
}


// ---- RCCL through dlopen: the library has no link-time dependency on it (single-GPU hosts never load it) ---------------------------
struct RcclApi {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
  if (g_rccl.h) return 0;
  // By SONAME: in a process that has already mapped an RCCL (PyTorch-ROCm links its bundled librccl.so.1) this returns THAT copy, so that
  // one RCCL and one HIP runtime serve the process; RTLD_LOCAL: a second copy must never interpose its symbols on the first.
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(std::string("RCCL not found (dlopen librccl.so.1): ") + dlerror());
#define SYM_(f) g_rccl.f = (decltype(g_rccl.f))dlsym(h, "nccl" #f); if (!g_rccl.f) return fail("librccl: symbol nccl" #f " missing")
  SYM_(GetUniqueId); SYM_(CommInitRank); SYM_(CommDestroy); SYM_(CommCount); SYM_(GroupStart); SYM_(GroupEnd); SYM_(Send); SYM_(Recv); SYM_(GetErrorString);
#undef SYM_
  g_rccl.h = h;
  return 0;
}
#define NCCLCHK(x)                                                                                   \
  do {                                                                                               \
    ncclResult_t r_ = (x);                                                                           \
    if (r_ != ncclSuccess) return fail(std::string(#x) + ": " + g_rccl.GetErrorString(r_));        \
  } while (0)

static void halo_release(ecwam_hip_ctx* c) {
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->d_send_idx) (void)hipFree(c->d_send_idx);
  for (auto& q : c->slot) {
    if (q.buf) (void)hipFree(q.buf);
    if (q.ev_packed) (void)hipEventDestroy(q.ev_packed);
    if (q.ev_done) (void)hipEventDestroy(q.ev_done);
    q = ecwam_hip_ctx::HaloSlot();
  }
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  c->comm = nullptr; c->d_send_idx = nullptr; c->comm_stream = nullptr;
}

// the first fast_gk frequencies of rows [kijs, kijl) of FL1 -> the compact rows of ecwam_hip_set_fastwave_copy (the kernels that do not
// write them from their tile)
static void fastwave_copy(ecwam_hip_ctx* c, const void* fl1, int kijs, int kijl, hipStream_t s) {
  const size_t rb = (size_t)c->real_bytes;
  const char* src = (const char*)fl1 + (size_t)kijs * c->NANG * c->NFRE * rb;
  char* dst = (char*)c->fast_g + (size_t)kijs * c->NANG * c->fast_gk * rb;
  if (c->real_bytes == 4) launch_copy_freq_range<float>(src, dst, kijl - kijs, c->NANG, c->NFRE, 0, c->fast_gk, c->fast_gk, s);
  else launch_copy_freq_range<double>(src, dst, kijl - kijs, c->NANG, c->NFRE, 0, c->fast_gk, c->fast_gk, s);
}

// ---- one member of a FIELD_API-shaped host type <-> its slot of the library's packed per-point rows ---------------------------------
// chunked member M(NPROMA[,NM],NCHNK) (Fortran order: element (ip, m, ich) at ((ich NM + m) NPROMA + ip)) and rows[ij][row_stride]
// with the member's NM values at row_off; E = the element as an integer of its size (reals, doubles and int32 move alike)
template <typename E>
__global__ void k_member_scatter(const E* __restrict__ src, E* __restrict__ rows, int nproma, int npts, int nm, long long row_stride, long long row_off) {
  const long long total = (long long)npts * nm;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ij = (int)(g / nm), m = (int)(g - (long long)ij * nm);
    const int ich = ij / nproma, ip = ij - ich * nproma;
    rows[(long long)ij * row_stride + row_off + m] = src[((size_t)ich * nm + m) * nproma + ip];
  }
}
template <typename E>
__global__ void k_member_gather(const E* __restrict__ rows, E* __restrict__ dst, int nproma, int nchnk, int npts, int nm, long long row_stride, long long row_off) {
  const long long total = (long long)nchnk * nproma * nm;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int ip = (int)(g % nproma);
    const long long r = g / nproma;
    const int m = (int)(r % nm), ich = (int)(r / nm);
    int ij = ich * nproma + ip;
    if (ij >= npts) ij = (nchnk - 1) * nproma;      // pad lanes of the ragged last chunk replicate its lane 1 (propag_wam.F90:388-398)
    dst[g] = rows[(long long)ij * row_stride + row_off + m];
  }
}

extern "C" {

const char* ecwam_hip_last_error(void) { return g_err.c_str(); }
int ecwam_hip_abi_version(void) { return ECWAM_HIP_ABI_VERSION; }

int ecwam_hip_selftest(int device) {
  HIPCHK(hipSetDevice(device));
  int* d = nullptr;
  int h[2] = {0, 0};
  HIPCHK(hipMalloc(&d, 2 * sizeof(int)));
  HIPCHK(hipMemset(d, 0, 2 * sizeof(int)));
  hipLaunchKernelGGL(k_selftest<float>, dim3(1), dim3(64), 0, 0, d);
  hipLaunchKernelGGL(k_selftest<double>, dim3(1), dim3(64), 0, 0, d + 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(h, d, 2 * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  if (h[0] || h[1]) return fail("ecwam_hip_selftest: a wavefront primitive returned a wrong value");
  return 0;
}

int ecwam_hip_create(const ecwam_hip_params* p, const ecwam_hip_tables* t, int real_bytes, int device, ecwam_hip_ctx** out) {
  if (!p || !t || !out) return fail("ecwam_hip_create: null argument");
  if (real_bytes != 4 && real_bytes != 8) return fail("ecwam_hip_create: real_bytes must be 4 or 8");
  if (p->nang < 4 || p->nang > MAXA || p->nfre < 8 || p->nfre > MAXF || p->nfre_red < 1 || p->nfre_red > p->nfre)
    return fail("ecwam_hip_create: NANG/NFRE/NFRE_RED out of the supported range");
  if (p->mlsthg > MAXMC || 2 * p->nsdsnth + 1 > MAXTAP || p->nwav_gc + 1 > MAXGC) return fail("ecwam_hip_create: table size exceeds library limits");
  if ((p->iphys != 0 && p->iphys != 1) || (p->isnonlin < 0 || p->isnonlin > 2) || p->irefra < 0 || p->irefra > 3 || p->icode < 1 || p->icode > 3)
    return fail("ecwam_hip_create: only IPHYS=0/1, ISNONLIN=0/1/2, IREFRA=0..3, ICODE=1..3 are on the hot path (SURVEY.md 8a)");
  if (p->lciwa1 && (!t->cideac || p->nict < 2 || p->nich < 2 || p->nict * p->nich > 36 * 16))
    return fail("ecwam_hip_create: LCIWA1 (SDICE1) needs the CIDEAC table (tables.cideac, NICT*NICH <= 576)");
  {   // every table the kernels read must be there (the optional ones: cideac, checked above)
    const void* need[] = {t->fr, t->dfim, t->dfimofr, t->dfimfr, t->dfim_sim, t->rhowg_dfim, t->zpifr, t->fr5, t->cofrm4, t->flmax, t->th, t->costh,
                          t->sinth, t->wtauhf, t->swellft, t->ikp, t->ikp1, t->ikm, t->ikm1, t->af11, t->k1w, t->k2w, t->k11w, t->k21w, t->inlcoef,
                          t->rnlcoef, t->indicessat, t->satweights, t->kpm, t->jxo, t->jyo, t->kcr};
    for (const void* q : need) if (!q) return fail("ecwam_hip_create: a required table pointer is NULL");
    if (p->nwav_gc > 0 && (!t->xk_gc || !t->xkm_gc || !t->omega_gc || !t->omxkm3_gc || !t->cm_gc || !t->c2osqrtvg_gc || !t->xkmsqrtvgoc2_gc ||
                           !t->om3gmkm_gc || !t->delkcc_gc_ns || !t->delkcc_omxkm3_gc)) return fail("ecwam_hip_create: a gravity-capillary table pointer is NULL");
  }
  HIPCHK(hipSetDevice(device));
  ecwam_hip_ctx* c = new ecwam_hip_ctx();
  c->real_bytes = real_bytes; c->device = device; c->NANG = p->nang; c->NFRE = p->nfre; c->NFRE_RED = p->nfre_red; c->p = *p;
#define HIPCHK_CTX(x)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) { if (c->dtab) (void)hipFree(c->dtab); delete c; return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } \
  } while (0)
  if (real_bytes == 4) {
    std::vector<DevTab<float>> h(1);
    build_tab<float>(p, t, h.data());
    if (!h[0].DIA_PULL) { delete c; return fail("ecwam_hip_create: the interaction tables do not have the rotation structure of INISNONLIN (K1W = K -+ r1, K2W = K +- r2, INLCOEF = MC, MC+2, MC+3, MC-4, MC-3; KFRH = 8, MFRSTLW = -3): not supported"); }
    v4_probe<float>(h[0], c);
    HIPCHK_CTX(hipMalloc(&c->dtab, sizeof(DevTab<float>)));
    HIPCHK_CTX(hipMemcpy(c->dtab, h.data(), sizeof(DevTab<float>), hipMemcpyHostToDevice));
  } else {
    std::vector<DevTab<double>> h(1);
    build_tab<double>(p, t, h.data());
    if (!h[0].DIA_PULL) { delete c; return fail("ecwam_hip_create: the interaction tables do not have the rotation structure of INISNONLIN (K1W = K -+ r1, K2W = K +- r2, INLCOEF = MC, MC+2, MC+3, MC-4, MC-3; KFRH = 8, MFRSTLW = -3): not supported"); }
    v4_probe<double>(h[0], c);
    HIPCHK_CTX(hipMalloc(&c->dtab, sizeof(DevTab<double>)));
    HIPCHK_CTX(hipMemcpy(c->dtab, h.data(), sizeof(DevTab<double>), hipMemcpyHostToDevice));
  }
#undef HIPCHK_CTX
  // IMPLSCH runs k_implsch4 (implsch_v4.h) and nothing else since round 5: refuse here what its builds do not cover, with the reason
  {
    const char* why = nullptr;
    const bool inst = (c->NANG == 48 && c->v4_r1 == 1 && c->v4_r2 == 4 && c->v4_nh == 11) || (c->NANG == 36 && c->v4_r1 == 1 && c->v4_r2 == 3 && c->v4_nh == 8) ||
                      (c->NANG == 24 && c->v4_r1 == 0 && c->v4_r2 == 2 && c->v4_nh == 5) || (c->NANG == 12 && c->v4_r1 == 0 && c->v4_r2 == 1 && c->v4_nh == 3);
    if (p->nfre != 36) why = "NFRE must be 36 (the frequency grid of every configuration of the reference)";
    else if (!c->v4_ok) why = "the saturation weights / indices do not have the structure of INIT_SDISS_ARDH (COS**2 taps, symmetric, the same for every direction) or the DIA mirror images differ";
    else if (!inst) why = "NANG must be 48, 36, 24 or 12 with the interaction rotations and the saturation half-width of INISNONLIN / INIT_SDISS_ARDH on that grid";
    else if (p->iphys == 1 && p->llnormagam && c->v4_shelter) why = "LLNORMAGAM needs TAUWSHELTER = 0 (setwavphys.F90:150-190: the normalised growth rate replaces the sheltering)";
    else if (p->iphys == 1 && !p->llnormagam && !c->v4_shelter) why = "IPHYS = 1 without LLNORMAGAM needs TAUWSHELTER /= 0 (setwavphys.F90:150-190)";
    // (the context is created all the same: advection, OUTBS, halo exchange and restart do not depend on IMPLSCH's builds, and the reference
    // accepts any NANG -- ecwam_hip_implsch / ecwam_hip_propags2_implsch then refuse with this reason)
    if (why) c->implsch_why = std::string("configuration not covered by the IMPLSCH kernel: ") + why;
  }
  *out = c;
  return 0;
}

int ecwam_hip_destroy(ecwam_hip_ctx* c) {
  if (!c) return 0;
  if (c->dtab) (void)hipFree(c->dtab);
  if (c->norm_scratch) (void)hipFree(c->norm_scratch);
  if (c->fin) (void)hipFree(c->fin);
  if (c->wi) (void)hipFree(c->wi);
  if (c->adv_pt) (void)hipFree(c->adv_pt);
  if (c->adv_dir) (void)hipFree(c->adv_dir);
  halo_release(c);
  delete c;
  return 0;
}

#define DISPATCH(call_f, call_d) \
  do {                           \
    if (c->real_bytes == 4) {    \
      call_f;                    \
    } else {                     \
      call_d;                    \
    }                            \
  } while (0)

int ecwam_hip_set_obstructions(ecwam_hip_ctx* c, const void* obs, int n) {
  if (!c) return fail("null context");
  if (n < 0 || (obs && n == 0)) return fail("ecwam_hip_set_obstructions: bad size");
  c->obs = obs; c->n_obs = obs ? n : 0;
  return 0;
}

int ecwam_hip_propags2(ecwam_hip_ctx* c, const void* f1, void* f3, const int* klon, const int* klat, const int* kcor, const void* w,
                       int kijs, int kijl, int nd3s, int nd3e, int copy_rest, void* stream) {
  if (!c) return fail("null context");
  if (kijl < kijs || nd3s < 1 || nd3e > c->NFRE_RED || nd3e < nd3s - 1) return fail("ecwam_hip_propags2: bad range");
  if (kijl > kijs && (!f1 || !f3 || !klon || !klat || !kcor || !w)) return fail("ecwam_hip_propags2: null pointer");
  if (f1 == f3) return fail("ecwam_hip_propags2: F1 and F3 must not alias");
  hipStream_t s = (hipStream_t)stream;
  const int N = (c->NANG << 16) | (c->NFRE << 8) | c->NFRE_RED;
  DISPATCH(launch_propags2<float>(c->dtab, f1, f3, klon, klat, kcor, w, kijs, kijl, nd3s - 1, nd3e, copy_rest, N, s),
           launch_propags2<double>(c->dtab, f1, f3, klon, klat, kcor, w, kijs, kijl, nd3s - 1, nd3e, copy_rest, N, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_ctuw(ecwam_hip_ctx* c, int n, int nland, int ngy, double delpro, int mstart, int mend, const int* kxlt, const void* zdello,
                   double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat, const int* kcor, void* wlat,
                   void* wcor, const void* cgroup_ext, const void* cosphm1_ext, void* w, int* cflfail, void* stream) {
  if (!c) return fail("null context");
  if (n < 0 || mstart < 1 || mend > c->NFRE_RED || mend < mstart) return fail("ecwam_hip_ctuw: bad range");
  if (n > 0 && (!kxlt || !zdello || !cosph || !sinph || !klon || !klat || !kcor || !wlat || !wcor || !cgroup_ext || !cosphm1_ext || !cflfail))
    return fail("ecwam_hip_ctuw: null pointer");
  if (c->obs && n > c->n_obs) return fail("ecwam_hip_ctuw: more points than the obstruction table holds");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_ctuw<float>(c->dtab, n, nland, ngy, delpro, mstart - 1, mend, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, cosphm1_ext, w, cflfail, c->NANG, c->obs, s),
           launch_ctuw<double>(c->dtab, n, nland, ngy, delpro, mstart - 1, mend, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, cosphm1_ext, w, cflfail, c->NANG, c->obs, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_propags2_otf_split(ecwam_hip_ctx* c, const void* f1, void* f3, int n, int ngy, double delpro, double delpro_lf, int ifrelfmax,
                                 int in_nfre, void* gout, int gout_nfre, const int* kxlt, const void* zdello, double xdella, const void* cosph, const void* sinph, const int* klon,
                                 const int* klat, const int* kcor, const void* wlat, const void* wcor, const void* cgroup_ext,
                                 const void* cosphm1_ext, const int* order, int kijs, int kijl, int nd3s, int nd3e, int copy_rest,
                                 void* stream) {
  return ecwam_hip_propags2_otf_fast(c, f1, f3, n, ngy, delpro, delpro_lf, ifrelfmax, in_nfre, nullptr, 0, 0, gout, gout_nfre, kxlt, zdello, xdella, cosph,
                                     sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, cosphm1_ext, order, kijs, kijl, nd3s, nd3e, copy_rest, stream);
}

int ecwam_hip_propags2_otf_fast(ecwam_hip_ctx* c, const void* f1, void* f3, int n, int ngy, double delpro, double delpro_lf, int ifrelfmax,
                                int in_nfre, const void* gin, int gin_nfre, int out_nfre, void* gout, int gout_nfre, const int* kxlt,
                                const void* zdello, double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat,
                                const int* kcor, const void* wlat, const void* wcor, const void* cgroup_ext, const void* cosphm1_ext,
                                const int* order, int kijs, int kijl, int nd3s, int nd3e, int copy_rest, void* stream) {
  if (!c) return fail("null context");
  if (out_nfre == 0) out_nfre = c->NFRE;
  if (out_nfre != c->NFRE && (out_nfre < nd3e || out_nfre > c->NFRE || gout))
    return fail("ecwam_hip_propags2_otf_fast: a compact output buffer must hold every advected frequency and excludes a second compact copy");
  if (gin && (in_nfre != 0 && in_nfre != c->NFRE)) return fail("ecwam_hip_propags2_otf_fast: the compact fast-wave input goes with full input rows");
  if (gin && (gin_nfre < 1 || gin_nfre > c->NFRE || gin == f3 || gin == gout)) return fail("ecwam_hip_propags2_otf_fast: bad compact input buffer");
  {
    // every compact row format is read and written with 16-byte accesses along the frequencies
    const int vec = 16 / c->real_bytes;
    if ((gin && (gin_nfre % vec != 0 || ((uintptr_t)gin % 16) != 0)) || (out_nfre != c->NFRE && (out_nfre % vec != 0 || ((uintptr_t)f3 % 16) != 0)) ||
        (in_nfre != 0 && in_nfre != c->NFRE && (in_nfre % vec != 0 || ((uintptr_t)f1 % 16) != 0)))
      return fail("ecwam_hip_propags2_otf_fast: compact rows must be 16-byte aligned and hold a multiple of 16 bytes per direction");
  }
  // with a processing order kijs..kijl index its entries (entries < 0 are padding and skipped: the order may be longer than n)
  if (kijl < kijs || kijs < 0 || (!order && kijl > n) || nd3s < 1 || nd3e > c->NFRE_RED || nd3e < nd3s - 1 || ifrelfmax < 0 || ifrelfmax > c->NFRE_RED)
    return fail("ecwam_hip_propags2_otf: bad range");
  if ((copy_rest & 4) && !order) return fail("ecwam_hip_propags2_otf: 2-D tiles need a processing order");
  if (in_nfre == 0) in_nfre = c->NFRE;
  if (gout && (gout_nfre < 1 || gout_nfre > c->NFRE || gout_nfre % (16 / c->real_bytes) != 0 || gout == f1 || ((uintptr_t)gout % 16) != 0))
    return fail("ecwam_hip_propags2_otf: the compact output buffer must be 16-byte aligned, distinct from F1, and hold a multiple of 16 bytes per direction");
  if (in_nfre != c->NFRE && (in_nfre < nd3e || in_nfre > c->NFRE || ((copy_rest & 1) && out_nfre != in_nfre)))
    return fail("ecwam_hip_propags2_otf: a compact input buffer must hold every advected frequency; copy_rest only into a compact output of the same width");
  if (kijl > kijs && (!f1 || !f3 || !kxlt || !zdello || !cosph || !sinph || !klon || !klat || !kcor || !wlat || !wcor || !cgroup_ext || !cosphm1_ext))
    return fail("ecwam_hip_propags2_otf: null pointer");
  if (f1 == f3) return fail("ecwam_hip_propags2_otf: F1 and F3 must not alias");
  if (c->obs && (order ? n : kijl) > c->n_obs) return fail("ecwam_hip_propags2_otf: more points than the obstruction table holds");
  hipStream_t s = (hipStream_t)stream;
  const int N = (c->NANG << 16) | (c->NFRE << 8) | c->NFRE_RED;
  copy_rest = (copy_rest & 1) | (copy_rest & 4);   // bit 0: carry the other frequencies over; bit 2: the order describes 2-D tiles
#ifdef ECWAM_HIP_DIAGNOSTICS
  { const char* e_ = getenv("ECWAM_HIP_OTF_WALK"); if (e_ && atoi(e_) == 0) copy_rest |= 2; }  // plain grid-stride tile walk
#endif
  DISPATCH(launch_propags2_otf<float>(c->dtab, f1, f3, n, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, cosphm1_ext, order, kijs, kijl, nd3s - 1, nd3e, copy_rest, N, c->obs, ifrelfmax, delpro_lf, in_nfre, gout, gout_nfre, gin, gin_nfre, out_nfre, s),
           launch_propags2_otf<double>(c->dtab, f1, f3, n, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, cosphm1_ext, order, kijs, kijl, nd3s - 1, nd3e, copy_rest, N, c->obs, ifrelfmax, delpro_lf, in_nfre, gout, gout_nfre, gin, gin_nfre, out_nfre, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_propags2_otf(ecwam_hip_ctx* c, const void* f1, void* f3, int n, int ngy, double delpro, const int* kxlt, const void* zdello,
                           double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat, const int* kcor,
                           const void* wlat, const void* wcor, const void* cgroup_ext, const void* cosphm1_ext, const int* order,
                           int kijs, int kijl, int nd3s, int nd3e, int copy_rest, void* stream) {
  return ecwam_hip_propags2_otf_split(c, f1, f3, n, ngy, delpro, delpro, 0, 0, nullptr, 0, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor,
                                      cgroup_ext, cosphm1_ext, order, kijs, kijl, nd3s, nd3e, copy_rest, stream);
}

int ecwam_hip_copy_freq_range(ecwam_hip_ctx* c, const void* src, void* dst, int n, int m_first, int m_last, int dst_nfre, void* stream) {
  if (!c) return fail("null context");
  if (dst_nfre == 0) dst_nfre = c->NFRE;
  if (n < 0 || m_first < 1 || m_last > c->NFRE || m_last < m_first - 1 || m_last > dst_nfre) return fail("ecwam_hip_copy_freq_range: bad range");
  if (n > 0 && (!src || !dst)) return fail("ecwam_hip_copy_freq_range: null pointer");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_copy_freq_range<float>(src, dst, n, c->NANG, c->NFRE, m_first - 1, m_last, dst_nfre, s),
           launch_copy_freq_range<double>(src, dst, n, c->NANG, c->NFRE, m_first - 1, m_last, dst_nfre, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_propdot(ecwam_hip_ctx* c, int n, int nland, const int* kxlt, const void* zdello, double xdella, const void* cosph,
                      const int* klon, const int* klat, const void* wlat, const void* cosphm1_ext, const void* depth_ext,
                      const void* u_ext, const void* v_ext, void* refr, void* stream) {
  if (!c) return fail("null context");
  if (n < 0) return fail("ecwam_hip_propdot: bad range");
  if (n > 0 && (!kxlt || !zdello || !cosph || !klon || !klat || !wlat || !cosphm1_ext || !depth_ext || !u_ext || !v_ext || !refr))
    return fail("ecwam_hip_propdot: null pointer");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_propdot<float>(c->dtab, n, nland, c->p.irefra, kxlt, zdello, xdella, cosph, klon, klat, wlat, cosphm1_ext, depth_ext, u_ext, v_ext, refr, s),
           launch_propdot<double>(c->dtab, n, nland, c->p.irefra, kxlt, zdello, xdella, cosph, klon, klat, wlat, cosphm1_ext, depth_ext, u_ext, v_ext, refr, s));
  HIPCHK(hipGetLastError());
  return 0;
}

// shared argument checks + launch of the general-IREFRA kernel (f1 == NULL: checks only)
static int propags2_gen_launch(ecwam_hip_ctx* c, const char* who, const void* f1, void* f3, int ngy, double delpro,
                               const int* kxlt, const void* zdello, double xdella, const void* cosph, const void* sinph,
                               const int* klon, const int* klat, const int* kcor, const void* wlat, const void* wcor,
                               const void* cgroup_ext, const void* omosnh2kd_ext, const void* wavnum_ext, const void* cosphm1_ext,
                               const void* refr, int* cflfail, int slot, int kijs, int kijl, int m0, int m1, int copy_rest,
                               hipStream_t s) {
  if (kijl > kijs && (!kxlt || !zdello || !cosph || !sinph || !klon || !klat || !kcor || !wlat || !wcor || !cgroup_ext || !omosnh2kd_ext ||
                      !wavnum_ext || !cosphm1_ext || !refr))
    return fail((std::string(who) + ": null pointer").c_str());
  const int N = (c->NANG << 16) | (c->NFRE << 8) | c->NFRE_RED;
  DISPATCH(launch_propags2_gen<float>(c->dtab, c->p.irefra, f1, f3, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, omosnh2kd_ext, wavnum_ext, cosphm1_ext, refr, cflfail, slot, kijs, kijl, m0, m1, copy_rest, N, c->obs, s),
           launch_propags2_gen<double>(c->dtab, c->p.irefra, f1, f3, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat, wcor, cgroup_ext, omosnh2kd_ext, wavnum_ext, cosphm1_ext, refr, cflfail, slot, kijs, kijl, m0, m1, copy_rest, N, c->obs, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_ctuw_refra(ecwam_hip_ctx* c, int n, int nland, int ngy, double delpro, int mstart, int mend, const int* kxlt,
                         const void* zdello, double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat,
                         const int* kcor, void* wlat, void* wcor, const void* cgroup_ext, const void* omosnh2kd_ext,
                         const void* wavnum_ext, const void* cosphm1_ext, void* refr, int llcflcuroff, int range, int* cflfail,
                         void* stream) {
  if (!c) return fail("null context");
  if (n < 0 || mstart < 1 || mend > c->NFRE_RED || mend < mstart || range < 0 || range > 1) return fail("ecwam_hip_ctuw_refra: bad range");
  if (n > 0 && (!wlat || !wcor || !cflfail || !klat || !kcor || !refr)) return fail("ecwam_hip_ctuw_refra: null pointer");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  // CTUWINI (ctuwupdt.F90:204-214; idempotent), then CTUWDRV for this frequency range (ctuwdrv.F90:93-118)
  DISPATCH(launch_ctuwini_only<float>(n, nland, klat, kcor, wlat, wcor, s), launch_ctuwini_only<double>(n, nland, klat, kcor, wlat, wcor, s));
  const bool cur = c->p.irefra == 2 || c->p.irefra == 3;
  std::vector<int> prev;  // flags of the first range survive the second range's own CTUWDRV
  if (range == 1) {
    prev.resize((size_t)n);
    HIPCHK(hipMemcpyAsync(prev.data(), cflfail, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  HIPCHK(hipMemsetAsync(cflfail, 0, sizeof(int) * (size_t)n, s));
  DISPATCH(launch_curmask<float>(n, c->NANG, range, nullptr, refr, s), launch_curmask<double>(n, c->NANG, range, nullptr, refr, s));  // CURMASK = 1
  int rc = propags2_gen_launch(c, "ecwam_hip_ctuw_refra", nullptr, nullptr, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor,
                               wlat, wcor, cgroup_ext, omosnh2kd_ext, wavnum_ext, cosphm1_ext, refr, cflfail, range, 0, n, mstart - 1, mend, 0, s);
  if (rc) return rc;
  if (cur && llcflcuroff) {
    // second CTUW call: current refraction / frequency shift switched off where the first one failed, flags reset
    std::vector<int> h((size_t)n);
    HIPCHK(hipMemcpyAsync(h.data(), cflfail, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    bool any = false;
    for (int i = 0; i < n && !any; i++) any = h[i] != 0;
    if (any) {
      DISPATCH(launch_curmask<float>(n, c->NANG, range, cflfail, refr, s), launch_curmask<double>(n, c->NANG, range, cflfail, refr, s));
      HIPCHK(hipMemsetAsync(cflfail, 0, sizeof(int) * (size_t)n, s));
      rc = propags2_gen_launch(c, "ecwam_hip_ctuw_refra", nullptr, nullptr, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat,
                               kcor, wlat, wcor, cgroup_ext, omosnh2kd_ext, wavnum_ext, cosphm1_ext, refr, cflfail, range, 0, n, mstart - 1, mend, 0, s);
      if (rc) return rc;
    }
  }
  if (range == 1) {
    std::vector<int> h((size_t)n);
    HIPCHK(hipMemcpyAsync(h.data(), cflfail, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int i = 0; i < n; i++) h[i] |= prev[i];
    HIPCHK(hipMemcpyAsync(cflfail, h.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  return 0;
}

int ecwam_hip_propags2_refra(ecwam_hip_ctx* c, const void* f1, void* f3, int n, int ngy, double delpro, const int* kxlt,
                             const void* zdello, double xdella, const void* cosph, const void* sinph, const int* klon,
                             const int* klat, const int* kcor, const void* wlat, const void* wcor, const void* cgroup_ext,
                             const void* omosnh2kd_ext, const void* wavnum_ext, const void* cosphm1_ext, const void* refr, int range,
                             int kijs, int kijl, int nd3s, int nd3e, int copy_rest, void* stream) {
  if (!c) return fail("null context");
  if (kijl < kijs || kijs < 0 || kijl > n || nd3s < 1 || nd3e > c->NFRE_RED || nd3e < nd3s - 1 || range < 0 || range > 1)
    return fail("ecwam_hip_propags2_refra: bad range");
  if (kijl > kijs && (!f1 || !f3)) return fail("ecwam_hip_propags2_refra: null pointer");
  if (f1 == f3) return fail("ecwam_hip_propags2_refra: F1 and F3 must not alias");
  if (c->obs && kijl > c->n_obs) return fail("ecwam_hip_propags2_refra: more points than the obstruction table holds");
  return propags2_gen_launch(c, "ecwam_hip_propags2_refra", f1, f3, ngy, delpro, kxlt, zdello, xdella, cosph, sinph, klon, klat, kcor, wlat,
                             wcor, cgroup_ext, omosnh2kd_ext, wavnum_ext, cosphm1_ext, refr, nullptr, range, kijs, kijl, nd3s - 1, nd3e,
                             copy_rest ? 1 : 0, (hipStream_t)stream);
}

// The tables the one-kernel step takes its CTU scalars from (k_ctu_prep fills them every call): sized with the fin rows, so that a host that
// called ecwam_hip_implsch_reserve has no allocation in its time loop on this path either.  Contexts without a one-kernel build hold none.
static bool fused_ok(const ecwam_hip_ctx* c);
static int adv_reserve(ecwam_hip_ctx* c, int npts) {
  if (!fused_ok(c) || !implsch4_adv_forms(c->NANG, c->real_bytes)) return 0;
  const size_t need = (size_t)(npts > 0 ? npts : 0) * 12 * c->real_bytes;
  if (need > c->adv_pt_bytes) {   // hipFree waits for the kernels still reading the old table
    if (c->adv_pt) HIPCHK(hipFree(c->adv_pt));
    c->adv_pt = nullptr; c->adv_pt_bytes = 0;
    HIPCHK(hipMalloc(&c->adv_pt, need));
    c->adv_pt_bytes = need;
  }
  if (!c->adv_dir) HIPCHK(hipMalloc(&c->adv_dir, (size_t)(6 * c->NANG + 4) * c->real_bytes + (size_t)4 * c->NANG * sizeof(int)));
  return 0;
}
// the configurations that run the RARE builds of k_implsch4 (ecwam_hip_implsch below makes the same choice)
static bool runs_rare_builds(const ecwam_hip_ctx* c) {
  const int ext = (c->p.llnormagam || c->p.llgcbz0) ? 1 : 0;
  const bool rare4 = c->p.lciwa2 || c->p.lwnemocouwrs || c->p.lwnemocoustrn || c->p.isnonlin > 1 || c->p.icode != 3 || !c->p.lwvflx_snl;
  const int alt = (c->p.iphys == 0 ? 1 : 0) | (c->p.isnonlin == 1 ? 2 : 0);
  return !(!rare4 && (alt == 0 || (!ext && alt != 3)));
}
static int implsch_reserve_on(ecwam_hip_ctx* c, int npts, hipStream_t s, bool sync);
int ecwam_hip_implsch(ecwam_hip_ctx* c, int kijs, int kijl, void* fl1, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws,
                      double* wam2nemo, void* dbg, void* stream) {
  if (!c) return fail("null context");
  if (kijl < kijs || kijs < 0) return fail("ecwam_hip_implsch: bad range");
  HIPCHK(hipSetDevice(c->device));
  if (kijl > kijs && (!fl1 || !wvprpt || !ff || !intf || !mij || !xllws)) return fail("ecwam_hip_implsch: null pointer");
  if (kijl > kijs && c->p.lwnemocou && !wam2nemo) return fail("ecwam_hip_implsch: LWNEMOCOU needs the WAVE2OCEAN buffer");
  if (!c->p.lwnemocou) wam2nemo = nullptr;
  hipStream_t s = (hipStream_t)stream;
  int rc = -1;
  if (!c->implsch_why.empty()) return fail("ecwam_hip_implsch: " + c->implsch_why);
  if (dbg) return fail("ecwam_hip_implsch: the per-point debug rows were an output of the retired one-point-per-wavefront kernel: pass NULL");
  // k_implsch4 (implsch_v4.h).  The common builds: flag sets A and B (LLGCBZ0, LLNORMAGAM), with or without the sea-ice
  // damping rates that depend on the frequency only (LCIWA1, LCIWA3, LCISCAL), and on flag set A IPHYS 0 or ISNONLIN 1 -- what the
  // reference's registered configurations select.  Everything else runs its RARE builds (implsch4r.hip: LCIWA2, the NEMO ice stress and
  // strain, ISNONLIN 2, ICODE 1 / 2, LWVFLX_SNL = F, ISNONLIN 1 beside flag set B or IPHYS 0, IPHYS 0 beside flag set B).  ecwam_hip_create
  // has refused what no build covers.
  const int ext = (c->p.llnormagam || c->p.llgcbz0) ? 1 : 0;
  const bool rare4 = c->p.lciwa2 || c->p.lwnemocouwrs || c->p.lwnemocoustrn || c->p.isnonlin > 1 || c->p.icode != 3 || !c->p.lwvflx_snl;
  // the alternate physics the registered configurations select, on flag set A only: 1 = IPHYS 0 (sinput_jan + sdissip_jan; its
  // TAUWSHELTER is 0), 2 = ISNONLIN 1 (TRANSF per interaction frequency)
  const int alt = (c->p.iphys == 0 ? 1 : 0) | (c->p.isnonlin == 1 ? 2 : 0);
  const bool common_ok = !rare4 && (alt == 0 || (!ext && alt != 3));      // (= !runs_rare_builds(c))
  {
    if (int rc2 = implsch_reserve_on(c, kijl, s, false)) return rc2;   // no-op once the buffer covers kijl
    if (!common_ok)
      DISPATCH(rc = launch_implsch4r<float>(c->dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, c->fast_g, c->fast_gk, c->wi, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, c->p.iphys == 0 ? 1 : 0, s),
               rc = launch_implsch4r<double>(c->dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, c->fast_g, c->fast_gk, c->wi, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, c->p.iphys == 0 ? 1 : 0, s));
    else if (alt)
      DISPATCH(rc = launch_implsch4x<float>(c->dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, c->fast_g, c->fast_gk, c->wi, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, alt, s),
               rc = launch_implsch4x<double>(c->dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, c->fast_g, c->fast_gk, c->wi, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, alt, s));
    else
      DISPATCH(rc = launch_implsch4<float>(c->dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, c->fast_g, c->fast_gk, c->wi, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, ext, s),
               rc = launch_implsch4<double>(c->dtab, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, c->fast_g, c->fast_gk, c->wi, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, ext, s));
    if (rc == 0) { HIPCHK(hipGetLastError()); c->implsch_last = 4; return 0; }
  }
  return fail("ecwam_hip_implsch: no build of k_implsch4 covers the configuration (ecwam_hip_create should have refused it)");
}

// the configurations the one-kernel step covers (implsch4a.hip); everything else runs ecwam_hip_propags2_otf + ecwam_hip_implsch
static bool fused_ok(const ecwam_hip_ctx* c) {      // (declared above adv_reserve)
  if (!c->implsch_why.empty()) return false;
  const bool rare4 = c->p.lciwa2 || c->p.lwnemocouwrs || c->p.lwnemocoustrn || c->p.isnonlin > 1 || c->p.icode != 3 || !c->p.lwvflx_snl;
  return !rare4 && c->p.iphys == 1 && c->p.isnonlin == 0 && c->NFRE == 36;      // (the direction count: one of k_implsch4's, checked at create)
}
// bit 0: the one-kernel step covers the context; bit 1: also with fast-wave sub-steps (gin); bit 2: also with the obstructions of
// ecwam_hip_set_obstructions -- a caller takes the one-kernel step when the bits of what it needs are set
int ecwam_hip_propags2_implsch_supported(ecwam_hip_ctx* c) { return c && fused_ok(c) ? implsch4_adv_forms(c->NANG, c->real_bytes) : 0; }

int ecwam_hip_propags2_implsch(ecwam_hip_ctx* c, const void* f1, void* f3, int n, int ngy, double delpro, const int* kxlt, const void* zdello,
                               double xdella, const void* cosph, const void* sinph, const int* klon, const int* klat, const int* kcor,
                               const void* wlat, const void* wcor, const void* cgroup_ext, const void* cosphm1_ext, int kijs, int kijl, int nd3s,
                               int nd3e, const void* wvprpt, void* ff, void* intf, int* mij, void* xllws, double* wam2nemo, double delpro_lf,
                               int ifrelfmax, const void* gin, int gin_nfre, int flags, void* stream) {
  if (!c) return fail("null context");
  if (!fused_ok(c) || !implsch4_adv_forms(c->NANG, c->real_bytes)) return fail("ecwam_hip_propags2_implsch: no one-kernel build covers the configuration (ecwam_hip_propags2_implsch_supported): call ecwam_hip_propags2_otf and ecwam_hip_implsch");
  if (kijl < kijs || kijs < 0 || kijl > n || nd3s < 1 || nd3e > c->NFRE_RED || nd3e < nd3s - 1) return fail("ecwam_hip_propags2_implsch: bad range");
  if (kijl > kijs && (!f1 || !f3 || !kxlt || !zdello || !cosph || !sinph || !klon || !klat || !kcor || !wlat || !wcor || !cgroup_ext || !cosphm1_ext ||
                      !wvprpt || !ff || !intf || !mij || !xllws))
    return fail("ecwam_hip_propags2_implsch: null pointer");
  if (f1 == f3) return fail("ecwam_hip_propags2_implsch: F1 and F3 must not alias (the neighbours of a point are read while other points are stored)");
  if (((uintptr_t)f1 % 16) != 0 || ((uintptr_t)f3 % 16) != 0) return fail("ecwam_hip_propags2_implsch: the spectra must be 16-byte aligned");
  // fast waves (propag_wam.F90:247-313): frequencies 1..ifrelfmax advance with delpro_lf and are read from the compact rows their sub-steps left
  if (ifrelfmax < 0 || ifrelfmax > c->NFRE_RED || (ifrelfmax > 0) != (gin != nullptr))
    return fail("ecwam_hip_propags2_implsch: fast waves (ifrelfmax > 0) come with their compact input rows (gin), and only then");
  if (gin && (nd3s != 1 || gin_nfre < ifrelfmax || gin_nfre > c->NFRE || gin_nfre % (16 / c->real_bytes) != 0 || ((uintptr_t)gin % 16) != 0 || gin == c->fast_g ||
              !(delpro_lf > 0.0)))
    return fail("ecwam_hip_propags2_implsch: the compact fast-wave rows must be 16-byte aligned, hold the fast waves in a multiple of 16 bytes per direction, and differ from the rows of ecwam_hip_set_fastwave_copy");
  if (c->p.lwnemocou && kijl > kijs && !wam2nemo) return fail("ecwam_hip_propags2_implsch: LWNEMOCOU needs the WAVE2OCEAN buffer");
  if (c->obs && kijl > c->n_obs) return fail("ecwam_hip_propags2_implsch: more points than the obstruction table holds");
  if (!c->p.lwnemocou) wam2nemo = nullptr;
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s = (hipStream_t)stream;
  if (kijl == kijs) return 0;
  if (int rc2 = implsch_reserve_on(c, kijl, s, false)) return rc2;      // (also the tables of the weights: adv_reserve)
  const size_t dir_reals = (size_t)(6 * c->NANG + 4);      // [NANG][4], CMTODEG (+ 3 pad), [NANG][2] for the fast waves' time step
  int* dirI = reinterpret_cast<int*>(reinterpret_cast<char*>(c->adv_dir) + dir_reals * c->real_bytes);
  const double dlf = gin ? delpro_lf : 0.0;
  DISPATCH(launch_ctu_prep<float>(c->dtab, kijs, kijl, ngy, delpro, dlf, kxlt, zdello, xdella, cosph, sinph, wlat, wcor, cosphm1_ext, c->adv_pt, c->adv_dir, dirI, s),
           launch_ctu_prep<double>(c->dtab, kijs, kijl, ngy, delpro, dlf, kxlt, zdello, xdella, cosph, sinph, wlat, wcor, cosphm1_ext, c->adv_pt, c->adv_dir, dirI, s));
  Implsch4AdvArgs a;
  a.f_in = f1; a.klon = klon; a.klat = klat; a.kcor = kcor; a.cg = cgroup_ext; a.pt = c->adv_pt; a.dirT = c->adv_dir; a.dirI = dirI;
  a.xdella = xdella; a.delpro = delpro; a.m0 = nd3s - 1; a.m1 = nd3e;
  a.gin = gin; a.delpro_lf = dlf; a.gin_k = gin ? gin_nfre : 0; a.mlf = gin ? ifrelfmax : 0;
  a.obs = c->obs;      // ecwam_hip_set_obstructions (LSUBGRID)
  a.gfast = c->fast_g; a.gfast_k = c->fast_g ? c->fast_gk : 0;      // ecwam_hip_set_fastwave_copy, as for ecwam_hip_implsch
  // flags bit 0: the workgroups in the XCD-aware order of k_propags2_otf (diagnostics: 1 % slower here); bits 8..19: G > 1 = groups of G
  // consecutive waves per XCD (diagnostics)
  a.xcd_walk = (flags & 1) ? 1 : (((flags >> 8) & 0xFFF) > 1 ? ((flags >> 8) & 0xFFF) : 0);
  a.mode = (flags & 2) ? 2 : 1;          // flags bit 1: the go / no-go probe (libraries built with -DV4_ADV_PROBE only)
  const int ext = (c->p.llnormagam || c->p.llgcbz0) ? 1 : 0;
  int rc = -1;
  DISPATCH(rc = launch_implsch4_adv<float>(c->dtab, kijs, kijl, f3, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, &a, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, ext, s),
           rc = launch_implsch4_adv<double>(c->dtab, kijs, kijl, f3, wvprpt, ff, intf, mij, xllws, c->fin, wam2nemo, &a, c->NANG, c->NFRE, c->v4_r1, c->v4_r2, c->v4_nh, ext, s));
  if (rc != 0) return fail("ecwam_hip_propags2_implsch: no one-kernel build covers the configuration (the strict build of the weights has neither the fast-wave nor the obstruction form)");
  HIPCHK(hipGetLastError());
  c->implsch_last = 4;
  return 0;
}

int ecwam_hip_set_fastwave_copy(ecwam_hip_ctx* c, void* g, int g_nfre) {
  if (!c) return fail("null context");
  if (g && (g_nfre < 1 || g_nfre > c->NFRE || g_nfre % (16 / c->real_bytes) != 0 || ((uintptr_t)g % 16) != 0))
    return fail("ecwam_hip_set_fastwave_copy: the compact rows must be 16-byte aligned and hold a multiple of 16 bytes per direction");
  c->fast_g = g; c->fast_gk = g ? g_nfre : 0;
  return 0;
}

// grows the context's per-point buffers to npts points.  `s`: the stream of the IMPLSCH call that needs them (the rows are zeroed ON that
// stream: a hipMemset on the null stream is not ordered against a non-blocking stream -- the caller's kernels could write their rows
// before the zeroes arrive; found by test_implsch_in_blocks_is_bit_identical in a long test process), or sync = true for the set-up call
static int implsch_reserve_on(ecwam_hip_ctx* c, int npts, hipStream_t s, bool sync) {
  HIPCHK(hipSetDevice(c->device));
  const size_t need = (size_t)(npts > 0 ? npts : 0) * implsch4_fin_row() * c->real_bytes;
  if (need > c->fin_bytes) {   // hipFree waits for the kernels still reading the old rows
    if (c->fin) HIPCHK(hipFree(c->fin));
    c->fin = nullptr; c->fin_bytes = 0;
    HIPCHK(hipMalloc(&c->fin, need));
    HIPCHK(hipMemsetAsync(c->fin, 0, need, s));
    // growth is a stall anyway (hipFree / hipMalloc): wait for the zeroes, so that a call on ANOTHER stream issued right after this one
    // (which needs no growth itself) cannot have its rows overwritten by a memset still pending here
    HIPCHK(hipStreamSynchronize(s));
    (void)sync;
    c->fin_bytes = need;
  }
  // the split kernel pair parks the wind-input coefficient of every bin between its two halves: the whole library built as the split (build
  // variant), or a double precision context whose configuration runs the RARE builds (implsch4r.hip: their dp form is the split) -- NOT
  // every double precision context: the rows are a fourth spectrum-sized array (68 GB at O1280)
  const bool split = implsch4_split_all() || (c->real_bytes == 8 && implsch4r_dp_split() && runs_rare_builds(c));
  const size_t need_wi = split ? (size_t)(npts > 0 ? npts : 0) * c->NANG * c->NFRE * c->real_bytes : 0;
  if (need_wi > c->wi_bytes) {
    if (c->wi) HIPCHK(hipFree(c->wi));
    c->wi = nullptr; c->wi_bytes = 0;
    HIPCHK(hipMalloc(&c->wi, need_wi));
    c->wi_bytes = need_wi;
  }
  return adv_reserve(c, npts);
}

int ecwam_hip_implsch_reserve(ecwam_hip_ctx* c, int npts) {
  if (!c) return fail("ecwam_hip_implsch_reserve: null context");
  return implsch_reserve_on(c, npts, nullptr, true);
}

int ecwam_hip_implsch_generation_used(ecwam_hip_ctx* c) { return c ? c->implsch_last : 0; }

// The device copy of the module tables (DevTab<float> or DevTab<double>, csrc/dev.h): for diagnostics and for second implementations of a
// kernel that want to run on exactly the tables the product runs on (tests/csrc/implsch_v2.hip).
const void* ecwam_hip_device_tables(ecwam_hip_ctx* c) { return c ? c->dtab : nullptr; }

int ecwam_hip_outbs(ecwam_hip_ctx* c, int kijs, int kijl, const void* fl1, double zmiss, void* out, void* stream) {
  if (!c) return fail("null context");
  if (kijl < kijs || kijs < 0) return fail("ecwam_hip_outbs: bad range");
  if (kijl > kijs && (!fl1 || !out)) return fail("ecwam_hip_outbs: null pointer");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  DISPATCH(rc = launch_outbs<float>(c->dtab, kijs, kijl, fl1, zmiss, out, c->NANG, c->NFRE, s),
           rc = launch_outbs<double>(c->dtab, kijs, kijl, fl1, zmiss, out, c->NANG, c->NFRE, s));
  if (rc) return fail("ecwam_hip_outbs: unsupported spectral size");
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_outwnorm(ecwam_hip_ctx* c, const void* field, int stride, int n, double zmiss, double* result, void* stream) {
  if (!c) return fail("null context");
  if (n < 0 || stride < 1 || !result || (n > 0 && !field)) return fail("ecwam_hip_outwnorm: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int nb = 256;
  HIPCHK(hipSetDevice(c->device));
  if (!c->norm_scratch) HIPCHK(hipMalloc(&c->norm_scratch, (size_t)(4 + 4 * nb) * sizeof(double)));
  double* scratch = c->norm_scratch;
  DISPATCH(launch_norm<float>(field, stride, n, zmiss, scratch, nb, s), launch_norm<double>(field, stride, n, zmiss, scratch, nb, s));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(result, scratch, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecwam_hip_newwind_icode(ecwam_hip_ctx* c, int n, void* ff, const void* ff_next, int icode_wnd, void* stream) {
  if (!c) return fail("null context");
  if (n > 0 && (!ff || !ff_next)) return fail("ecwam_hip_newwind: null pointer");
  if (icode_wnd < 1 || icode_wnd > 3) return fail("ecwam_hip_newwind: ICODE_WND must be 1, 2 or 3");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_newwind<float>(c->dtab, n, ff, ff_next, icode_wnd, s), launch_newwind<double>(c->dtab, n, ff, ff_next, icode_wnd, s));
  HIPCHK(hipGetLastError());
  return 0;
}
int ecwam_hip_newwind(ecwam_hip_ctx* c, int n, void* ff, const void* ff_next, void* stream) {
  if (!c) return fail("null context");
  return ecwam_hip_newwind_icode(c, n, ff, ff_next, c->p.icode, stream);
}

int ecwam_hip_nosource(ecwam_hip_ctx* c, int kijs, int kijl, void* fl1, int* mij, void* xllws, void* stream) {
  if (!c) return fail("null context");
  if (kijs < 0 || kijl < kijs) return fail("ecwam_hip_nosource: bad range");
  if (kijl > kijs && (!mij || !xllws)) return fail("ecwam_hip_nosource: null pointer");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_nosource<float>(c->dtab, kijs, kijl, c->NANG * c->NFRE, fl1, xllws, mij, s),
           launch_nosource<double>(c->dtab, kijs, kijl, c->NANG * c->NFRE, fl1, xllws, mij, s));
  HIPCHK(hipGetLastError());
  if (c->fast_g && fl1 && kijl > kijs) { fastwave_copy(c, fl1, kijs, kijl, s); HIPCHK(hipGetLastError()); }
  return 0;
}

int ecwam_hip_host_register(ecwam_hip_ctx* c, void* host, unsigned long long bytes) {
  if (!c || !host) return fail("ecwam_hip_host_register: null argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipHostRegister(host, bytes, hipHostRegisterDefault));
  return 0;
}
int ecwam_hip_host_unregister(ecwam_hip_ctx* c, void* host) {
  if (!c || !host) return fail("ecwam_hip_host_unregister: null argument");
  HIPCHK(hipHostUnregister(host));
  return 0;
}

int ecwam_hip_chunks_to_points(ecwam_hip_ctx* c, const void* chunked, void* points, int nproma, int nchnk, int npts, int n2, int n3, void* stream) {
  if (!c) return fail("null context");
  if (nproma < 1 || nchnk < 0 || npts > nproma * nchnk || n2 < 1 || n3 < 1) return fail("ecwam_hip_chunks_to_points: bad shape");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_c2p<float>(chunked, points, nproma, nchnk, npts, n2, n3, s), launch_c2p<double>(chunked, points, nproma, nchnk, npts, n2, n3, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_points_to_chunks(ecwam_hip_ctx* c, const void* points, void* chunked, int nproma, int nchnk, int npts, int n2, int n3, void* stream) {
  if (!c) return fail("null context");
  if (nproma < 1 || nchnk < 0 || npts > nproma * nchnk || npts <= nproma * (nchnk - 1) || n2 < 1 || n3 < 1) return fail("ecwam_hip_points_to_chunks: bad shape");
  hipStream_t s = (hipStream_t)stream;
  DISPATCH(launch_p2c<float>(points, chunked, nproma, nchnk, npts, n2, n3, s), launch_p2c<double>(points, chunked, nproma, nchnk, npts, n2, n3, s));
  HIPCHK(hipGetLastError());
  return 0;
}

static int member_args(const char* who, int nproma, int nchnk, int npts, int nm, long long row_stride, long long row_off, int elem_bytes,
                       const void* a, const void* b) {
  if (nproma < 1 || nchnk < 0 || npts < 0 || npts > nproma * nchnk || (npts > 0 && npts <= nproma * (nchnk - 1)) || nm < 1 || row_off < 0 ||
      row_off + nm > row_stride || (elem_bytes != 4 && elem_bytes != 8))
    return fail(std::string(who) + ": bad shape");
  if (npts > 0 && (!a || !b)) return fail(std::string(who) + ": null pointer");
  return 0;
}

int ecwam_hip_member_scatter(ecwam_hip_ctx* c, const void* chunked, void* rows, int nproma, int nchnk, int npts, int nm, long long row_stride,
                             long long row_off, int elem_bytes, void* stream) {
  if (!c) return fail("null context");
  if (member_args("ecwam_hip_member_scatter", nproma, nchnk, npts, nm, row_stride, row_off, elem_bytes, chunked, rows)) return 1;
  if (npts == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)npts * nm;
  const int nb = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_member_scatter<unsigned int>, dim3(nb), dim3(256), 0, s, (const unsigned int*)chunked, (unsigned int*)rows, nproma, npts, nm, row_stride, row_off);
  else
    hipLaunchKernelGGL(k_member_scatter<unsigned long long>, dim3(nb), dim3(256), 0, s, (const unsigned long long*)chunked, (unsigned long long*)rows, nproma, npts, nm, row_stride, row_off);
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_member_gather(ecwam_hip_ctx* c, const void* rows, void* chunked, int nproma, int nchnk, int npts, int nm, long long row_stride,
                            long long row_off, int elem_bytes, void* stream) {
  if (!c) return fail("null context");
  if (member_args("ecwam_hip_member_gather", nproma, nchnk, npts, nm, row_stride, row_off, elem_bytes, rows, chunked)) return 1;
  if (npts == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const long long total = (long long)nchnk * nproma * nm;
  const int nb = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
  if (elem_bytes == 4)
    hipLaunchKernelGGL(k_member_gather<unsigned int>, dim3(nb), dim3(256), 0, s, (const unsigned int*)rows, (unsigned int*)chunked, nproma, nchnk, npts, nm, row_stride, row_off);
  else
    hipLaunchKernelGGL(k_member_gather<unsigned long long>, dim3(nb), dim3(256), 0, s, (const unsigned long long*)rows, (unsigned long long*)chunked, nproma, nchnk, npts, nm, row_stride, row_off);
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_pack_rows(ecwam_hip_ctx* c, const void* fl, const int* idx, int n, void* buf, void* stream) {
  if (!c) return fail("null context");
  if (n > 0 && (!fl || !idx || !buf)) return fail("ecwam_hip_pack_rows: null pointer");
  hipStream_t s = (hipStream_t)stream;
  const int rowlen = c->NANG * c->NFRE;
  DISPATCH(launch_pack<float>(fl, idx, n, rowlen, buf, s), launch_pack<double>(fl, idx, n, rowlen, buf, s));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_unpack_rows(ecwam_hip_ctx* c, const void* buf, int n, void* fl, int dst0, void* stream) {
  if (!c) return fail("null context");
  if (n > 0 && (!fl || !buf)) return fail("ecwam_hip_unpack_rows: null pointer");
  if (n <= 0) return 0;
  const size_t row = (size_t)c->NANG * c->NFRE * c->real_bytes;
  HIPCHK(hipMemcpyAsync((char*)fl + (size_t)dst0 * row, buf, (size_t)n * row, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// ---- MPEXCHNG inside the library (mpexchng.F90:141-231) -----------------------------------------------------------------------------
int ecwam_hip_halo_setup(ecwam_hip_ctx* c, int rank, int nranks, int npeers, const int* peer, const int* send_count, const int* send_idx,
                         const int* recv_dst0, const int* recv_count) {
  if (!c) return fail("null context");
  if (nranks < 1 || rank < 0 || rank >= nranks || npeers < 0 || (npeers > 0 && (!peer || !send_count || !recv_dst0 || !recv_count)))
    return fail("ecwam_hip_halo_setup: bad arguments");
  HIPCHK(hipSetDevice(c->device));
  if (c->d_send_idx) { (void)hipFree(c->d_send_idx); c->d_send_idx = nullptr; }
  c->rank = rank; c->nranks = nranks;
  c->peer.assign(peer, peer + npeers); c->send_cnt.assign(send_count, send_count + npeers);
  c->recv_dst0.assign(recv_dst0, recv_dst0 + npeers); c->recv_cnt.assign(recv_count, recv_count + npeers);
  c->send_off.resize(npeers);
  c->n_send = 0; c->n_recv = 0;
  for (int i = 0; i < npeers; i++) {
    // (peer[i] == rank is accepted: an exchange of a rank with itself -- rows that are their owner's own halo, e.g. a band that wraps
    //  around; RCCL pairs a send and a receive of the same rank inside one group)
    if (peer[i] < 0 || peer[i] >= nranks || send_count[i] < 0 || recv_count[i] < 0 || recv_dst0[i] < 0)
      return fail("ecwam_hip_halo_setup: bad peer entry");
    c->send_off[i] = c->n_send; c->n_send += send_count[i]; c->n_recv += recv_count[i];
  }
  if (c->n_send > 0) {
    if (!send_idx) return fail("ecwam_hip_halo_setup: send_idx missing");
    HIPCHK(hipMalloc(&c->d_send_idx, (size_t)c->n_send * sizeof(int)));
    HIPCHK(hipMemcpy(c->d_send_idx, send_idx, (size_t)c->n_send * sizeof(int), hipMemcpyHostToDevice));
  }
  if (!c->comm_stream) HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  return 0;
}

int ecwam_hip_proenvhalo_pack(ecwam_hip_ctx* c, int n, const void* wvprpt, const void* omosnh2kd, const void* depth, const void* ucur, const void* vcur,
                              void* buffer_ext, void* stream) {
  if (!c) return fail("null context");
  if (n < 0 || (n > 0 && (!wvprpt || !omosnh2kd || !depth || !ucur || !vcur || !buffer_ext))) return fail("ecwam_hip_proenvhalo_pack: bad arguments");
  DISPATCH(launch_proenv_pack<float>(n, c->NFRE, wvprpt, omosnh2kd, depth, ucur, vcur, buffer_ext, (hipStream_t)stream),
           launch_proenv_pack<double>(n, c->NFRE, wvprpt, omosnh2kd, depth, ucur, vcur, buffer_ext, (hipStream_t)stream));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_proenvhalo_unpack(ecwam_hip_ctx* c, int nrows, const void* buffer_ext, const void* land, void* wavnum_ext, void* cgroup_ext,
                                void* omosnh2kd_ext, void* depth_ext, void* u_ext, void* v_ext, void* stream) {
  if (!c) return fail("null context");
  if (nrows < 0 || !buffer_ext || !land || !wavnum_ext || !cgroup_ext || !omosnh2kd_ext || !depth_ext || !u_ext || !v_ext)
    return fail("ecwam_hip_proenvhalo_unpack: bad arguments");
  DISPATCH(launch_proenv_unpack<float>(nrows, c->NFRE, buffer_ext, land, wavnum_ext, cgroup_ext, omosnh2kd_ext, depth_ext, u_ext, v_ext, (hipStream_t)stream),
           launch_proenv_unpack<double>(nrows, c->NFRE, buffer_ext, land, wavnum_ext, cgroup_ext, omosnh2kd_ext, depth_ext, u_ext, v_ext, (hipStream_t)stream));
  HIPCHK(hipGetLastError());
  return 0;
}

int ecwam_hip_halo_counts(ecwam_hip_ctx* c, int* n_send, int* n_recv) {
  if (!c || !n_send || !n_recv) return fail("ecwam_hip_halo_counts: null argument");
  *n_send = c->n_send; *n_recv = c->n_recv;
  return 0;
}

int ecwam_hip_comm_unique_id(void* id128) {
  if (!id128) return fail("ecwam_hip_comm_unique_id: null argument");
  if (rccl_load()) return 1;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId");
  NCCLCHK(g_rccl.GetUniqueId((ncclUniqueId*)id128));
  return 0;
}

int ecwam_hip_comm_init(ecwam_hip_ctx* c, const void* id128) {
  if (!c || !id128) return fail("ecwam_hip_comm_init: null argument");
  // ncclCommInitRank is collective over all nranks: a rank of a multi-rank run joins even when it has no halo peer (a band that is an
  // enclosed basin), or the others would wait for it; only the single rank without a self exchange needs no communicator
  if (c->nranks < 2 && c->peer.empty()) return 0;
  if (rccl_load()) return 1;
  HIPCHK(hipSetDevice(c->device));
  if (c->comm) {   // a second initialisation replaces the communicator: let the posted exchange finish, then release the old one
    if (c->comm_stream) HIPCHK(hipStreamSynchronize(c->comm_stream));
    (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
    for (auto& q : c->slot) q.used = q.pending = false;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  NCCLCHK(g_rccl.CommInitRank(&c->comm, c->nranks, id, c->rank));
  return 0;
}

int ecwam_hip_comm_count(ecwam_hip_ctx* c, int* nranks) {
  if (!c || !nranks) return fail("ecwam_hip_comm_count: null argument");
  *nranks = 0;
  if (!c->comm) return 0;
  NCCLCHK(g_rccl.CommCount(c->comm, nranks));
  return 0;
}

// the send buffer for rows of `rowlen` reals: the slot already serving that length, else a free one, else the least recently used
// (whose outstanding sends the caller then waits for, as for any reuse)
static ecwam_hip_ctx::HaloSlot* halo_slot(ecwam_hip_ctx* c, int rowlen) {
  ecwam_hip_ctx::HaloSlot* q = nullptr;
  for (auto& x : c->slot) if (x.rowlen == rowlen) { q = &x; break; }
  if (!q) for (auto& x : c->slot) if (x.rowlen == 0) { q = &x; break; }
  if (!q) { q = &c->slot[0]; for (auto& x : c->slot) if (x.age < q->age) q = &x; }
  q->rowlen = rowlen; q->age = ++c->slot_clock;
  return q;
}

static int halo_pack(ecwam_hip_ctx* c, ecwam_hip_ctx::HaloSlot* q, const void* fl, int rowlen, hipStream_t s) {
  const size_t need = (size_t)c->n_send * rowlen * c->real_bytes;
  if (!q->ev_packed) {
    HIPCHK(hipEventCreateWithFlags(&q->ev_packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&q->ev_done, hipEventDisableTiming));
  }
  // the buffer is reused: the sends of the exchange posted from it before must have read it before this pack overwrites it
  if (q->used) HIPCHK(hipStreamWaitEvent(s, q->ev_done, 0));
  if (need > q->bytes) {
    if (q->buf) HIPCHK(hipFree(q->buf));      // (hipFree waits for the device)
    q->buf = nullptr; q->bytes = 0;
    HIPCHK(hipMalloc(&q->buf, need));
    q->bytes = need;
  }
  if (c->n_send > 0) {
    DISPATCH(launch_pack<float>(fl, c->d_send_idx, c->n_send, rowlen, q->buf, s),
             launch_pack<double>(fl, c->d_send_idx, c->n_send, rowlen, q->buf, s));
    HIPCHK(hipGetLastError());
  }
  return 0;
}

int ecwam_hip_halo_start(ecwam_hip_ctx* c, void* fl, int rowlen, void* stream) {
  if (!c) return fail("null context");
  if (c->peer.empty()) return 0;
  if (!fl || rowlen < 1) return fail("ecwam_hip_halo_start: bad arguments");
  if (!c->comm) return fail("ecwam_hip_halo_start: no communicator (ecwam_hip_comm_init)");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->device));
  ecwam_hip_ctx::HaloSlot* q = halo_slot(c, rowlen);
  if (halo_pack(c, q, fl, rowlen, s)) return 1;
  HIPCHK(hipEventRecord(q->ev_packed, s));                     // everything enqueued on `stream` so far, the pack included
  HIPCHK(hipStreamWaitEvent(c->comm_stream, q->ev_packed, 0));
  const size_t rb = (size_t)rowlen * c->real_bytes;
  NCCLCHK(g_rccl.GroupStart());
  for (size_t i = 0; i < c->peer.size(); i++) {
    if (c->send_cnt[i] > 0)
      NCCLCHK(g_rccl.Send((const char*)q->buf + (size_t)c->send_off[i] * rb, (size_t)c->send_cnt[i] * rb, ncclChar, c->peer[i], c->comm, c->comm_stream));
    if (c->recv_cnt[i] > 0)
      NCCLCHK(g_rccl.Recv((char*)fl + (size_t)c->recv_dst0[i] * rb, (size_t)c->recv_cnt[i] * rb, ncclChar, c->peer[i], c->comm, c->comm_stream));
  }
  NCCLCHK(g_rccl.GroupEnd());
  HIPCHK(hipEventRecord(q->ev_done, c->comm_stream));
  q->used = q->pending = true;
  return 0;
}

// every exchange posted and not yet waited for: `stream` continues behind all of them
int ecwam_hip_halo_finish(ecwam_hip_ctx* c, void* stream) {
  if (!c) return fail("null context");
  if (c->peer.empty()) return 0;
  for (auto& q : c->slot)
    if (q.pending) {
      HIPCHK(hipStreamWaitEvent((hipStream_t)stream, q.ev_done, 0));
      q.pending = false;
    }
  return 0;
}

int ecwam_hip_halo_pack_host(ecwam_hip_ctx* c, const void* fl, int rowlen, void* host_send, void* stream) {
  if (!c) return fail("null context");
  if (c->n_send == 0) return 0;
  if (!fl || !host_send || rowlen < 1) return fail("ecwam_hip_halo_pack_host: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipSetDevice(c->device));
  ecwam_hip_ctx::HaloSlot* q = halo_slot(c, rowlen);
  if (halo_pack(c, q, fl, rowlen, s)) return 1;
  HIPCHK(hipMemcpyAsync(host_send, q->buf, (size_t)c->n_send * rowlen * c->real_bytes, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecwam_hip_halo_unpack_host(ecwam_hip_ctx* c, void* fl, int rowlen, const void* host_recv, void* stream) {
  if (!c) return fail("null context");
  if (c->n_recv == 0) return 0;
  if (!fl || !host_recv || rowlen < 1) return fail("ecwam_hip_halo_unpack_host: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const size_t rb = (size_t)rowlen * c->real_bytes;
  size_t off = 0;
  for (size_t i = 0; i < c->peer.size(); i++) {
    if (c->recv_cnt[i] > 0)
      HIPCHK(hipMemcpyAsync((char*)fl + (size_t)c->recv_dst0[i] * rb, (const char*)host_recv + off, (size_t)c->recv_cnt[i] * rb, hipMemcpyHostToDevice, s));
    off += (size_t)c->recv_cnt[i] * rb;
  }
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int ecwam_hip_malloc(ecwam_hip_ctx* c, unsigned long long bytes, void** dptr) {
  if (!c || !dptr) return fail("ecwam_hip_malloc: null argument");
  HIPCHK(hipSetDevice(c->device));
  HIPCHK(hipMalloc(dptr, bytes ? bytes : 1));
  return 0;
}
int ecwam_hip_free(ecwam_hip_ctx* c, void* dptr) {
  if (!c) return fail("null context");
  if (dptr) HIPCHK(hipFree(dptr));
  return 0;
}
int ecwam_hip_memcpy_h2d(ecwam_hip_ctx* c, void* dst, const void* src, unsigned long long bytes, void* stream) {
  if (!c || (bytes && (!dst || !src))) return fail("ecwam_hip_memcpy_h2d: null argument");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}
int ecwam_hip_memcpy_d2h(ecwam_hip_ctx* c, void* dst, const void* src, unsigned long long bytes, void* stream) {
  if (!c || (bytes && (!dst || !src))) return fail("ecwam_hip_memcpy_d2h: null argument");
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return 0;
}
int ecwam_hip_memset(ecwam_hip_ctx* c, void* dst, int value, unsigned long long bytes, void* stream) {
  if (!c || (bytes && !dst)) return fail("ecwam_hip_memset: null argument");
  HIPCHK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return 0;
}
int ecwam_hip_sync(ecwam_hip_ctx* c, void* stream) {
  if (!c) return fail("null context");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}
int ecwam_hip_queue_create(ecwam_hip_ctx* c, void** queue) {
  if (!c || !queue) return fail("ecwam_hip_queue_create: null argument");
  HIPCHK(hipSetDevice(c->device));
  hipStream_t s;
  HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *queue = (void*)s;
  return 0;
}
int ecwam_hip_queue_destroy(ecwam_hip_ctx* c, void* queue) {
  if (!c) return fail("null context");
  if (queue) HIPCHK(hipStreamDestroy((hipStream_t)queue));
  return 0;
}
int ecwam_hip_queue_wait_for(ecwam_hip_ctx* c, void* waiter, void* waited) {
  if (!c) return fail("null context");
  if (waiter == waited) return 0;
  hipEvent_t e;
  HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(hipEventRecord(e, (hipStream_t)waited));
  HIPCHK(hipStreamWaitEvent((hipStream_t)waiter, e, 0));
  HIPCHK(hipEventDestroy(e));   // released once the recorded work has completed
  return 0;
}

}  // extern "C"
