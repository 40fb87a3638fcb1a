/*
 * ecwam_hip.h -- C ABI of libecwam_hip.so: the MI355X-native WAMINTGR hot path
 * (PROPAG_WAM/PROPAGS2 advection + NEWWIND + IMPLSCH source-term integration) of ecWAM.
 *
 * This is the drop-in boundary.  Every entry point names the reference
 * interface it replaces (paths relative to ecwam/src/ecwam).  All functions return 0 on success
 * and a non-zero status otherwise (the reference aborts through WAM_ABORT/ABORT1, yowabort.F90;
 * the Fortran wrapper fortran/wamintgr_hip.F90 turns a non-zero status into WAM_ABORT).
 * ecwam_hip_last_error() gives the message.  No torch / C++ types cross this boundary.
 *
 * Precision: `real_bytes` = 4 (JWRB single build) or 8 (double build), parkind_wave.F90:23-35.
 * Every `void*` array below holds reals of that size unless the name says int.
 *
 * DEVICE LAYOUT (private to the library; converted from the reference's chunked layout by
 * ecwam_hip_chunks_to_points / _points_to_chunks):
 *   spectra   FL[ij][K][M]           ij = local sea point (0-based), K = direction, M = frequency,
 *                                    M fastest; slot ij = nland (= "NSUP+1", propag_wam.F90:146)
 *                                    holds the zero land spectrum, halo points sit between the
 *                                    owned range and nland (mpdecomp.F90:89-100 convention)
 *   WVPRPT    [ij][5][NFRE]          WAVNUM, CGROUP, CINV, XK2CG, STOKFAC (FREQUENCY type, yowdrvtype_config.yml)
 *   FF        [ij][16]               AIRD WDWAVE CICOVER WSWAVE WSTAR USTRA VSTRA UFRIC TAUW TAUWDIR Z0M Z0B
 *                                    CHRNCK CITHICK (FORCING_FIELDS) + EMAXDPT DEPTH (ENVIRONMENT)
 *   INTF      [ij][16]               WSEMEAN WSFMEAN USTOKES VSTOKES STRNMS TAUXD TAUYD TAUOCXD TAUOCYD TAUOC
 *                                    TAUICX TAUICY PHIOCD PHIEPS PHIAW (INTGT_PARAM_FIELDS) + 1 pad
 *   WAM2NEMO  double[ij][13]         NEMOUSTOKES NEMOVSTOKES NEMOSTRN NPHIEPS NTAUOC NSWH NMWP NEMOTAUX NEMOTAUY NEMOTAUICX
 *                                    NEMOTAUICY NEMOWSWAVE NEMOPHIF (WAVE2OCEAN, always double = JWRO); only with LWNEMOCOU
 *   weights   W[ij][8][NANG*NFRE_RED] OPTIONAL (ecwam_hip_propags2_otf needs none): the 8 CTU weights PROPAGS2 reads when
 *                                    IREFRA=0 (propags2.F90:107-116):
 *                                    SUMWN, WLONN(JXO(K,1)), WLATN(JYO(K,1),1), WLATN(JYO(K,1),2),
 *                                    WCORN(1,1), WCORN(1,2), WKPMN(-1), WKPMN(+1)
 */
#ifndef ECWAM_HIP_H
#define ECWAM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define ECWAM_HIP_MAXANG 48
/* bumped whenever ecwam_hip_params / ecwam_hip_tables or an entry point changes: 2 = refraction entry points, SDICE1 table and
 * ice break-up parameters added.  ecwam_hip_abi_version() returns the value the library was built with. */
#define ECWAM_HIP_ABI_VERSION 6
#define ECWAM_HIP_MAXFRE 48
#define ECWAM_HIP_MAXMC 56     /* MLSTHG = NFRE - ISM <= 48 + 8 */
#define ECWAM_HIP_MAXTAP 47    /* 2*NSDSNTH+1, NSDSNTH <= NANG/2-1 */
#define ECWAM_HIP_MAXGC 96     /* NWAV_GC (initgc.F90:67) = 82 */
#define ECWAM_HIP_IAB 200      /* yowtabl.F90:25 */
#define ECWAM_HIP_JTOT_TAUHF 19 /* yowcoup.F90:60 */
#define ECWAM_HIP_NFF 16
#define ECWAM_HIP_NINTF 16
#define ECWAM_HIP_NWPR 5

/* Scalars and flags the kernels read from the reference's modules (replaces `!$loki update_device`
 * of the module globals, wamintgr_loki_gpu.F90:95-97).  Reals are passed as double and rounded
 * to the working precision inside the library. */
typedef struct ecwam_hip_params {
  /* YOWPARAM / YOWSTAT / YOWWNDG */
  int nang, nfre, nfre_red, nfre_odd;
  int idelt;                /* IDELT: source time step [s] */
  double ximp;              /* XIMP */
  int iphys, isnonlin, irefra, icode;
  /* YOWCOUP / YOWSTAT / YOWICE flags (0/1) */
  int llgcbz0, llnormagam, llcapchnk, lbiwbk, licerun, lmaskice, lwamrsetci;
  int lciwa1, lciwa2, lciwa3, lciscal;
  int lwvflx_snl, lwflux, lwfluxout, lwnemocou, lwcou, lwcouast, lwnemocouwrs, lwnemotauoc;
  int lwnemocousend, lwnemocoustk, lwnemocoustrn; /* YOWCOUP: which WAVE2OCEAN members STOKESTRN fills (stokestrn.F90:75-88) */
  /* YOWPCONS */
  double g, gm1, pi, zpi, zpi4gm1, zpi4gm2, epsmin, rowater, rowaterm1, epsus, epsu10, acd, bcd, acdlin, bcdlin, cdmax;
  double tauocmin, tauocmax, phiepsmin, phiepsmax, wsemean_min, circ, r_earth;
  /* YOWFRED scalars */
  double fratio, wetail, frtail, wp1tail, fric, delth, flogsprdm1;
  /* YOWPHYS */
  double xkappa, xnlev, rnu, rnum, betamaxoxkappa2, bmaxokap, gamnconst, zalp, alpha, alphamin, alphamax, chnkmin_u, alphapmax;
  double tauwshelter, dthrn_a, dthrn_u, tailfactor, tailfactor_pm, ang_gc_a, ang_gc_b, ang_gc_c, rn1_rn;
  double swellf, swellf2, swellf3, swellf4, swellf5, swellf6, swellf7, swellf7m1, z0rat, z0tubmax, abmin, abmax;
  double sdsbr, ssdsc2, ssdsc3, ssdsc4, ssdsc5, ssdsc6, miche;
  int nsdsnth, ipsat;
  double egrcrv, afcrv, bfcrv;
  /* YOWCOUP / YOWTABL / YOWICE / YOWSHAL / YOWWIND */
  double x0tauhf, eps1, flmin, cithrsh, ciblock, cithrsh_tail, zalpwrs, bathymax, wspmin, wspmin_reset_tauw;
  double cdis, delta_sdis, cdisvis; int idamping; /* YOWPHYS / YOWSTAT: IPHYS=0 dissipation and swell damping switch */
  double cdicwa, zalpfacb, zalpfacx; /* YOWICE: SDICE2 drag coefficient, attenuation scale factors (sdice2.F90, implsch.F90:195) */
  /* YOWICE / YOWCOUP: ice break-up coupling (icebreak_modify_attenuation.F90) and the SDICE1 scattering table geometry
   * (cigetdeac.F90:64-71, yowice.F90:23) */
  int lwnemocouibr;
  double zibrw_thrsh;
  int nict, nich;
  double ticmin, dtic, dhic, hicmin;
  /* YOWINDN scalars */
  int mfrstlw, mlsthg, kfrh;
  double dal1, dal2;
  /* gravity-capillary model (YOWFRED *_GC) */
  int nwav_gc;
  double xlogkratiom1_gc, sqrtgosurft;
} ecwam_hip_params;

/* Host pointers to the reference's module tables, reals in the working precision, ints 32-bit.
 * Index values are passed exactly as the reference holds them (1-based where the reference is
 * 1-based); INDICESSAT is 1-based on this interface as in yowphys.F90:154. */
typedef struct ecwam_hip_tables {
  const void *fr, *dfim, *dfimofr, *dfimfr, *dfim_sim, *rhowg_dfim, *zpifr, *fr5, *cofrm4, *flmax; /* [NFRE]  yowfred.F90 */
  const void *th, *costh, *sinth;                                                                     /* [NANG] */
  const void *wtauhf;                                                                                 /* [19]   yowcoup.F90:61 */
  const void *swellft;                                                                                /* [200]  yowtabl.F90:56 */
  const int *ikp, *ikp1, *ikm, *ikm1;                                                                 /* [MLSTHG] values for MC=1.. (yowindn) */
  const void *af11;                                                                                   /* [MLSTHG] AF11(1:MLSTHG) */
  const int *k1w, *k2w, *k11w, *k21w;                                                                 /* [NANG][2] C order of K1W(K,KH) */
  const int *inlcoef;                                                                                 /* [MLSTHG][5]  INLCOEF(:,MC) */
  const void *rnlcoef;                                                                                /* [MLSTHG][25] RNLCOEF(:,MC) */
  const int *indicessat;                                                                              /* [NANG][2*NSDSNTH+1] */
  const void *satweights;                                                                             /* [NANG][2*NSDSNTH+1] */
  const int *kpm, *jxo, *jyo, *kcr;                                                                   /* [NANG][3],[2],[2],[4] (yowubuf) */
  const void *xk_gc, *xkm_gc, *omega_gc, *omxkm3_gc, *cm_gc, *c2osqrtvg_gc, *xkmsqrtvgoc2_gc, *om3gmkm_gc, *delkcc_gc_ns,
      *delkcc_omxkm3_gc;                                                                              /* [NWAV_GC] */
  const void *cideac;                                                                                 /* [NICH][NICT] = CIDEAC(IT,IH) (yowice); may be NULL unless LCIWA1 */
} ecwam_hip_tables;

typedef struct ecwam_hip_ctx ecwam_hip_ctx;

const char *ecwam_hip_last_error(void);
int ecwam_hip_abi_version(void);
/* diagnostics: runs the wavefront reduction / permutation primitives of the kernels on `device` against serial sums;
   0 = all agree.  No counterpart in the reference (the OpenACC build relies on the compiler's reductions). */
int ecwam_hip_selftest(int device);

/* Context: one per (device, configuration).  Replaces the device residency of the module globals. */
int ecwam_hip_create(const ecwam_hip_params *p, const ecwam_hip_tables *t, int real_bytes, int device, ecwam_hip_ctx **out);
int ecwam_hip_destroy(ecwam_hip_ctx *ctx);

/*
 * LSUBGRID: sub-grid obstructions of YOWUBUF (OBSLAT, OBSLON, OBSCOR; getbobstrct.F90 reads them, ctuw.F90:703-733 scales
 * the space weights with them).  obs: device real OBS[n][8][NFRE], planes OBSLAT(IJ,M,1:2), OBSLON(IJ,M,1:2), OBSCOR(IJ,M,1:4)
 * (values in [0,1], 1 = open); the library keeps the POINTER (the caller owns the memory) and every later ecwam_hip_ctuw /
 * ecwam_hip_propags2_otf / ecwam_hip_propags2_refra applies it.  obs = NULL switches it off (LSUBGRID = F, the default).
 */
int ecwam_hip_set_obstructions(ecwam_hip_ctx *ctx, const void *obs, int n);

/*
 * PROPAGS2 (propags2.F90:10, IREFRA=0 branch :99-121) on device pointers.
 *   f1, f3      FL[npts+1][NANG][NFRE]; only rows [kijs,kijl) of f3 are written; frequencies
 *               [nd3s-1, nd3e) are advected (1-based inclusive as in the reference);
 *               if copy_rest != 0 the remaining frequencies of those rows are copied f1 -> f3
 *               (the reference leaves them in FL1, propag_wam.F90:379-386)
 *   klon[ij][2] klat[ij][2][2] kcor[ij][4][2]  0-based local indices, land = nland (yowubuf KLON/KLAT/KCOR)
 *   w           W[ij][8][NANG*NFRE_RED]
 */
int ecwam_hip_propags2(ecwam_hip_ctx *ctx, const void *f1, void *f3, const int *klon, const int *klat, const int *kcor,
                       const void *w, int kijs, int kijl, int nd3s, int nd3e, int copy_rest, void *stream);

/*
 * CTUWINI + CTUW (ctuwini.F90:58-164, ctuw.F90:146-275 + :407-484 + :536-608) for IREFRA=0, ICASE=1,
 * LSUBGRID=F: builds W for rows [0,n) and frequencies [mstart,mend] (1-based) with time step delpro.
 *   kxlt[ij] 0-based latitude row; zdello/cosph/sinph [ngy] (yowmap ZDELLO, yowgrid COSPH/SINPH)
 *   wlat[ij][2], wcor[ij][4] are MODIFIED near land exactly as ctuwini.F90:66-97 does
 *   cgroup_ext [npts+1][NFRE] group velocity incl. halo and land rows (proenvhalo.F90 BUFFER_EXT)
 *   cosphm1_ext[npts+1]
 *   cflfail  int[n] set to 1 where a CFL/weight-range check of ctuw.F90:288-357,541-685 fails
 *   w        may be NULL: checks and WLAT/WCOR snapping only (see ecwam_hip_propags2_otf)
 */
int ecwam_hip_ctuw(ecwam_hip_ctx *ctx, int n, int nland, int ngy, double delpro, int mstart, int mend, const int *kxlt,
                   const void *zdello, double xdella, const void *cosph, const void *sinph, const int *klon, const int *klat,
                   const int *kcor, void *wlat, void *wcor, const void *cgroup_ext, const void *cosphm1_ext, void *w,
                   int *cflfail, void *stream);

/*
 * PROPAGS2 with the CTU weights rebuilt inside the stencil ("on the fly") instead of read from W: the device-side
 * replacement of PROPAGS2 + the WLATN/WLONN/WCORN/WKPMN/SUMWN module arrays of yowubuf.F90:175-193 when the weights are
 * static (IREFRA=0: ctuwupdt.F90:204-238 builds them once).  Same arguments as ecwam_hip_ctuw (the geometry, with
 * wlat/wcor as LEFT BY ecwam_hip_ctuw, i.e. after the CTUWINI land snapping) plus those of ecwam_hip_propags2; the
 * result is bit-identical to ecwam_hip_ctuw followed by ecwam_hip_propags2 with the same delpro.  ecwam_hip_ctuw with
 * w == NULL performs the CTUWINI snapping and the CFL / weight-range checks without storing the weights.
 *   order  optional int[>=kijl]: the rows order[kijs..kijl) are processed, in that sequence (chosen by the host for cache
 *          locality, e.g. longitude strips); entries < 0 are padding and skipped, so the list may be longer than n; NULL =
 *          natural order.  Results do not depend on it.
 *   copy_rest  bit 0: the frequencies outside nd3s..nd3e are copied from f1; bit 2 (value 4, needs order): the list describes
 *          2-D tiles of 16 entries, entry 4 g + w = g-th point of a longitude segment of the w-th of four adjacent latitude
 *          rows; the four wavefronts of a workgroup then work on latitude neighbours at the same time (40 % less fabric
 *          read traffic at O320, same time: DESIGN.md section 3).
 */
int ecwam_hip_propags2_otf(ecwam_hip_ctx *ctx, const void *f1, void *f3, int n, int ngy, double delpro, const int *kxlt,
                           const void *zdello, double xdella, const void *cosph, const void *sinph, const int *klon,
                           const int *klat, const int *kcor, const void *wlat, const void *wcor, const void *cgroup_ext,
                           const void *cosphm1_ext, const int *order, int kijs, int kijl, int nd3s, int nd3e, int copy_rest,
                           void *stream);

/*
 * The fast / slow wave split of PROPAG_WAM (IFRELFMAX > 0, propag_wam.F90:247-313) in one pass: frequencies 1..ifrelfmax advance
 * with delpro_lf, the others with delpro (the reference calls PROPAGS2 once per range with separately built weights).
 * ifrelfmax = 0: identical to ecwam_hip_propags2_otf.  Bit-identical to the two separate calls.
 * in_nfre: 0 (f1 has the FL layout) or the width of a COMPACT input buffer f1[npts+1][NANG][in_nfre] that holds only the first
 * in_nfre frequencies of every direction (the fast waves between sub-steps: with M fastest in memory a frequency sub-range of
 * the full rows touches every cache line; the compact rows are 4.5x smaller at IFRELFMAX = 5).  f3 always has the FL layout.
 * gout (may be NULL): compact buffer gout[npts+1][NANG][gout_nfre] that ALSO receives the first gout_nfre advected frequencies of
 * every direction (what the next fast-wave sub-step starts from; gout_nfre a multiple of 16 bytes).
 */
int ecwam_hip_propags2_otf_split(ecwam_hip_ctx *ctx, const void *f1, void *f3, int n, int ngy, double delpro, double delpro_lf,
                                 int ifrelfmax, int in_nfre, void *gout, int gout_nfre, const int *kxlt, const void *zdello, double xdella, const void *cosph,
                                 const void *sinph, const int *klon, const int *klat, const int *kcor, const void *wlat,
                                 const void *wcor, const void *cgroup_ext, const void *cosphm1_ext, const int *order, int kijs,
                                 int kijl, int nd3s, int nd3e, int copy_rest, void *stream);
/*
 * The same with the two further row formats of the fast-wave sub-steps (round 3): the sub-steps 1 .. NSTEP_LF-1 of the fast waves run
 * compact -> compact BEFORE the full pass (the fast waves do not depend on the slow ones: the same arithmetic in another order, hence
 * the same bits as propag_wam.F90:247-313), and the full pass takes their last state as the input of the last sub-step and writes
 * complete rows -- no pass writes a frequency sub-range into full rows (which touches every line of them).
 *   gin / gin_nfre (may be NULL / 0; needs full input rows, in_nfre = 0): the first gin_nfre frequencies of every direction are read
 *     from the compact buffer gin[npts+1][NANG][gin_nfre] (own point and neighbours alike) instead of f1.
 *   out_nfre: 0 (f3 has the FL layout) or the width of a compact OUTPUT buffer f3[npts+1][NANG][out_nfre] (then gout must be NULL; with a
 *     compact input of the same width copy_rest carries the frequencies beyond nd3e over, so that whole 16-byte vectors stay valid).
 */
int ecwam_hip_propags2_otf_fast(ecwam_hip_ctx *ctx, const void *f1, void *f3, int n, int ngy, double delpro, double delpro_lf,
                                int ifrelfmax, int in_nfre, const void *gin, int gin_nfre, int out_nfre, void *gout, int gout_nfre,
                                const int *kxlt, const void *zdello, double xdella, const void *cosph, const void *sinph,
                                const int *klon, const int *klat, const int *kcor, const void *wlat, const void *wcor,
                                const void *cgroup_ext, const void *cosphm1_ext, const int *order, int kijs, int kijl, int nd3s,
                                int nd3e, int copy_rest, void *stream);
/* g (may be NULL: off) = compact rows g[npts+1][NANG][g_nfre] that every later ecwam_hip_implsch / ecwam_hip_nosource call ALSO leaves the
 * first g_nfre frequencies of the spectra it wrote in: the state the next advection step's fast-wave sub-steps start from, written from
 * IMPLSCH's tile instead of extracted from the FL1 rows by a pass of its own (ecwam_hip_copy_freq_range).  The library keeps the pointer. */
int ecwam_hip_set_fastwave_copy(ecwam_hip_ctx *ctx, void *g, int g_nfre);
/* dst[ij][K][m_first-1 .. m_last-1] = src[...] for rows [0,n): FL1_EXT(:,:,1:IFRELFMAX) = FL3_EXT(...) between the fast-wave
 * sub-steps (propag_wam.F90:287-291).  dst_nfre: 0 (dst has the FL layout) or the width of a compact buffer dst[..][NANG][dst_nfre] */
int ecwam_hip_copy_freq_range(ecwam_hip_ctx *ctx, const void *src, void *dst, int n, int m_first, int m_last, int dst_nfre,
                              void *stream);

/*
 * Refraction, IREFRA = 1 (depth), 2 (currents), 3 (depth + currents) -- params.irefra selects it at ecwam_hip_create.
 *
 * ecwam_hip_propdot: GRADI + the per-point part of PROPDOT (gradi.F90:113-232, propdot.F90:108-196), called whenever the
 * reference sets LLUPDTTD (new currents, propag_wam.F90:175-213).  Replaces the module arrays THDD/THDC/SDOT of yowubuf:
 *   depth_ext, u_ext, v_ext [npts+1]  DEPTH / UCUR / VCUR incl. halo rows and the land slot as proenvhalo.F90:99-106
 *                                     fills it (BATHYMAX, 0, 0)
 *   wlat[ij][2]                       as read from the grid tables (before or after CTUWINI: GRADI only uses entries
 *                                     CTUWINI never changes)
 *   refr  out, real[n][2*NANG+5]:     THD(K) (THDD for IREFRA=1, THDC for 2/3), S0(K) (current-gradient factor of SDOT,
 *                                     propdot.F90:172), U, V, OMDD (propdot.F90:126), CURMASK of frequency range 0 and 1 (=1)
 *
 * ecwam_hip_ctuw_refra: CTUWINI + CTUWDRV (ctuwupdt.F90:204-238, ctuwdrv.F90:93-118) without storing weights: snaps
 * wlat/wcor, runs the CFL / range checks of every weight for frequencies [mstart,mend] with time step delpro, and when
 * llcflcuroff != 0 and a point fails with currents, repeats them with the current refraction and frequency shift of that
 * point switched off (its CURMASK in refr becomes 0, as ctuw.F90:117-131).  cflfail[n]: points still failing (the caller
 * aborts, ctuwdrv.F90:124-146).  range = 0 for the only (or the fast-wave) frequency range, 1 for the slow-wave range of a
 * split call (ctuwupdt.F90:220-256 runs CTUWDRV once per range, each with its own CURMASK); range 0 resets cflfail, range 1
 * adds its failures to those of range 0.
 *
 * ecwam_hip_propags2_refra: PROPAGS2 (propags2.F90:99-192) with all weights rebuilt inside the stencil from the geometry,
 * cgroup_ext, the point's own omosnh2kd/wavnum rows and refr.  IREFRA=1: the eight-term stencil with depth refraction in
 * WKPMN; IREFRA=2/3: all 2+4+8 space neighbours, both direction and both frequency neighbours, in the reference's
 * summation order.  Arguments as ecwam_hip_propags2_otf; range selects the CURMASK of ecwam_hip_ctuw_refra.
 */
int ecwam_hip_propdot(ecwam_hip_ctx *ctx, int n, int nland, const int *kxlt, const void *zdello, double xdella,
                      const void *cosph, const int *klon, const int *klat, const void *wlat, const void *cosphm1_ext,
                      const void *depth_ext, const void *u_ext, const void *v_ext, void *refr, void *stream);
int ecwam_hip_ctuw_refra(ecwam_hip_ctx *ctx, int n, int nland, int ngy, double delpro, int mstart, int mend, const int *kxlt,
                         const void *zdello, double xdella, const void *cosph, const void *sinph, const int *klon,
                         const int *klat, const int *kcor, void *wlat, void *wcor, const void *cgroup_ext,
                         const void *omosnh2kd_ext, const void *wavnum_ext, const void *cosphm1_ext, void *refr,
                         int llcflcuroff, int range, int *cflfail, void *stream);
int ecwam_hip_propags2_refra(ecwam_hip_ctx *ctx, const void *f1, void *f3, int n, int ngy, double delpro, const int *kxlt,
                             const void *zdello, double xdella, const void *cosph, const void *sinph, const int *klon,
                             const int *klat, const int *kcor, const void *wlat, const void *wcor, const void *cgroup_ext,
                             const void *omosnh2kd_ext, const void *wavnum_ext, const void *cosphm1_ext, const void *refr,
                             int range, int kijs, int kijl, int nd3s, int nd3e, int copy_rest, void *stream);

/*
 * IMPLSCH (implsch.F90:10-23) for local points [kijs,kijl) on device pointers (layouts above).
 *   fl1 inout, wvprpt in, ff inout, intf inout, mij out (1-based), xllws out
 *   intf[ij][15] (the spare slot of the 16-wide row) carries the INPUT ENVIRONMENT%IBRMEM when LWNEMOCOUIBR
 *   wam2nemo: double[npts][13] inout, the WAVE2OCEAN members (always JWRO = double, yowdrvtype_config.yml:44-55) in the order
 *             NEMOUSTOKES NEMOVSTOKES NEMOSTRN NPHIEPS NTAUOC NSWH NMWP NEMOTAUX NEMOTAUY NEMOTAUICX NEMOTAUICY NEMOWSWAVE
 *             NEMOPHIF; updated as wnfluxes.F90:304-328 (LNUPD = T) and stokestrn.F90:75-88 do; required when LWNEMOCOU,
 *             ignored (may be NULL) otherwise
 *   dbg: must be NULL (ABI 4: an intermediate dump of the retired one-point-per-wavefront kernel)
 */
int ecwam_hip_implsch(ecwam_hip_ctx *ctx, int kijs, int kijl, void *fl1, const void *wvprpt, void *ff, void *intf, int *mij,
                      void *xllws, double *wam2nemo, void *dbg, void *stream);
/*
 * IMPLSCH runs ONE kernel generation since ABI 5: k_implsch4 (csrc/implsch_v4.h: several sea points per wavefront; common builds for the
 * configurations the reference registers as tests, RARE builds for every other switch of SURVEY.md 8a), 48 / 36 / 24 / 12 directions x 36
 * frequencies, single and double precision.  ecwam_hip_create refuses what no build covers.  (ABI 4 carried a second, slower generation for
 * the remainder and ecwam_hip_set_implsch_generation to choose; that kernel is now test infrastructure, tests/csrc/implsch_v2.h.)
 * The generation the last ecwam_hip_implsch call of this context launched (4; 0 before the first call). */
int ecwam_hip_implsch_generation_used(ecwam_hip_ctx *ctx);
/* The device copy of the module tables the kernels read (private layout, csrc/dev.h DevTab): for diagnostics and for a second
 * implementation of a kernel that is to run on exactly the tables the library runs on (the tests' k_implsch2).  Owned by the context. */
const void *ecwam_hip_device_tables(ecwam_hip_ctx *ctx);
/*
 * k_implsch4 is bracketed by two one-point-per-lane kernels (first TAUT_Z0 before, second STRESSO and WNFLUXES after it) that exchange
 * 36 scalars per sea point through a context-owned device buffer indexed by the point number.  ecwam_hip_implsch grows that buffer when
 * kijl exceeds what it holds -- a device allocation and a device-wide wait inside an otherwise stream-ordered call.  A host that wants
 * none of that in its time loop (or captures the step into a hipGraph) sizes it once: npts = the largest kijl it will pass.  Calls on
 * several streams may share the buffer as long as their [kijs,kijl) do not overlap.  The same call sizes the tables of the one-kernel step
 * (ecwam_hip_propags2_implsch: 12 reals per point) where the context has such a build.
 */
int ecwam_hip_implsch_reserve(ecwam_hip_ctx *ctx, int npts);

/*
 * One WAMINTGR step with a 1:1 ratio of advection and source-term steps as ONE pass over the spectra (wamintgr.F90:94-146: PROPAG_WAM then
 * IMPLSCH; propag_wam.F90:124-147,247-251,373-400; propags2.F90:99-121): the kernel that integrates the source terms of rows [kijs,kijl)
 * advects them itself while it loads them -- PROPAGS2 (IREFRA = 0, one time step for every frequency, CTU weights rebuilt on the fly from the
 * arguments of ecwam_hip_propags2_otf) from the rows of f1 (owned + halo + land rows: read only) straight into the kernel's working tile --
 * and stores the new spectrum to the rows of f3.  The advected spectrum never goes to memory.  Result: bit for bit what
 * ecwam_hip_propags2_otf(f1 -> f3, copy_rest = 1) followed by ecwam_hip_implsch(f3) leaves in f3, FF, INTF, MIJ, XLLWS, WAM2NEMO.
 * The caller runs NEWWIND (ecwam_hip_newwind) BEFORE this call (it touches the forcing only) and swaps f1 / f3 after it, as after PROPAGS2.
 * Rows whose stencil reads halo rows are passed in a second call behind ecwam_hip_halo_finish, exactly as with ecwam_hip_propags2_otf.
 * Fast waves (the native O1280 cycle, propag_wam.F90:247-313): ifrelfmax > 0 with gin = the compact rows [rows][NANG][gin_nfre] that hold the
 * fast waves after their sub-steps 1 .. NSTEP_LF - 1 (ecwam_hip_propags2_otf_fast, compact -> compact): the call is then the LAST sub-step
 * of the fast waves (time step delpro_lf, read from gin) together with the slow waves' step (delpro, read from f1) -- what
 * ecwam_hip_propags2_otf_fast(f1 -> f3, gin) does -- and the source terms; with ecwam_hip_set_fastwave_copy the new fast waves also go to
 * the compact rows the next advection step starts from.  Else ifrelfmax = 0, gin = NULL.
 * Covered: what ecwam_hip_propags2_implsch_supported reports (48 / 36 / 24 / 12 directions x 36 frequencies in single, 36 x 36 in double precision, the common builds of
 * IMPLSCH with IPHYS = 1 / ISNONLIN = 0, no refraction; with or without the obstructions of ecwam_hip_set_obstructions); everything else runs
 * the two calls.
 * flags: 0 (bit 0: workgroups in the XCD-aware order of the stencil kernel instead of the natural one; bit 1: the go / no-go probe of
 * diagnostics builds).
 */
/* 0: no one-kernel build covers the context; else a mask: bit 0 the plain step, bit 1 also with fast waves (gin), bit 2 also with obstructions */
int ecwam_hip_propags2_implsch_supported(ecwam_hip_ctx *ctx);
int ecwam_hip_propags2_implsch(ecwam_hip_ctx *ctx, const void *f1, void *f3, int n, int ngy, double delpro, const int *kxlt, const void *zdello,
                               double xdella, const void *cosph, const void *sinph, const int *klon, const int *klat, const int *kcor,
                               const void *wlat, const void *wcor, const void *cgroup_ext, const void *cosphm1_ext, int kijs, int kijl,
                               int nd3s, int nd3e, const void *wvprpt, void *ff, void *intf, int *mij, void *xllws, double *wam2nemo,
                               double delpro_lf, int ifrelfmax, const void *gin, int gin_nfre, int flags, void *stream);

/*
 * Integrated output parameters without a spectrum copy-back (the device-side part of OUTBS: outblock.F90:204,223-243,
 * LSECONDORDER=F) for rows [kijs,kijl):  out[npts][5] = significant wave height 4*SQRT(EM) (FEMEAN), mean direction in
 * degrees / meteorological convention (STHQ), mean period 1/FM or zmiss, EM, peak period pp1d (DOMINANT_PERIOD) or zmiss.
 * ecwam_hip_outwnorm: the OUTWNORM statistics of one such field: result[4] (HOST doubles) = average, minimum, maximum over the
 * n values field[i*stride] that differ from zmiss, and their count (outwnorm.F90).  Synchronises the stream.
 */
int ecwam_hip_outbs(ecwam_hip_ctx *ctx, int kijs, int kijl, const void *fl1, double zmiss, void *out, void *stream);
int ecwam_hip_outwnorm(ecwam_hip_ctx *ctx, const void *field, int stride, int n, double zmiss, double *result, void *stream);

/* NEWWIND forcing hand-over (newwind.F90:126-161): FF <- FF_NEXT members + TAUW cap.  ecwam_hip_newwind takes ICODE_WND = ICODE
 * of the parameters; a coupled host (LWCOU) passes ICODE_CPL through ecwam_hip_newwind_icode (newwind.F90:120-124). */
int ecwam_hip_newwind(ecwam_hip_ctx *ctx, int n, void *ff, const void *ff_next, void *stream);
int ecwam_hip_newwind_icode(ecwam_hip_ctx *ctx, int n, void *ff, const void *ff_next, int icode_wnd, void *stream);
/* The LLSOURCE = F branch of WAMINTGR (wamintgr.F90:152-160) on device rows [kijs, kijl): FL1 = MAX(FL1, EPSMIN), MIJ = NFRE,
 * XLLWS = 0.  fl1 == NULL: the branch of a call before the next source-term date (wamintgr.F90:178-186): MIJ = NFRE, XLLWS = 0 only. */
int ecwam_hip_nosource(ecwam_hip_ctx *ctx, int kijs, int kijl, void *fl1, int *mij, void *xllws, void *stream);

/*
 * Layout conversion between the reference's chunked host-shaped arrays and the device layout
 * (replaces the chunk<->block copies of propag_wam.F90:124-137,373-400).
 *   chunked spectra: FL1(NPROMA,NANG,NFRE,NCHNK) Fortran order; point ij = ichnk*nproma + iprm (mchunk.F90:62-68)
 *   pad lanes (iprm >= KIJL4CHNK) replicate lane 1 on the way back (propag_wam.F90:388-398)
 */
int ecwam_hip_chunks_to_points(ecwam_hip_ctx *ctx, const void *chunked, void *points, int nproma, int nchnk, int npts,
                               int n2, int n3, void *stream);
int ecwam_hip_points_to_chunks(ecwam_hip_ctx *ctx, const void *points, void *chunked, int nproma, int nchnk, int npts,
                               int n2, int n3, void *stream);

/*
 * One member of a FIELD_API-backed host type <-> its slot in the library's packed per-point rows: what FIELD_API's per-member
 * GET_DEVICE_DATA_* / SYNC_DEVICE_* (host -> device) and GET_HOST_DATA_* / SYNC_HOST_* (device -> host) copies are in the reference's
 * GPU build (drvtype_mod.fypp:116-480; the member selectors of wamodel.F90:207-226,376-385,435-470,614-642 and
 * wamintgr_loki_gpu.F90:100-157,197-200), for a device layout that is not the host's.
 *   chunked: DEVICE image of the host member M(NPROMA[,NM],NCHNK) (Fortran order, as copied with ecwam_hip_memcpy_h2d / _d2h),
 *     elements of elem_bytes = 4 or 8 bytes (JWRB reals, JWRO doubles, JWIM integers move alike);
 *   rows: device [npts][row_stride] elements of the same size; the member's NM values of point ij at rows[ij*row_stride + row_off .. + NM):
 *     FF rows (row_stride 16: the 14 FORCING_FIELDS members + ENVIRONMENT%EMAXDPT, %DEPTH), INTF rows (16: 15 INTGT_PARAM_FIELDS members +
 *     ENVIRONMENT%IBRMEM), WAVE2OCEAN rows (13 doubles), WVPRPT rows (5*NFRE: member j at row_off j*NFRE, NM = NFRE), MIJ (1 int32);
 *   scatter: chunked -> rows for the npts points; gather: rows -> chunked, pad lanes of the last chunk replicating its lane 1.
 */
int ecwam_hip_member_scatter(ecwam_hip_ctx *ctx, const void *chunked, void *rows, int nproma, int nchnk, int npts, int nm,
                             long long row_stride, long long row_off, int elem_bytes, void *stream);
int ecwam_hip_member_gather(ecwam_hip_ctx *ctx, const void *rows, void *chunked, int nproma, int nchnk, int npts, int nm,
                            long long row_stride, long long row_off, int elem_bytes, void *stream);

/* Halo pack/unpack for the advection exchange (mpexchng.F90:124-138, 217-231):
 *   pack:   buf[i][:] = fl[idx[i]][:]   for i < n   (row = NANG*NFRE reals)
 *   unpack: fl[dst0+i][:] = buf[i][:]  */
int ecwam_hip_pack_rows(ecwam_hip_ctx *ctx, const void *fl, const int *idx, int n, void *buf, void *stream);
int ecwam_hip_unpack_rows(ecwam_hip_ctx *ctx, const void *buf, int n, void *fl, int dst0, void *stream);

/*
 * MPEXCHNG inside the library (mpexchng.F90:141-231: pack the rows the neighbouring ranks need, point-to-point exchange,
 * receive into the halo rows): what a Fortran host needs to run the sea-point block decomposition (mpdecomp.F90:58-100) on
 * several GPUs, one process per GPU.
 *   ecwam_hip_halo_setup: rank / nranks of this process and, per neighbouring rank i < npeers: peer[i], send_count[i] owned local
 *     rows to send (send_idx: their 0-based local indices, the npeers lists concatenated), and the contiguous halo segment
 *     [recv_dst0[i], recv_dst0[i] + recv_count[i]) of the local row space their rows land in (the reference's NTOPE / IJTOPE and
 *     NFROMPE / NIJSTART).
 *   transport 1, RCCL over xGMI (librccl is loaded with dlopen on first use): ecwam_hip_comm_unique_id on rank 0 gives the 128-byte
 *     id the host broadcasts (MPI_Bcast); ecwam_hip_comm_init(ctx, id) on every rank (collective).  ecwam_hip_halo_start packs on
 *     `stream` and posts the grouped sends / receives on the library's own stream (the receives write fl's halo rows; rowlen = reals
 *     per row: NANG*NFRE, or NANG*LFP for the compact fast-wave rows); work enqueued on `stream` afterwards overlaps with the
 *     exchange and must not touch fl's halo rows until ecwam_hip_halo_finish(ctx, stream) has made `stream` wait for it.
 *   transport 2, host staged (an MPI library without device-pointer support; several ranks on one GPU in tests):
 *     ecwam_hip_halo_pack_host fills host_send (n_send rows, peer order, ecwam_hip_halo_counts gives the sizes) and synchronises; the
 *     host exchanges the segments; ecwam_hip_halo_unpack_host copies host_recv (n_recv rows, peer order) into the halo rows.
 */
int ecwam_hip_halo_setup(ecwam_hip_ctx *ctx, int rank, int nranks, int npeers, const int *peer, const int *send_count,
                         const int *send_idx, const int *recv_dst0, const int *recv_count);
int ecwam_hip_halo_counts(ecwam_hip_ctx *ctx, int *n_send, int *n_recv);
int ecwam_hip_comm_unique_id(void *id128);
int ecwam_hip_comm_init(ecwam_hip_ctx *ctx, const void *id128);
int ecwam_hip_comm_count(ecwam_hip_ctx *ctx, int *nranks);   /* ranks of the RCCL communicator (ncclCommCount); 0 without one */
int ecwam_hip_halo_start(ecwam_hip_ctx *ctx, void *fl, int rowlen, void *stream);
int ecwam_hip_halo_finish(ecwam_hip_ctx *ctx, void *stream);
int ecwam_hip_halo_pack_host(ecwam_hip_ctx *ctx, const void *fl, int rowlen, void *host_send, void *stream);
int ecwam_hip_halo_unpack_host(ecwam_hip_ctx *ctx, void *fl, int rowlen, const void *host_recv, void *stream);

/*
 * PROENVHALO (proenvhalo.F90:63-107) on the device: the extended (own + halo + land) rows of the fields the weights and the refraction
 * terms are built from, assembled without a round trip through the host when the currents (or the depth) change on a decomposed grid.
 *   buffer_ext: device real [nrows][3*NFRE + 3] -- per local row WAVNUM(1:NFRE), CGROUP(1:NFRE), OMOSNH2KD(1:NFRE), DEPTH, UCUR, VCUR (the
 *     reference's BUFFER_EXT without DELLAM1 / COSPHM1, which the geometry arrays of the set-up hold).
 *   ecwam_hip_proenvhalo_pack fills the n owned rows from the device-resident WVPRPT rows (members 0, 1), OMOSNH2KD[n][NFRE] and the
 *     per-point DEPTH / UCUR / VCUR;  the halo rows [n, nrows) are then exchanged like the spectra: ecwam_hip_halo_start(ctx, buffer_ext,
 *     3*NFRE + 3, stream) + ecwam_hip_halo_finish (or the host-staged pair) = MPEXCHNG(BUFFER_EXT, 3*NFRE_RED+5, 1, 1);
 *   ecwam_hip_proenvhalo_unpack spreads the nrows rows over the extended arrays [nrows + 1][..] the advection entry points read and fills
 *     the land slot (row nrows) from land[3*NFRE + 3] = (WVPRPT_LAND%WAVNUM, %CGROUP, %OMOSNH2KD, BATHYMAX, 0, 0).
 */
int ecwam_hip_proenvhalo_pack(ecwam_hip_ctx *ctx, int n, const void *wvprpt, const void *omosnh2kd, const void *depth,
                              const void *ucur, const void *vcur, void *buffer_ext, void *stream);
int ecwam_hip_proenvhalo_unpack(ecwam_hip_ctx *ctx, int nrows, const void *buffer_ext, const void *land, void *wavnum_ext,
                                void *cgroup_ext, void *omosnh2kd_ext, void *depth_ext, void *u_ext, void *v_ext, void *stream);

/* Device-memory helpers for hosts without their own HIP binding (the Fortran layer): the counterpart of FIELD_API's
 * device allocation / GET_DEVICE_DATA / SYNC_HOST copies (drvtype_mod.fypp:116-480).  `stream` may be NULL. */
int ecwam_hip_malloc(ecwam_hip_ctx *ctx, unsigned long long bytes, void **dptr);
int ecwam_hip_free(ecwam_hip_ctx *ctx, void *dptr);
int ecwam_hip_memcpy_h2d(ecwam_hip_ctx *ctx, void *dst_dev, const void *src_host, unsigned long long bytes, void *stream);
int ecwam_hip_memcpy_d2h(ecwam_hip_ctx *ctx, void *dst_host, const void *src_dev, unsigned long long bytes, void *stream);
int ecwam_hip_memset(ecwam_hip_ctx *ctx, void *dst_dev, int value, unsigned long long bytes, void *stream);
int ecwam_hip_sync(ecwam_hip_ctx *ctx, void *stream);
/* Asynchronous queues (FIELD_API's QUEUE= of SYNC_HOST_* / SYNC_DEVICE_*, WAIT_FOR_ASYNC_QUEUE: field_async_module; the OpenACC
 * build's async(1..6), wamintgr_loki_gpu.F90:100-200): a queue is a non-blocking HIP stream, usable as the `stream` of every entry
 * point.  ecwam_hip_queue_wait_for makes work enqueued on `waiter` from now on start after everything enqueued on `waited` so far
 * (either may be NULL = the default stream); ecwam_hip_sync(ctx, queue) is WAIT_FOR_ASYNC_QUEUE. */
int ecwam_hip_queue_create(ecwam_hip_ctx *ctx, void **queue);
int ecwam_hip_queue_destroy(ecwam_hip_ctx *ctx, void *queue);
int ecwam_hip_queue_wait_for(ecwam_hip_ctx *ctx, void *waiter, void *waited);
/* Page-lock host arrays the host keeps copying to / from (the reference pins its fields when WAM_HAVE_CUDA, wvalloc.F90:49-52) */
int ecwam_hip_host_register(ecwam_hip_ctx *ctx, void *host, unsigned long long bytes);
int ecwam_hip_host_unregister(ecwam_hip_ctx *ctx, void *host);

#ifdef __cplusplus
}
#endif
#endif
