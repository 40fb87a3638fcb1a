/*
 * oracle/ora_tables.c -- TEST INFRASTRUCTURE ONLY. See ora.h header ("parity unpinned").
 * Restates the reference's table initialisers (host-side, run once):
 *   iniwcst.F90, mfredir.F90 + mfr.F90, setwavphys.F90, initmdl.F90:436-508, tabu_swellft.F90
 *   (+ kerkei.F90, kzeone.F90), init_x0tauhf.F90, initgc.F90, inisnonlin.F90 + nlweigt.F90 +
 *   jafu.F90, init_sdiss_ardh.F90, ctuwupdt.F90:93-166, userin.F90:913-976 (ice/wind thresholds).
 */
#include "ora.h"

ora_state S;

int ora_real_size(void) { return (int)sizeof(real); }

void ora_default_cfg(ora_cfg *c) {
  /* ecwam_run_model.sh:211-272 hard-wired values + mpuserin.F90:548-808 defaults (flag set A) */
  memset(c, 0, sizeof(*c));
  c->nang = 36; c->nfre = 36; c->nfre_red = 36;
  c->ifre1 = 3; c->fr1 = 4.177248E-02;
  c->idelt = 900; c->idelpro = 900; c->ximp = 1.0;
  c->iphys = 1; c->isnonlin = 0; c->irefra = 0; c->icode = 3;
  c->llgcbz0 = 0; c->llnormagam = 0; c->llcapchnk = 1;
  c->lbiwbk = 1; c->licerun = 1; c->lmaskice = 1; c->lwamrsetci = 1;
  c->lciwa1 = 0; c->lciwa2 = 0; c->lciwa3 = 0; c->lciscal = 0;
  c->lwvflx_snl = 1; c->lwflux = 0; c->lwfluxout = 1; c->lwnemocou = 0; c->lwcou = 0; c->lwcouast = 1;
  c->lwnemocouwrs = 0; c->lwnemocouibr = 0; c->lwnemotauoc = 0; c->lwnemocousend = 1; c->lwnemocoustk = 1;
  c->wspmin = -1.0;
  c->lwnemocoustrn = 0; c->zalpfacb = 1.0; c->zalpfacx = 1.0; c->zalpwrs = 1.0; c->zibrw_thrsh = 0.5; /* mpuserin.F90:780-786 */
  c->rnu = 1.5E-5; c->rnum = 0.11 * 1.5E-5;
}

/* ---- kzeone.F90:11-179 (double: JWRU) ------------------------------------------------- */
static void kzeone(double X, double Y, double *RE0, double *IM0, double *RE1, double *IM1) {
  static const double EXSQ[8] = {0.5641003087264E0, 0.4120286874989E0, 0.1584889157959E0, 0.3078003387255E-1,
                                 0.2778068842913E-2, 0.1000044412325E-3, 0.1059115547711E-5, 0.1522475804254E-8};
  static const double TSQ[8] = {0.0E0, 3.19303633920635E-1, 1.29075862295915E0, 2.95837445869665E0,
                                5.40903159724444E0, 8.80407957805676E0, 1.34685357432515E1, 2.02499163658709E1};
  double X2, Y2, R1, R2, T1, T2, P1, P2, RTERM, ITERM, L;
  int N, M, K, LL;
  R2 = X * X + Y * Y;
  if (R2 >= 1.96E2) goto L50;
  if (R2 >= 1.849E1) goto L30;
  X2 = X / 2.0; Y2 = Y / 2.0;
  P1 = X2 * X2; P2 = Y2 * Y2;
  T1 = -(log(P1 + P2) / 2.0 + 0.5772156649015329E0);
  T2 = -atan2(Y, X);
  X2 = P1 - P2; Y2 = X * Y2;
  RTERM = 1.0; ITERM = 0.0;
  *RE0 = T1; *IM0 = T2;
  T1 = T1 + 0.5;
  *RE1 = T1; *IM1 = T2;
  P2 = sqrt(R2);
  L = 2.106E0 * P2 + 4.4E0;
  if (P2 < 8.0E-1) L = 2.129E0 * P2 + 4.0E0;
  LL = (int)lround(L);
  for (N = 1; N <= LL; N++) {
    P1 = N; P2 = (double)N * N;
    R1 = RTERM;
    RTERM = (R1 * X2 - ITERM * Y2) / P2;
    ITERM = (R1 * Y2 + ITERM * X2) / P2;
    T1 = T1 + 0.5 / P1;
    *RE0 = *RE0 + T1 * RTERM - T2 * ITERM;
    *IM0 = *IM0 + T1 * ITERM + T2 * RTERM;
    P1 = P1 + 1.0;
    T1 = T1 + 0.5 / P1;
    *RE1 = *RE1 + (T1 * RTERM - T2 * ITERM) / P1;
    *IM1 = *IM1 + (T1 * ITERM + T2 * RTERM) / P1;
  }
  R1 = X / R2 - 0.5 * (X * *RE1 - Y * *IM1);
  R2 = -Y / R2 - 0.5 * (X * *IM1 + Y * *RE1);
  P1 = exp(X);
  *RE0 = P1 * *RE0; *IM0 = P1 * *IM0; *RE1 = P1 * R1; *IM1 = P1 * R2;
  return;
L30:
  X2 = 2.0 * X; Y2 = 2.0 * Y;
  R1 = Y2 * Y2;
  P1 = sqrt(X2 * X2 + R1);
  P2 = sqrt(P1 + X2);
  T1 = EXSQ[0] / (2.0 * P1);
  *RE0 = T1 * P2; *IM0 = T1 / P2; *RE1 = 0.0; *IM1 = 0.0;
  for (N = 1; N < 8; N++) {
    T2 = X2 + TSQ[N];
    P1 = sqrt(T2 * T2 + R1);
    P2 = sqrt(P1 + T2);
    T1 = EXSQ[N] / P1;
    *RE0 = *RE0 + T1 * P2;
    *IM0 = *IM0 + T1 / P2;
    T1 = EXSQ[N] * TSQ[N];
    *RE1 = *RE1 + T1 * P2;
    *IM1 = *IM1 + T1 / P2;
  }
  T2 = -Y2 * *IM0;
  *RE1 = *RE1 / R2;
  R2 = Y2 * *IM1 / R2;
  RTERM = 1.41421356237309E0 * cos(Y);
  ITERM = -1.41421356237309E0 * sin(Y);
  *IM0 = *RE0 * ITERM + T2 * RTERM;
  *RE0 = *RE0 * RTERM - T2 * ITERM;
  T1 = *RE1 * RTERM - R2 * ITERM;
  T2 = *RE1 * ITERM + R2 * RTERM;
  *RE1 = T1 * X + T2 * Y;
  *IM1 = -T1 * Y + T2 * X;
  return;
L50:
  RTERM = 1.0; ITERM = 0.0; *RE0 = 1.0; *IM0 = 0.0; *RE1 = 1.0; *IM1 = 0.0;
  P1 = 8.0 * R2;
  P2 = sqrt(R2);
  L = 3.91E0 + 8.12E1 / P2;
  LL = (int)lround(L);
  R1 = 1.0; R2 = 1.0; M = -8; K = 3;
  for (N = 1; N <= LL; N++) {
    M = M + 8; K = K - M;
    R1 = (double)(K - 4) * R1;
    R2 = (double)K * R2;
    T1 = (double)N * P1;
    T2 = RTERM;
    RTERM = (T2 * X + ITERM * Y) / T1;
    ITERM = (-T2 * Y + ITERM * X) / T1;
    *RE0 = *RE0 + R1 * RTERM; *IM0 = *IM0 + R1 * ITERM;
    *RE1 = *RE1 + R2 * RTERM; *IM1 = *IM1 + R2 * ITERM;
  }
  T1 = sqrt(P2 + X);
  T2 = -Y / T1;
  P1 = 8.86226925452758E-1 / P2;
  RTERM = P1 * cos(Y); ITERM = -P1 * sin(Y);
  R1 = *RE0 * RTERM - *IM0 * ITERM;
  R2 = *RE0 * ITERM + *IM0 * RTERM;
  *RE0 = T1 * R1 - T2 * R2; *IM0 = T1 * R2 + T2 * R1;
  R1 = *RE1 * RTERM - *IM1 * ITERM;
  R2 = *RE1 * ITERM + *IM1 * RTERM;
  *RE1 = T1 * R1 - T2 * R2; *IM1 = T1 * R2 + T2 * R1;
}

/* kerkei.F90:11-34 */
static void kerkei(real X, real *KER, real *KEI) {
  double ZR, ZI, CYR, CYI, CYR1, CYI1;
  ZR = (double)X * 0.50 * sqrt(2.0);
  ZI = ZR;
  kzeone(ZR, ZI, &CYR, &CYI, &CYR1, &CYI1);
  *KER = (real)(CYR / exp(ZR));
  *KEI = (real)(CYI / exp(ZR));
}

/* tabu_swellft.F90:47-83 */
static void tabu_swellft(void) {
  const int NITER = 100;
  const real ABMIN = C_(0.3), ABMAX = C_(8.0), KAPPA = C_(0.40);
  real DELAB, KER, KEI, ABR, ABRLOG, L10, FACT, FSUBW, FSUBWMEMO, DZETA0, DZETA0MEMO;
  DZETA0 = C_(0.0);
  DELAB = (ABMAX - ABMIN) / (real)ORA_IAB;
  L10 = LOG(C_(10.0));
  for (int I = 1; I <= ORA_IAB; I++) {
    ABRLOG = ABMIN + (real)I * DELAB;
    ABR = EXP(ABRLOG * L10);
    FACT = 1 / ABR / (C_(21.2) * KAPPA);
    FSUBW = C_(0.05);
    for (int ITER = 1; ITER <= NITER; ITER++) {
      FSUBWMEMO = FSUBW;
      DZETA0MEMO = DZETA0;
      DZETA0 = FACT * POW(FSUBW, C_(-0.5));
      kerkei(C_(2.0) * SQRT(DZETA0), &KER, &KEI);
      FSUBW = C_(0.08) / (KER * KER + KEI * KEI);
      FSUBW = C_(0.5) * (FSUBWMEMO + FSUBW);
      DZETA0 = C_(0.5) * (DZETA0MEMO + DZETA0);
    }
    S.SWELLFT[I] = FSUBW;
  }
}

/* jafu.F90:10-65 */
static int jafu(real CL, int J, int IAN) {
  int IDPH = (int)CL; /* Fortran real->integer assignment truncates */
  int JA = J + IDPH;
  if (JA <= 0) JA = IAN + JA - 1;
  if (JA >= IAN) JA = JA - IAN + 1;
  return JA;
}

/* nlweigt.F90:94-262.  Arrays indexed MFRSTLW:MLSTHG are stored with offset OFF so that
 * reference index M lives at [M+OFF]. */
#define NLOFF 8
static real FRH[32];
static int IKPx[ORA_MAXFRE + 32], IKP1x[ORA_MAXFRE + 32], IKMx[ORA_MAXFRE + 32], IKM1x[ORA_MAXFRE + 32];
static real FKLAPx[ORA_MAXFRE + 32], FKLAP1x[ORA_MAXFRE + 32], FKLAMx[ORA_MAXFRE + 32], FKLAM1x[ORA_MAXFRE + 32];
static real AF11x[ORA_MAXFRE + 32];

static void nlweigt(void) {
  const real ALAMD = C_(0.25), CON = C_(3000.0);
  const int NANG = S.NANG, NFRE = S.NFRE;
  int ISP, ISM, KLP1, IC, KH, KLH, KS, ISG, K1, K11, K2, K21, IKN, K, M;
  static int JA1[ORA_MAXANG + 2][3], JA2[ORA_MAXANG + 2][3];
  real F1P1, XF, COSTH3, DELPHI1, COSTH4, DELPHI2, CL1, CL2, CH, CL1H, CL2H, FRG, FLP, FLM, FKP, FKM, DELTHA, AL11, AL12;
  static real FRLONx[ORA_MAXFRE + 64];
#define FRLON(m) FRLONx[(m) + NLOFF]

  F1P1 = LOG10(S.FRATIO);
  ISP = (int)(LOG10(C_(1.0) + ALAMD) / F1P1 + C_(.000001));
  ISM = FLOORI(LOG10(C_(1.0) - ALAMD) / F1P1 + C_(.0000001));
  S.MFRSTLW = 1 + ISM;
  S.MLSTHG = NFRE - ISM;
  S.KFRH = -ISM + ISP + 2;

  XF = powi((C_(1.0) + ALAMD) / (C_(1.0) - ALAMD), 4);
  COSTH3 = (C_(1.0) + C_(2.0) * ALAMD + C_(2.0) * ALAMD * ALAMD * ALAMD) / ((C_(1.0) + ALAMD) * (C_(1.0) + ALAMD));
  DELPHI1 = -C_(180.0) / S.PI * ACOS(COSTH3);
  COSTH4 = SQRT(C_(1.0) - XF + XF * COSTH3 * COSTH3);
  DELPHI2 = C_(180.0) / S.PI * ACOS(COSTH4);
  DELTHA = S.DELTH * S.DEG;
  CL1 = DELPHI1 / DELTHA;
  CL2 = DELPHI2 / DELTHA;

  /* :143-159 */
  KLP1 = NANG + 1;
  IC = 1;
  for (KH = 1; KH <= 2; KH++) {
    KLH = NANG;
    if (KH == 2) KLH = KLP1;
    for (K = 1; K <= KLH; K++) {
      KS = K;
      if (KH > 1) KS = KLP1 - K + 1;
      if (KS > NANG) continue;
      CH = IC * CL1;
      JA1[KS][KH] = jafu(CH, K, KLP1);
      CH = IC * CL2;
      JA2[KS][KH] = jafu(CH, K, KLP1);
    }
    IC = -1;
  }
  /* :164-173 */
  CL1 = CL1 - (int)CL1;
  CL2 = CL2 - (int)CL2;
  S.ACL1 = FABS(CL1);
  S.ACL2 = FABS(CL2);
  S.CL11 = C_(1.0) - S.ACL1;
  S.CL21 = C_(1.0) - S.ACL2;
  AL11 = powi(C_(1.0) + ALAMD, 4);
  AL12 = powi(C_(1.0) - ALAMD, 4);
  S.DAL1 = C_(1.0) / AL11;
  S.DAL2 = C_(1.0) / AL12;
  /* :178-208 */
  ISG = 1;
  for (KH = 1; KH <= 2; KH++) {
    CL1H = ISG * CL1;
    CL2H = ISG * CL2;
    for (K = 1; K <= NANG; K++) {
      KS = K;
      if (KH == 2) KS = NANG - K + 2;
      if (K == 1) KS = 1;
      K1 = JA1[K][KH];
      S.K1W[KS - 1][KH - 1] = K1;
      if (CL1H < C_(0.0)) { K11 = K1 - 1; if (K11 < 1) K11 = NANG; }
      else { K11 = K1 + 1; if (K11 > NANG) K11 = 1; }
      S.K11W[KS - 1][KH - 1] = K11;
      K2 = JA2[K][KH];
      S.K2W[KS - 1][KH - 1] = K2;
      if (CL2H < 0) { K21 = K2 - 1; if (K21 < 1) K21 = NANG; }
      else { K21 = K2 + 1; if (K21 > NANG) K21 = 1; }
      S.K21W[KS - 1][KH - 1] = K21;
    }
    ISG = -1;
  }
  /* :213-254 */
  for (M = 1; M <= NFRE; M++) FRLON(M) = S.FR[M - 1];
  for (M = 0; M >= S.MFRSTLW; M--) FRLON(M) = FRLON(M + 1) / S.FRATIO;
  for (M = NFRE + 1; M <= NFRE + S.KFRH; M++) FRLON(M) = S.FRATIO * FRLON(M - 1);
  for (M = S.MFRSTLW; M <= S.MLSTHG; M++) {
    FRG = FRLON(M);
    AF11x[M + NLOFF] = CON * powi(FRG, 11);
    FLP = FRG * (C_(1.0) + ALAMD);
    FLM = FRG * (C_(1.0) - ALAMD);
    IKN = M + ISP;
    IKPx[M + NLOFF] = IKN;
    FKP = FRLON(IKN);
    IKP1x[M + NLOFF] = IKN + 1;
    FKLAPx[M + NLOFF] = (FLP - FKP) / (FRLON(IKN + 1) - FKP);
    FKLAP1x[M + NLOFF] = C_(1.0) - FKLAPx[M + NLOFF];
    IKN = M + ISM;
    if (IKN >= S.MFRSTLW) {
      IKMx[M + NLOFF] = IKN;
      FKM = FRLON(IKN);
      IKM1x[M + NLOFF] = IKN + 1;
      FKLAMx[M + NLOFF] = (FLM - FKM) / (FRLON(IKN + 1) - FKM);
      FKLAM1x[M + NLOFF] = C_(1.0) - FKLAMx[M + NLOFF];
    } else if (IKN + 1 == S.MFRSTLW) {
      IKMx[M + NLOFF] = 1;
      IKM1x[M + NLOFF] = S.MFRSTLW;
      FKM = FRLON(S.MFRSTLW) / S.FRATIO;
      FKLAMx[M + NLOFF] = (FLM - FKM) / (FRLON(S.MFRSTLW) - FKM);
      FKLAM1x[M + NLOFF] = C_(0.0);
    } else {
      IKMx[M + NLOFF] = 1;
      FKLAMx[M + NLOFF] = C_(0.0);
      IKM1x[M + NLOFF] = 1;
      FKLAM1x[M + NLOFF] = C_(0.0);
    }
  }
  /* :259-262 */
  for (int I = 1; I <= S.KFRH; I++) {
    M = NFRE + I - 1;
    FRH[I] = powi(FRLON(NFRE) / FRLON(M), 5);
  }
#undef FRLON
}

/* inisnonlin.F90:79-270 */
static real epmma(real X) { return EXP(-RMIN(C_(1.25) * powi(X, 4), C_(50.0))) * powi(X, 5); }

static void inisnonlin(void) {
  const int NFRE = S.NFRE;
  static real FTRFx[32];
#define FTRF(m) FTRFx[(m) + NLOFF]
  int MC, MP, MP1, MM, MM1, IC, IP, IP1, IM, IM1, ITEMP;
  real ALPH, FRR, FFACP, FFACP1, FFACM, FFACM1, FTAIL, FKLAMP, FKLAMP1, FKLAMPA, FKLAMPB, FKLAMP2, FKLAPA2, FKLAPB2;
  real FKLAP12, FKLAP22, FKLAMM, FKLAMM1, FKLAMMA, FKLAMMB, FKLAMM2, FKLAMA2, FKLAMB2, FKLAM12, FKLAM22;
  real GW1, GW2, GW3, GW4, GW5, GW6, GW7, GW8;

  nlweigt();
  ALPH = C_(1.0) / epmma(C_(1.0));
  FRR = C_(1.0);
  for (MC = 1; MC >= S.MFRSTLW; MC--) {
    FTRF(MC) = ALPH * epmma(FRR);
    FRR = FRR * S.FRATIO;
  }
  for (MC = 1; MC <= S.MLSTHG; MC++) {
    MP = IKPx[MC + NLOFF]; MP1 = IKP1x[MC + NLOFF]; MM = IKMx[MC + NLOFF]; MM1 = IKM1x[MC + NLOFF];
    S.IKP[MC - 1] = MP; S.IKP1[MC - 1] = MP1; S.IKM[MC - 1] = MM; S.IKM1[MC - 1] = MM1;
    S.FKLAP[MC - 1] = FKLAPx[MC + NLOFF]; S.FKLAP1[MC - 1] = FKLAP1x[MC + NLOFF];
    S.FKLAM[MC - 1] = FKLAMx[MC + NLOFF]; S.FKLAM1[MC - 1] = FKLAM1x[MC + NLOFF];
    S.AF11[MC - 1] = AF11x[MC + NLOFF];
    FFACP = C_(1.0); FFACP1 = C_(1.0); FFACM = C_(1.0); FFACM1 = C_(1.0); FTAIL = C_(1.0);
    IC = MC;
    if (IC < 1) IC = 1;
    IP = MP; IP1 = MP1; IM = MM; IM1 = MM1;
    if (IP < 1) { FFACP = FTRF(IP); IP = 1; }
    if (IP1 < 1) { FFACP1 = FTRF(IP1); IP1 = 1; }
    if (IM < S.MFRSTLW) { FFACM = C_(0.0); IM = 1; }
    else if (IM < 1) { FFACM = FTRF(IM); IM = 1; }
    if (IM1 < S.MFRSTLW) { FFACM1 = C_(0.0); IM1 = 1; }
    else if (IM1 < 1) { FFACM1 = FTRF(IM1); IM1 = 1; }
    if (IP1 > NFRE) {
      ITEMP = IP1 - NFRE + 1;
      if (ITEMP > S.KFRH) ITEMP = S.KFRH;
      FFACP1 = FRH[ITEMP];
      IP1 = NFRE;
      if (IP > NFRE) {
        FFACP = FRH[IP - NFRE + 1];
        IP = NFRE;
        if (IC > NFRE) {
          FTAIL = FRH[IC - NFRE + 1];
          IC = NFRE;
          if (IM1 > NFRE) { FFACM1 = FRH[IM1 - NFRE + 1]; IM1 = NFRE; }
        }
      }
    }
    S.INLCOEF[MC - 1][0] = IC; S.INLCOEF[MC - 1][1] = IP; S.INLCOEF[MC - 1][2] = IP1;
    S.INLCOEF[MC - 1][3] = IM; S.INLCOEF[MC - 1][4] = IM1;

    FKLAMP = S.FKLAP[MC - 1]; FKLAMP1 = S.FKLAP1[MC - 1];
    GW2 = FKLAMP1 * FFACP * S.DAL1; GW1 = GW2 * S.CL11; GW2 = GW2 * S.ACL1;
    GW4 = FKLAMP * FFACP1 * S.DAL1; GW3 = GW4 * S.CL11; GW4 = GW4 * S.ACL1;
    FKLAMPA = FKLAMP * S.CL11; FKLAMPB = FKLAMP * S.ACL1;
    FKLAMP2 = FKLAMP1 * S.ACL1; FKLAMP1 = FKLAMP1 * S.CL11;
    FKLAPA2 = FKLAMPA * FKLAMPA; FKLAPB2 = FKLAMPB * FKLAMPB;
    FKLAP12 = FKLAMP1 * FKLAMP1; FKLAP22 = FKLAMP2 * FKLAMP2;
    real *R = S.RNLCOEF[MC - 1];
    R[0] = FTAIL; R[1] = GW1; R[2] = GW2; R[3] = GW3; R[4] = GW4;
    R[5] = FKLAMPA; R[6] = FKLAMPB; R[7] = FKLAMP2; R[8] = FKLAMP1;
    R[9] = FKLAPA2; R[10] = FKLAPB2; R[11] = FKLAP12; R[12] = FKLAP22;
    FKLAMM = S.FKLAM[MC - 1]; FKLAMM1 = S.FKLAM1[MC - 1];
    GW6 = FKLAMM1 * FFACM * S.DAL2; GW5 = GW6 * S.CL21; GW6 = GW6 * S.ACL2;
    GW8 = FKLAMM * FFACM1 * S.DAL2; GW7 = GW8 * S.CL21; GW8 = GW8 * S.ACL2;
    FKLAMMA = FKLAMM * S.CL21; FKLAMMB = FKLAMM * S.ACL2;
    FKLAMM2 = FKLAMM1 * S.ACL2; FKLAMM1 = FKLAMM1 * S.CL21;
    FKLAMA2 = FKLAMMA * FKLAMMA; FKLAMB2 = FKLAMMB * FKLAMMB;
    FKLAM12 = FKLAMM1 * FKLAMM1; FKLAM22 = FKLAMM2 * FKLAMM2;
    R[13] = GW5; R[14] = GW6; R[15] = GW7; R[16] = GW8;
    R[17] = FKLAMMA; R[18] = FKLAMMB; R[19] = FKLAMM2; R[20] = FKLAMM1;
    R[21] = FKLAMA2; R[22] = FKLAMB2; R[23] = FKLAM12; R[24] = FKLAM22;
  }
#undef FTRF
}

/* init_sdiss_ardh.F90:67-96 */
static void init_sdiss_ardh(void) {
  const int NANG = S.NANG;
  int NANGD = NANG / 2;
  real DELTH_TRUNC, DELTH_LOC;
  S.NSDSNTH = NINT(S.ISDSDTH * S.RAD / S.DELTH);
  if (S.NSDSNTH > NANGD - 1) S.NSDSNTH = NANGD - 1;
  DELTH_TRUNC = (S.TH[0] + S.ISDSDTH * S.RAD) - (S.TH[S.NSDSNTH] - C_(0.5) * S.DELTH);
  DELTH_TRUNC = RMAX(C_(0.0), RMIN(DELTH_TRUNC, S.DELTH));
  for (int K = 1; K <= NANG; K++) {
    for (int I_INT = K - S.NSDSNTH; I_INT <= K + S.NSDSNTH; I_INT++) {
      int J_INT = I_INT;
      if (I_INT < 1) J_INT = I_INT + NANG;
      if (I_INT > NANG) J_INT = I_INT - NANG;
      int idx = I_INT - (K - S.NSDSNTH);
      S.INDICESSAT[K - 1][idx] = J_INT - 1;
      if (I_INT == K - S.NSDSNTH || I_INT == K + S.NSDSNTH) DELTH_LOC = DELTH_TRUNC;
      else DELTH_LOC = S.DELTH;
      real cs = COS(S.TH[K - 1] - S.TH[J_INT - 1]);
      S.SATWEIGHTS[K - 1][idx] = DELTH_LOC * cs * cs; /* **ISB, ISB=2 */
    }
  }
}

/* initgc.F90:65-109, gc_dispersion.h */
static real fomeg_gc(real x) { return SQRT(S.G * x + S.SURFT * x * x * x); }
static real fvg_gc(real x) { return C_(0.5) / fomeg_gc(x) * (S.G + C_(3.0) * S.SURFT * x * x); }
static real fc_gc(real x) { return fomeg_gc(x) / x; }

static void initgc(void) {
  const real KRATIO_GC = C_(1.2), XKS_GC = C_(0.006), XKL_GC = C_(20000.0);
  S.XLOGKRATIOM1_GC = C_(1.0) / LOG(KRATIO_GC);
  S.SQRTGOSURFT = SQRT(S.G / S.SURFT);
  S.NWAV_GC = NINT(LOG(XKL_GC / XKS_GC) / (LOG(KRATIO_GC)));
  int N = S.NWAV_GC;
  for (int I = 1; I <= N; I++) {
    S.XK_GC[I] = XKS_GC * powi(KRATIO_GC, I - 1);
    S.XKM_GC[I] = C_(1.0) / S.XK_GC[I];
    S.OMEGA_GC[I] = fomeg_gc(S.XK_GC[I]);
    S.OMXKM3_GC[I] = S.OMEGA_GC[I] * S.XKM_GC[I] * S.XKM_GC[I] * S.XKM_GC[I];
    S.VG_GC[I] = fvg_gc(S.XK_GC[I]);
    S.C_GC[I] = fc_gc(S.XK_GC[I]);
    S.CM_GC[I] = C_(1.0) / S.C_GC[I];
    S.C2OSQRTVG_GC[I] = S.C_GC[I] * S.C_GC[I] / SQRT(S.VG_GC[I]);
    S.XKMSQRTVGOC2_GC[I] = S.XKM_GC[I] / S.C2OSQRTVG_GC[I];
    S.OM3GMKM_GC[I] = S.OMEGA_GC[I] * S.OMEGA_GC[I] * S.OMEGA_GC[I] / (S.G * S.XK_GC[I]);
  }
  S.DELKCC_GC[1] = (real)(0.5 * (S.XK_GC[2] - S.XK_GC[1])) / S.C2OSQRTVG_GC[1]; /* 0.5 default-kind literal, initgc.F90:98 */
  S.DELKCC_GC_NS[1] = S.DELKCC_GC[1];
  for (int I = 2; I <= N - 1; I++) {
    S.DELKCC_GC[I] = C_(0.5) * (S.XK_GC[I + 1] - S.XK_GC[I - 1]) / S.C2OSQRTVG_GC[I];
    S.DELKCC_GC_NS[I] = C_(0.5) * (S.XK_GC[I + 1] - S.XK_GC[I]) / S.C2OSQRTVG_GC[I];
  }
  S.DELKCC_GC[N] = C_(0.5) * (S.XK_GC[N] - S.XK_GC[N - 1]) / S.C2OSQRTVG_GC[N];
  S.DELKCC_GC_NS[N] = S.DELKCC_GC[N];
  for (int I = 1; I <= N; I++) S.DELKCC_OMXKM3_GC[I] = S.DELKCC_GC[I] * S.OMXKM3_GC[I];
}

int ora_init(const ora_cfg *c) {
  memset(&S, 0, sizeof(S));
  S.c = *c;
  const int NANG = c->nang, NFRE = c->nfre;
  if (NANG > ORA_MAXANG - 1 || NFRE > ORA_MAXFRE - 1 || NANG < 4 || NFRE < 8) return 1;
  S.NANG = NANG; S.NFRE = NFRE;
  S.NFRE_RED = (c->nfre_red <= 0) ? NFRE : c->nfre_red; /* mpuserin.F90:872 */
  if (S.NFRE_RED > NFRE) return 1;

  /* yowpcons.F90:19-66 + iniwcst.F90:54-69 */
  S.G = C_(9.806); S.GM1 = C_(0.101978381);
  S.PI = C_(4.0) * ATAN(C_(1.0));
  S.ZPI = C_(2.0) * S.PI;
  S.ZPI4GM1 = powi(S.ZPI, 4) / S.G;
  S.ZPI4GM2 = powi(S.ZPI, 4) / (S.G * S.G);
  S.RAD = S.PI / C_(180.0);
  S.DEG = (real)(180.) / S.PI; /* iniwcst.F90:63 */
  S.CIRC = C_(40007993.95);
  S.R = S.CIRC / S.ZPI * C_(1.0);
  S.EPSMIN = C_(0.1E-32);
  S.ROWATER = C_(1000.0); S.ROWATERM1 = C_(1.0) / S.ROWATER; S.ROAIR = C_(1.225);
  S.GAM_SURF = C_(0.0717); S.SURFT = S.GAM_SURF / S.ROWATER;
  S.EPSUS = C_(1.0E-6); S.EPSU10 = SQRT(C_(1.0E-3));
  S.ACD = C_(8.0E-4); S.BCD = C_(8.0E-5); S.ACDLIN = C_(0.0008); S.BCDLIN = C_(0.00047); S.CDMAX = C_(0.0025);
  S.TAUOCMIN = C_(0.01); S.TAUOCMAX = C_(50.0); S.PHIEPSMIN = C_(-3276.80); S.PHIEPSMAX = C_(-0.05);
  S.WSEMEAN_MIN = C_(0.001);

  /* yowfred.F90:50-82 */
  S.FRATIO = C_(1.1); S.WETAIL = C_(0.25); S.FRTAIL = C_(0.2); S.WP1TAIL = C_(1.0) / C_(3.0);
  S.COEF4 = C_(5.0E-07); S.FRIC = C_(28.0);

  /* mfr.F90:42-48 */
  S.FR[c->ifre1 - 1] = (real)c->fr1;
  for (int M = c->ifre1 - 1; M >= 1; M--) S.FR[M - 1] = S.FR[M] / S.FRATIO;
  for (int M = c->ifre1 + 1; M <= NFRE; M++) S.FR[M - 1] = S.FRATIO * S.FR[M - 2];
  /* mfredir.F90:112-129 */
  S.DELTH = S.ZPI / (real)NANG;
  for (int K = 1; K <= NANG; K++) {
    S.TH[K - 1] = (real)(K - 1) * S.DELTH + C_(0.5) * S.DELTH;
    S.COSTH[K - 1] = COS(S.TH[K - 1]);
    S.SINTH[K - 1] = SIN(S.TH[K - 1]);
  }
  real CO1 = C_(0.5) * (S.FRATIO - C_(1.0)) * S.DELTH;
  S.DFIM[0] = CO1 * S.FR[0];
  for (int M = 2; M <= NFRE - 1; M++) S.DFIM[M - 1] = CO1 * (S.FR[M - 1] + S.FR[M - 2]);
  S.DFIM[NFRE - 1] = CO1 * S.FR[NFRE - 2];

  /* yowphys.F90 PARAMETERs */
  S.XKAPPA = C_(0.40); S.XNLEV = C_(10.0); S.ALPHAMAX = C_(0.11);
  S.SWELLF = C_(0.66); S.SWELLF2 = C_(-0.018); S.SWELLF3 = C_(0.022); S.SWELLF5 = C_(1.2); S.SWELLF6 = C_(1.0);
  S.ABMIN = C_(0.3); S.ABMAX = C_(8.0);
  S.SDSBR = C_(9.0E-4); S.ISDSDTH = 80; S.ISB = 2; S.IPSAT = 2;
  S.SSDSC2 = C_(-2.2E-5); S.SSDSC4 = C_(1.0); S.SSDSC6 = C_(0.3); S.MICHE = C_(1.0); S.SSDSC3 = C_(0.0);
  S.RNU = (real)c->rnu; S.RNUM = (real)c->rnum;

  S.IDAMPING = 1; /* mpuserin.F90:609 */
  S.CDIS = C_(0.0); S.DELTA_SDIS = C_(0.0); S.CDISVIS = C_(0.0);
  if (c->iphys != 0 && c->iphys != 1) return 2;
  /* setwavphys.F90:115-202 (IPHYS == 1), :46-112 (IPHYS == 0, applied below) */
  S.ZALP = C_(0.008); S.TAILFACTOR = C_(2.5); S.TAILFACTOR_PM = C_(3.0);
  if (NANG <= 24) { S.ANG_GC_A = C_(0.40); S.ANG_GC_B = C_(0.60); S.ANG_GC_C = C_(3.0); }
  else { S.ANG_GC_A = C_(0.35); S.ANG_GC_B = C_(0.65); S.ANG_GC_C = C_(3.0); }
  S.RN1_RN = C_(0.25);
  if (c->llgcbz0) {
    S.ALPHA = C_(0.0055); S.ALPHAMIN = C_(0.0001); S.CHNKMIN_U = C_(28.); S.ALPHAPMAX = C_(0.03);
    S.DELTA_THETA_RN = C_(0.75); S.DTHRN_A = C_(0.60); S.DTHRN_U = C_(33.0);
    S.Z0TUBMAX = C_(0.05); S.Z0RAT = C_(0.02); S.SWELLF4 = C_(1.15E05); S.SWELLF7 = C_(4.32E05);
    S.SWELLF7M1 = C_(1.0) / S.SWELLF7; S.SSDSC5 = C_(0.0);
    if (c->llnormagam) { S.BETAMAX = C_(1.39); S.TAUWSHELTER = C_(0.0); }
    else { S.BETAMAX = C_(1.44); S.TAUWSHELTER = C_(0.25); }
  } else {
    S.ALPHA = C_(0.0065); S.ALPHAPMAX = C_(0.031);
    S.DELTA_THETA_RN = C_(0.75); S.DTHRN_A = C_(0.60); S.DTHRN_U = C_(200.0);
    S.Z0TUBMAX = C_(0.0005); S.Z0RAT = C_(0.04); S.SWELLF4 = C_(1.5E05); S.SWELLF7 = C_(3.6E05);
    S.SWELLF7M1 = C_(1.0) / S.SWELLF7; S.SSDSC5 = C_(0.0);
    if (c->llnormagam) { S.BETAMAX = C_(1.39); S.TAUWSHELTER = C_(0.0); S.ALPHAMIN = C_(0.0005); S.CHNKMIN_U = C_(30.); }
    else { S.BETAMAX = C_(1.40); S.TAUWSHELTER = C_(0.25); S.ALPHAMIN = C_(0.0001); S.CHNKMIN_U = C_(33.); }
  }
  S.EGRCRV = C_(1065.0); S.AFCRV = C_(2.453E-4); S.BFCRV = C_(-3.1236);
  if (c->iphys == 0) { /* Janssen wind input + WAM cycle 4 dissipation, setwavphys.F90:46-112 */
    S.ZALP = C_(0.008); S.TAILFACTOR = C_(2.5); S.ALPHAMIN = C_(0.0001); S.ALPHAPMAX = C_(0.03); S.TAUWSHELTER = C_(0.0);
    S.DELTA_THETA_RN = C_(0.75); S.DTHRN_A = C_(0.80); S.DTHRN_U = C_(33.0); S.RN1_RN = C_(0.25); S.TAILFACTOR_PM = C_(0.0);
    if (c->llgcbz0) {
      S.ALPHA = C_(0.0055); S.CHNKMIN_U = C_(28.);
      S.BETAMAX = c->llnormagam ? C_(1.32) : C_(1.25);
      S.CDIS = C_(-1.3); S.DELTA_SDIS = C_(0.6); S.CDISVIS = C_(-4.0);
    } else {
      S.ALPHA = C_(0.0065); S.CHNKMIN_U = C_(33.); S.BETAMAX = C_(1.20);
      S.CDIS = C_(-1.33); S.DELTA_SDIS = C_(0.5); S.CDISVIS = C_(0.0);
    }
    S.EGRCRV = C_(1108.0); S.AFCRV = C_(4.0E-4); S.BFCRV = C_(-3.0);
  }

  /* initmdl.F90:436-508 */
  for (int M = 1; M <= NFRE; M++) {
    S.DFIMOFR[M - 1] = S.DFIM[M - 1] / S.FR[M - 1];
    S.DFIMFR[M - 1] = S.DFIM[M - 1] * S.FR[M - 1];
    S.ZPIFR[M - 1] = S.ZPI * S.FR[M - 1];
    S.FR5[M - 1] = powi(S.FR[M - 1], 5);
    S.COFRM4[M - 1] = S.COEF4 * S.G / powi(S.FR[M - 1], 4);
    S.FLMAX[M - 1] = (S.ALPHAPMAX / S.PI) / (S.ZPI4GM2 * S.FR5[M - 1]);
  }
  S.FLOGSPRDM1 = C_(1.0) / LOG10(S.FRATIO);
  real XLOGFRATIO = LOG(S.FRATIO);
  S.RHOWG_DFIM[0] = C_(0.5) * S.ROWATER * S.G * S.DELTH * XLOGFRATIO * S.FR[0];
  for (int M = 2; M <= NFRE - 1; M++) S.RHOWG_DFIM[M - 1] = S.ROWATER * S.G * S.DELTH * XLOGFRATIO * S.FR[M - 1];
  S.RHOWG_DFIM[NFRE - 1] = C_(0.5) * S.ROWATER * S.G * S.DELTH * XLOGFRATIO * S.FR[NFRE - 1];
  S.NFRE_ODD = NFRE - 1 + (NFRE % 2);
  S.DFIM_SIM[NFRE - 1] = C_(0.0);
  S.DFIM_SIM[0] = S.DELTH * XLOGFRATIO * S.FR[0] / C_(3.0);
  for (int M = 2; M <= S.NFRE_ODD - 1; M += 2) {
    S.DFIM_SIM[M - 1] = C_(4.0) * S.DELTH * XLOGFRATIO * S.FR[M - 1] / C_(3.0);
    S.DFIM_SIM[M] = C_(2.0) * S.DELTH * XLOGFRATIO * S.FR[M] / C_(3.0);
  }
  S.DFIM_SIM[S.NFRE_ODD - 1] = S.DELTH * XLOGFRATIO * S.FR[S.NFRE_ODD - 1] / C_(3.0);

  tabu_swellft();

  /* init_x0tauhf.F90:65-100 */
  S.BETAMAXOXKAPPA2 = S.BETAMAX / (S.XKAPPA * S.XKAPPA);
  S.BMAXOKAP = S.DELTA_THETA_RN * S.BETAMAXOXKAPPA2 / S.XKAPPA;
  S.BMAXOKAPDTH = S.BMAXOKAP * S.DELTH;
  S.GAMNCONST = S.BMAXOKAP * C_(0.5) * powi(S.ZPI, 4) * S.GM1 * S.GM1 * S.GM1;
  {
    real ALPH = (c->llgcbz0 || c->llcapchnk || c->llnormagam) ? S.ALPHAMIN : S.ALPHA;
    real X0 = C_(0.005), FF, F, DF;
    for (int J = 1; J <= 30; J++) {
      FF = EXP(S.XKAPPA / (X0 + S.ZALP));
      F = ALPH * X0 * X0 * FF - C_(1.0);
      if (F == C_(0.0)) break;
      real q = X0 / (X0 + S.ZALP);
      DF = ALPH * FF * (C_(2.0) * X0 - S.XKAPPA * q * q);
      X0 = X0 - F / DF;
    }
    S.X0TAUHF = X0;
    real CONST1 = S.BETAMAXOXKAPPA2 / C_(3.0);
    S.WTAUHF[0] = CONST1;
    for (int J = 2; J <= ORA_JTOT_TAUHF - 1; J += 2) {
      S.WTAUHF[J - 1] = C_(4.0) * CONST1;
      S.WTAUHF[J] = C_(2.0) * CONST1;
    }
    S.WTAUHF[ORA_JTOT_TAUHF - 1] = CONST1;
  }
  S.EPS1 = C_(0.00001);

  initgc();
  inisnonlin();
  init_sdiss_ardh();

  /* userin.F90:913-918, 957-976; yowice.F90:22; yowshal.F90:21-22; yowwind.F90:19 */
  S.WSPMIN = (c->wspmin > 0) ? (real)c->wspmin : (c->llgcbz0 ? C_(0.3) : C_(1.0));
  S.FLMIN = C_(0.00001);
  if (c->lmaskice) { S.CITHRSH = C_(0.3); S.CIBLOCK = C_(0.0); S.CITHRSH_TAIL = S.CITHRSH; S.CDICWA = C_(0.0); }
  else { S.CITHRSH = C_(1.0); S.CIBLOCK = C_(1.0); S.CITHRSH_TAIL = C_(0.1); S.CDICWA = c->lciwa2 ? C_(0.01) : C_(0.0); } /* userin.F90:971-977 */
  S.ZALPFACX = (real)S.c.zalpfacx; S.ZALPFACB = (real)S.c.zalpfacb; S.ZALPWRS = (real)S.c.zalpwrs; S.ZIBRW_THRSH = (real)S.c.zibrw_thrsh;
  S.GAM_B_J = C_(0.8); S.BATHYMAX = C_(998.999); S.WSPMIN_RESET_TAUW = C_(4.0);

  /* ctuwupdt.F90:97-161 */
  for (int K = 1; K <= NANG; K++) {
    int KM1 = K - 1; if (KM1 < 1) KM1 = NANG;
    int KP1 = K + 1; if (KP1 > NANG) KP1 = 1;
    S.KPM[K - 1][0] = KM1; S.KPM[K - 1][1] = K; S.KPM[K - 1][2] = KP1;
    int *jx = S.JXO[K - 1], *jy = S.JYO[K - 1], *kc = S.KCR[K - 1];
    if (S.COSTH[K - 1] >= C_(0.0)) {
      jy[0] = 1; jy[1] = 2;
      if (S.SINTH[K - 1] >= C_(0.0)) { jx[0] = 1; jx[1] = 2; kc[0] = 3; kc[1] = 2; kc[2] = 4; kc[3] = 1; }
      else { jx[0] = 2; jx[1] = 1; kc[0] = 2; kc[1] = 3; kc[2] = 1; kc[3] = 4; }
    } else {
      jy[0] = 2; jy[1] = 1;
      if (S.SINTH[K - 1] >= C_(0.0)) { jx[0] = 1; jx[1] = 2; kc[0] = 4; kc[1] = 1; kc[2] = 3; kc[3] = 2; }
      else { jx[0] = 2; jx[1] = 1; kc[0] = 1; kc[1] = 4; kc[2] = 2; kc[3] = 3; }
    }
  }
  return 0;
}

/* ---- table export for cross-checks (numpy side reads by name) --------------------------- */
#define EXPORT_ARR(nm, arr, n) if (!strcmp(name, nm)) { for (int i = 0; i < (n) && i < cap; i++) out[i] = (double)(arr)[i]; return (n); }
#define EXPORT_SC(nm, v) if (!strcmp(name, nm)) { if (cap > 0) out[0] = (double)(v); return 1; }

int ora_get(const char *name, double *out, int cap) {
  const int NANG = S.NANG, NFRE = S.NFRE;
  EXPORT_ARR("FR", S.FR, NFRE) EXPORT_ARR("DFIM", S.DFIM, NFRE) EXPORT_ARR("DFIMOFR", S.DFIMOFR, NFRE)
  EXPORT_ARR("DFIMFR", S.DFIMFR, NFRE) EXPORT_ARR("DFIM_SIM", S.DFIM_SIM, NFRE) EXPORT_ARR("RHOWG_DFIM", S.RHOWG_DFIM, NFRE)
  EXPORT_ARR("ZPIFR", S.ZPIFR, NFRE) EXPORT_ARR("FR5", S.FR5, NFRE) EXPORT_ARR("COFRM4", S.COFRM4, NFRE)
  EXPORT_ARR("FLMAX", S.FLMAX, NFRE) EXPORT_ARR("TH", S.TH, NANG) EXPORT_ARR("COSTH", S.COSTH, NANG)
  EXPORT_ARR("SINTH", S.SINTH, NANG) EXPORT_ARR("WTAUHF", S.WTAUHF, ORA_JTOT_TAUHF)
  EXPORT_ARR("SWELLFT", (S.SWELLFT + 1), ORA_IAB) EXPORT_ARR("AF11", S.AF11, S.MLSTHG)
  EXPORT_ARR("XK_GC", (S.XK_GC + 1), S.NWAV_GC) EXPORT_ARR("OMEGA_GC", (S.OMEGA_GC + 1), S.NWAV_GC)
  EXPORT_ARR("OMXKM3_GC", (S.OMXKM3_GC + 1), S.NWAV_GC) EXPORT_ARR("CM_GC", (S.CM_GC + 1), S.NWAV_GC)
  EXPORT_ARR("C2OSQRTVG_GC", (S.C2OSQRTVG_GC + 1), S.NWAV_GC) EXPORT_ARR("XKMSQRTVGOC2_GC", (S.XKMSQRTVGOC2_GC + 1), S.NWAV_GC)
  EXPORT_ARR("OM3GMKM_GC", (S.OM3GMKM_GC + 1), S.NWAV_GC) EXPORT_ARR("DELKCC_GC_NS", (S.DELKCC_GC_NS + 1), S.NWAV_GC)
  EXPORT_ARR("DELKCC_OMXKM3_GC", (S.DELKCC_OMXKM3_GC + 1), S.NWAV_GC) EXPORT_ARR("XKM_GC", (S.XKM_GC + 1), S.NWAV_GC)
  EXPORT_SC("DELTH", S.DELTH) EXPORT_SC("X0TAUHF", S.X0TAUHF) EXPORT_SC("NFRE_ODD", S.NFRE_ODD)
  EXPORT_SC("MFRSTLW", S.MFRSTLW) EXPORT_SC("MLSTHG", S.MLSTHG) EXPORT_SC("KFRH", S.KFRH)
  EXPORT_SC("DAL1", S.DAL1) EXPORT_SC("DAL2", S.DAL2) EXPORT_SC("NSDSNTH", S.NSDSNTH) EXPORT_SC("NWAV_GC", S.NWAV_GC)
  EXPORT_SC("BETAMAXOXKAPPA2", S.BETAMAXOXKAPPA2) EXPORT_SC("TAUWSHELTER", S.TAUWSHELTER) EXPORT_SC("FLOGSPRDM1", S.FLOGSPRDM1)
  EXPORT_SC("GAMNCONST", S.GAMNCONST) EXPORT_SC("BMAXOKAP", S.BMAXOKAP) EXPORT_SC("SQRTGOSURFT", S.SQRTGOSURFT)
  EXPORT_SC("XLOGKRATIOM1_GC", S.XLOGKRATIOM1_GC) EXPORT_SC("WSPMIN", S.WSPMIN)
  if (!strcmp(name, "INLCOEF")) { int n = S.MLSTHG * 5; for (int i = 0; i < n && i < cap; i++) out[i] = S.INLCOEF[i / 5][i % 5]; return n; }
  if (!strcmp(name, "RNLCOEF")) { int n = S.MLSTHG * 25; for (int i = 0; i < n && i < cap; i++) out[i] = (double)S.RNLCOEF[i / 25][i % 25]; return n; }
  if (!strcmp(name, "IKP")) { for (int i = 0; i < S.MLSTHG && i < cap; i++) out[i] = S.IKP[i]; return S.MLSTHG; }
  if (!strcmp(name, "IKP1")) { for (int i = 0; i < S.MLSTHG && i < cap; i++) out[i] = S.IKP1[i]; return S.MLSTHG; }
  if (!strcmp(name, "IKM")) { for (int i = 0; i < S.MLSTHG && i < cap; i++) out[i] = S.IKM[i]; return S.MLSTHG; }
  if (!strcmp(name, "IKM1")) { for (int i = 0; i < S.MLSTHG && i < cap; i++) out[i] = S.IKM1[i]; return S.MLSTHG; }
#define EXPORT_KW(nm, A) if (!strcmp(name, nm)) { int n = NANG * 2; for (int i = 0; i < n && i < cap; i++) out[i] = S.A[i / 2][i % 2]; return n; }
  EXPORT_KW("K1W", K1W) EXPORT_KW("K2W", K2W) EXPORT_KW("K11W", K11W) EXPORT_KW("K21W", K21W)
  EXPORT_KW("JXO", JXO) EXPORT_KW("JYO", JYO)
  if (!strcmp(name, "KCR")) { int n = NANG * 4; for (int i = 0; i < n && i < cap; i++) out[i] = S.KCR[i / 4][i % 4]; return n; }
  if (!strcmp(name, "KPM")) { int n = NANG * 3; for (int i = 0; i < n && i < cap; i++) out[i] = S.KPM[i / 3][i % 3]; return n; }
  if (!strcmp(name, "INDICESSAT")) { int w = 2 * S.NSDSNTH + 1, n = NANG * w; for (int i = 0; i < n && i < cap; i++) out[i] = S.INDICESSAT[i / w][i % w]; return n; }
  if (!strcmp(name, "SATWEIGHTS")) { int w = 2 * S.NSDSNTH + 1, n = NANG * w; for (int i = 0; i < n && i < cap; i++) out[i] = (double)S.SATWEIGHTS[i / w][i % w]; return n; }
  return -1;
}

/* cigetdeac.F90:64-82, 553-559: the SDICE1 table from its tabulated block raw[36][11] = CIDEAC(6:16, IH) (data file of the
 * product, ecwam_amd/data/cideac_kohout_meylan.txt, passed in by the test harness) */
void ora_set_cideac(const double *raw) {
  S.NICH = 36; S.DHIC = C_(0.1);
  S.NICT = 16; S.TICMIN = C_(1.0); S.DTIC = C_(1.0);
  S.HICMIN = C_(0.2); /* yowice.F90:23 */
  for (int IH = 1; IH <= S.NICH; IH++)
    for (int IT = 6; IT <= 16; IT++) S.CIDEAC[IT - 1][IH - 1] = (real)raw[(IH - 1) * 11 + (IT - 6)];
  S.CIDEAC[0][0] = C_(-2.00);
  S.CIDEAC[0][S.NICH - 1] = C_(-1.00);
  real DHI = S.CIDEAC[0][S.NICH - 1] - S.CIDEAC[0][0];
  for (int IH = 2; IH <= S.NICH - 1; IH++) S.CIDEAC[0][IH - 1] = S.CIDEAC[0][0] + (IH - 1) * DHI / (S.NICH - 1);
  for (int IH = 1; IH <= S.NICH; IH++) {
    real DCI = S.CIDEAC[5][IH - 1] - S.CIDEAC[0][IH - 1];
    for (int IT = 2; IT <= 5; IT++) S.CIDEAC[IT - 1][IH - 1] = S.CIDEAC[0][IH - 1] + DCI * (IT - 1) * S.DTIC / (5 * S.DTIC);
  }
}
