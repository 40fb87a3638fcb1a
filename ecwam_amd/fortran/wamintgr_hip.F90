! wamintgr_hip.F90 -- third variant of the reference's time-step driver next to WAMINTGR (wamintgr.F90:10-13) and
! WAMINTGR_LOKI_GPU (wamintgr_loki_gpu.F90:10-13): identical 14-argument interface, selected at the seam
! wamodel.F90:289-299.  Spectra, forcing and integrated parameters stay resident on the GPU between calls (as in the OpenACC
! variant, GET_DEVICE_DATA_RDWR, wamintgr_loki_gpu.F90:100-105).  The host types carry FIELD_API's method surface and status tracking
! (yowdrvtype_hip.F90): the driver makes the reference's own GET_DEVICE_DATA_* calls (wamintgr_loki_gpu.F90:100-105,121-127,141-157), which
! copy a member to the device only when the host holds the newer copy -- the first call, a new FF_NEXT, fields the host changed and
! said so (GET_HOST_DATA_RDWR / SYNC_DEVICE_*) -- and mark what the kernels write as newer on the device; nothing comes back unless the
! host asks (SYNC_HOST_* / GET_HOST_DATA_*: output / restart / coupling steps, wamodel.F90:376-385,435-470,614-642), or HIP_LSYNC_EAGER
! posts the reference's per-step SYNC_HOST_RDONLY calls (wamintgr_loki_gpu.F90:197-200).
!
! Standalone build (this repository): dates and steps that the reference takes from YOWSTAT/YOWWIND are module
! variables of ECWAM_HIP_DRV, and INCDATE is supplied by the caller.  In-tree build: see INTEGRATION.md.
MODULE ECWAM_HIP_DRV
  USE ECWAM_HIP_MOD, ONLY : JWIM
  IMPLICIT NONE
  CHARACTER(LEN=14) :: HIP_CDTPRO = ' '        ! YOWSTAT:CDTPRO  end date of the current propagation step
  INTEGER(KIND=JWIM) :: HIP_IDELT = 900        ! YOWSTAT:IDELT
  INTEGER(KIND=JWIM) :: HIP_IDELWO = 3600      ! YOWSTAT:IDELWO  wind output step (newwind.F90:172)
  ! the wind-file date counters of YOWWIND / YOWSTAT that NEWWIND and WAMINTGR advance (newwind.F90:110-112, wamintgr.F90:164-174)
  CHARACTER(LEN=14) :: HIP_CDATEWO = ' '       ! YOWWIND:CDATEWO  date of the winds in use
  CHARACTER(LEN=14) :: HIP_CDAWIFL = ' '       ! YOWWIND:CDAWIFL  date of the next wind file
  CHARACTER(LEN=14) :: HIP_CDATEFL = '99999999999999'   ! YOWWIND:CDATEFL  date from which the next wind file is used
  INTEGER(KIND=JWIM) :: HIP_IDELWI = 3600      ! YOWSTAT:IDELWI  wind input step
  INTEGER(KIND=JWIM) :: HIP_IDELPRO = 900      ! YOWSTAT:IDELPRO
  LOGICAL :: HIP_LLNEWFILE = .FALSE.           ! WAMINTGR's saved LLNEWFILE (wamintgr.F90:83-85)
  LOGICAL :: HIP_LLSOURCE = .TRUE.             ! YOWSTAT:LLSOURCE
  INTEGER(KIND=JWIM) :: HIP_NEMONTAU = 0       ! YOWCOUP:NEMONTAU  accumulation count of the WAVE2OCEAN stresses: advanced after every IMPLSCH
                                               ! call when LWNEMOCOU (wamintgr.F90:150); UPDNEMOSTRESS divides by it and resets it (updnemostress.F90:84-123)
  INTEGER(KIND=JWIM) :: HIP_ICODE_WND = 0      ! NEWWIND's ICODE_WND: ICODE_CPL when LWCOU (newwind.F90:120-124); 0 = ICODE of the set-up
  ! the 1:1 step as ONE kernel where a build covers the configuration (ecwam_hip_propags2_implsch: PROPAGS2 inside IMPLSCH's tile load, the same
  ! bits as the two kernels); .FALSE. keeps PROPAGS2 and IMPLSCH apart
  LOGICAL :: HIP_LFUSED_STEP = .TRUE.
  LOGICAL :: HIP_LSYNC_EAGER = .FALSE.         ! end every source-term step with the reference's asynchronous copies (wamintgr_loki_gpu.F90:197-200:
                                               ! FL1 + FF_NOW on queue 4, WVENVI on 5, WAM2NEMO on 6): 2.2 GB per step at O320 -- off by default,
                                               ! the host's GET_HOST_DATA_* at its output steps fetch what is missing
  LOGICAL :: HIP_LSYNC_SPECTRA = .FALSE.       ! (shorthand) FL1 / XLLWS back in the host arrays when the call returns
  LOGICAL :: HIP_LSYNC_FIELDS = .FALSE.        ! (shorthand) FF_NOW / INTFLDS / MIJ back in the host types when the call returns
  LOGICAL :: HIP_LFF_NOW_CHANGED = .FALSE.     ! (shorthand for FF_NOW%GET_HOST_DATA_RDWR + the same on WVENVI) the host has modified them: the
                                               ! host copies count, they go up again (reset by the call)
  LOGICAL :: HIP_LWVPRPT_CHANGED = .FALSE.     ! (shorthand) the host has modified WVPRPT (new depth / currents): it goes up again (reset by the call)
  ABSTRACT INTERFACE
    SUBROUTINE INCDATE_IF(CDATE, ISHIFT)
      IMPORT :: JWIM
      CHARACTER(LEN=*), INTENT(INOUT) :: CDATE
      INTEGER(KIND=JWIM), INTENT(IN) :: ISHIFT
    END SUBROUTINE
  END INTERFACE
  PROCEDURE(INCDATE_IF), POINTER :: HIP_INCDATE => NULL()   ! the reference's INCDATE (incdate.F90)
END MODULE ECWAM_HIP_DRV

! Device -> host copies of the step's results into the YOWDRVTYPE-shaped host types: what SYNC_HOST_RDONLY / GET_HOST_DATA_RDONLY are in
! the reference's GPU build (wamintgr_loki_gpu.F90:197-200, wamodel.F90:350-366,435-470).  LFIELDS: FF_NOW, INTFLDS, MIJ; LSPECTRA: FL1
! and XLLWS (device [ij][K][M] -> chunked (NPROMA,NANG,NFRE,NCHNK)).  Called by WAMINTGR_HIP when HIP_LSYNC_* are set, or by the host
! at its output steps.
MODULE ECWAM_HIP_HOSTSYNC
  IMPLICIT NONE
CONTAINS
SUBROUTINE ECWAM_HIP_SYNC_HOST(FF_NOW, INTFLDS, MIJ, VARS_4D, LFIELDS, LSPECTRA)
  USE, INTRINSIC :: ISO_C_BINDING
  USE ECWAM_HIP_MOD
  TYPE(FORCING_FIELDS), INTENT(INOUT)     :: FF_NOW
  TYPE(INTGT_PARAM_FIELDS), INTENT(INOUT) :: INTFLDS
  TYPE(MIJ_TYPE), INTENT(INOUT)           :: MIJ
  TYPE(TYPE_4D), INTENT(INOUT)            :: VARS_4D
  LOGICAL, INTENT(IN) :: LFIELDS, LSPECTRA
  ! whole-type, synchronous: GET_HOST_DATA_RDONLY copies what the device holds newer than the host and returns with the copy done.  A host
  ! that wants single members or an asynchronous queue calls the methods itself: CALL VARS_4D%SYNC_HOST_RDONLY(FL1=.TRUE., QUEUE=4) ...
  ! CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=4), as wamintgr_loki_gpu.F90:197-200 / wamodel.F90:376 do
  IF (LFIELDS) THEN
    CALL FF_NOW%GET_HOST_DATA_RDONLY()
    CALL INTFLDS%GET_HOST_DATA_RDONLY()
    CALL MIJ%GET_HOST_DATA_RDONLY()
  ENDIF
  IF (LSPECTRA) CALL VARS_4D%GET_HOST_DATA_RDONLY()
  ! pad lanes of the last chunk keep the value of lane 1 (propag_wam.F90:388-398) -- done by POINTS_TO_CHUNKS / MEMBER_GATHER

END SUBROUTINE ECWAM_HIP_SYNC_HOST
END MODULE ECWAM_HIP_HOSTSYNC

SUBROUTINE WAMINTGR_HIP (CDTPRA, CDATE, CDATEWH, CDTIMP, CDTIMPNEXT,  &
 &                       BLK2GLO,                                    &
 &                       WVENVI, WVPRPT, FF_NOW, FF_NEXT, INTFLDS,   &
 &                       WAM2NEMO, MIJ, VARS_4D)
  USE, INTRINSIC :: ISO_C_BINDING
  USE ECWAM_HIP_MOD
  USE ECWAM_HIP_DRV
  USE ECWAM_HIP_HOSTSYNC
  IMPLICIT NONE
  CHARACTER(LEN=14), INTENT(IN)    :: CDTPRA      ! DATE FOR CALL PROPAGATION
  CHARACTER(LEN=14), INTENT(INOUT) :: CDATE       ! CURRENT DATE
  CHARACTER(LEN=14), INTENT(INOUT) :: CDATEWH     ! DATE OF THE NEXT FORCING FIELDS
  CHARACTER(LEN=14), INTENT(INOUT) :: CDTIMP      ! START DATE OF SOURCE FUNCTION INTEGRATION
  CHARACTER(LEN=14), INTENT(INOUT) :: CDTIMPNEXT  ! NEXT START DATE OF SOURCE FUNCTION INTEGRATION
  TYPE(WVGRIDGLO), INTENT(IN)             :: BLK2GLO
  TYPE(ENVIRONMENT), INTENT(INOUT)        :: WVENVI
  TYPE(FREQUENCY), INTENT(INOUT)          :: WVPRPT
  TYPE(FORCING_FIELDS), INTENT(INOUT)     :: FF_NOW
  TYPE(FORCING_FIELDS), INTENT(IN)        :: FF_NEXT
  TYPE(INTGT_PARAM_FIELDS), INTENT(INOUT) :: INTFLDS
  TYPE(WAVE2OCEAN), INTENT(INOUT)         :: WAM2NEMO
  TYPE(MIJ_TYPE), INTENT(INOUT)           :: MIJ
  TYPE(TYPE_4D), INTENT(INOUT)            :: VARS_4D

  TYPE(C_PTR) :: S0, DTMP
  INTEGER :: NP, NPROMA, NFRE, NANG, ISUB
  INTEGER(C_INT) :: NPASS, PMS(2), PME(2), PCP(2), PRG(2)
  REAL(C_DOUBLE) :: PDEL(2)
  LOGICAL :: LSOURCE_NOW, LFUSE
  INTEGER(C_INT) :: IFORMS, INEED

  S0 = HIPST%QUEUE(0)     ! every kernel and copy of the step on the compute queue (a non-blocking stream)
  NP = HIPST%NPTS; NPROMA = HIPST%NPROMA; NFRE = HIPST%NFRE; NANG = HIPST%NANG
  IF (.NOT. C_ASSOCIATED(HIPST%CTX)) THEN
    WRITE(0,*) 'WAMINTGR_HIP: ECWAM_HIP_SETUP has not been called'
    ERROR STOP 1
  ENDIF

  ! FF_NOW works on the device rows IMPLSCH reads and writes, FF_NEXT on the rows NEWWIND reads
  CALL ECWAM_HIP_BIND_FORCING(FF_NOW, FF_NEXT)
  IF (HIP_LFF_NOW_CHANGED) THEN      ! the host changed them without saying so through GET_HOST_DATA_RDWR: its copies are the valid ones
    FF_NOW%HIP%ST(:) = HIP_HOST_FRESH
    IF (ASSOCIATED(WVENVI%HIP)) WVENVI%HIP%ST(:) = HIP_HOST_FRESH
    HIP_LFF_NOW_CHANGED = .FALSE.
  ENDIF
  IF (HIP_LWVPRPT_CHANGED) THEN
    WVPRPT%HIP%ST(:) = HIP_HOST_FRESH
    HIP_LWVPRPT_CHANGED = .FALSE.
  ENDIF

  ! ---- wamintgr_loki_gpu.F90:99-105: the copies the host posted on queue 1 (wamodel.F90:209-214) have arrived; whatever the host holds
  !      newer than the device goes up now (first call: everything; later: nothing), FL1 becomes the device's to change
  CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=1)
  CALL VARS_4D%GET_DEVICE_DATA_RDWR(FL1=.TRUE.)
  CALL WVPRPT%GET_DEVICE_DATA_RDONLY()
  CALL WVENVI%GET_DEVICE_DATA_RDONLY(DEPTH=.TRUE., DELLAM1=.TRUE., COSPHM1=.TRUE., UCUR=.TRUE., VCUR=.TRUE., &
 &                                   EMAXDPT=.TRUE., IOBND=.TRUE., IODP=.TRUE., IBRMEM=.TRUE.)
  CALL BLK2GLO%GET_DEVICE_DATA_RDONLY()

  !*     PROPAGATION TIME (wamintgr.F90:94-100)
  LFUSE = .FALSE.
  IF (CDATE == CDTPRA) THEN
    IF (HIPST%IREFRA /= 0) THEN
      ! depth / current refraction: every weight rebuilt inside the stencil from the dot terms of ECWAM_HIP_SET_ENVIRONMENT
      IF (.NOT. C_ASSOCIATED(HIPST%D_REFR)) THEN
        WRITE(0,*) 'WAMINTGR_HIP: IREFRA /= 0 needs ECWAM_HIP_SET_ENVIRONMENT before the first propagation step'
        ERROR STOP 1
      ENDIF
      ! every PROPAGS2 call of the sequence behind its MPEXCHNG (propag_wam.F90:166,293), the exchange hidden behind the interior rows
      NPASS = 1
      IF (HIPST%IFRELFMAX <= 0) THEN
        PDEL(1) = REAL(HIPST%IDELPRO, C_DOUBLE); PMS(1) = 1; PME(1) = HIPST%NFRE_RED; PCP(1) = 1; PRG(1) = 0
        CALL EXCHANGE_AND_ADVECT_REFRA()
      ELSE
        ! fast waves with refraction (propag_wam.F90:175-212 + :247-313): the reference's sequence of calls on the full rows -- PROPAGS2 on
        ! 1..IFRELFMAX with DELPRO_LF (frequency range 0 of the CTUWDRV checks) and on the rest with IDELPRO (range 1), then per further
        ! sub-step FL1_EXT(:,:,1:IFRELFMAX) = FL3_EXT(...) and PROPAGS2 on the fast waves again
        NPASS = 2
        PDEL(1) = HIPST%DELPRO_LF; PMS(1) = 1; PME(1) = HIPST%IFRELFMAX; PCP(1) = 1; PRG(1) = 0
        PDEL(2) = REAL(HIPST%IDELPRO, C_DOUBLE); PMS(2) = HIPST%IFRELFMAX + 1; PME(2) = HIPST%NFRE_RED; PCP(2) = 0; PRG(2) = 1
        CALL EXCHANGE_AND_ADVECT_REFRA()
        NPASS = 1
        PCP(1) = 0
        DO ISUB = 2, NINT(REAL(HIPST%IDELPRO, C_DOUBLE) / HIPST%DELPRO_LF)
          CALL ECWAM_HIP_CHECK(ECWAM_HIP_COPY_FREQ_RANGE(HIPST%CTX, HIPST%D_FL3, HIPST%D_FL1, NP, 1_C_INT, HIPST%IFRELFMAX, 0_C_INT, S0), &
 &            'ECWAM_HIP_COPY_FREQ_RANGE')
          CALL EXCHANGE_AND_ADVECT_REFRA()
        ENDDO
      ENDIF
    ELSE
    ! PROPAG_WAM (propag_wam.F90:166,247-313): MPEXCHNG, then fast (M <= IFRELFMAX, DELPRO_LF) and slow waves in one pass, then the
    ! remaining fast-wave sub-steps on the compact buffer (exchanged again before each, as propag_wam.F90:293 does).  With more than
    ! one rank the exchange runs on the library's stream while the rows that read no halo row are advected.
    ! When the source terms are due right after this propagation step (wamintgr.F90:110) and a one-kernel build covers the configuration, the
    ! advection is left to IMPLSCH's tile load: only the exchange is posted here, FL1 / FL3 are swapped behind that kernel
    LFUSE = HIP_LFUSED_STEP .AND. HIP_LLSOURCE .AND. (HIP_CDTPRO >= CDTIMPNEXT)
    IF (LFUSE .AND. HIPST%IFRELFMAX > 0) LFUSE = C_ASSOCIATED(HIPST%D_G1)
    IF (LFUSE) THEN      ! the forms this configuration needs: bit 0 the plain step, bit 1 fast waves, bit 2 obstructions (LSUBGRID)
      IFORMS = ECWAM_HIP_PROPAGS2_IMPLSCH_SUPPORTED(HIPST%CTX)
      INEED = 1
      IF (HIPST%IFRELFMAX > 0) INEED = INEED + 2
      IF (C_ASSOCIATED(HIPST%D_OBS)) INEED = INEED + 4
      LFUSE = IAND(IFORMS, INEED) == INEED
    ENDIF
    IF (HIPST%IFRELFMAX > 0) THEN
      ! the fast waves do not depend on the slow ones: their sub-steps 1 .. NSTEP_LF-1 first, compact rows -> compact rows, then one full
      ! pass that takes their last state from the compact rows as the input of the last sub-step and writes complete FL3 rows and the
      ! compact copy the next advection step starts from -- the arithmetic of propag_wam.F90:247-313 per element, no pass writing a
      ! frequency sub-range into full rows
      IF (.NOT. HIPST%LGFAST_VALID) CALL ECWAM_HIP_CHECK(ECWAM_HIP_COPY_FREQ_RANGE(HIPST%CTX, HIPST%D_FL1, HIPST%D_G1, NP, 1_C_INT, &
 &          HIPST%LFP, HIPST%LFP, S0), 'ECWAM_HIP_COPY_FREQ_RANGE')
      DO ISUB = 1, NINT(REAL(HIPST%IDELPRO, C_DOUBLE) / HIPST%DELPRO_LF) - 1
        CALL HIP_HALO_START(HIPST%D_G1, NANG * HIPST%LFP, S0)
        CALL ADVECT_FAST(HIPST%KIJS_INT - 1, HIPST%KIJL_INT)
        IF (HIPST%LDECOMP) THEN
          CALL HIP_HALO_FINISH(S0)
          CALL ADVECT_FAST(0_C_INT, HIPST%KIJS_INT - 1)
          CALL ADVECT_FAST(HIPST%KIJL_INT, NP)
        ENDIF
        DTMP = HIPST%D_G1; HIPST%D_G1 = HIPST%D_G2; HIPST%D_G2 = DTMP
      ENDDO
      CALL HIP_HALO_START(HIPST%D_G1, NANG * HIPST%LFP, S0)     ! the short rows first: the second exchange queues behind the first
    ENDIF
    CALL HIP_HALO_START(HIPST%D_FL1, NANG * NFRE, S0)
    IF (.NOT. LFUSE) THEN      ! (LFUSE: the full pass is the tile load of the source-term kernel below; the exchanges are posted)
    CALL ADVECT_FULL(HIPST%KIJS_INT - 1, HIPST%KIJL_INT)
    IF (HIPST%LDECOMP) THEN
      CALL HIP_HALO_FINISH(S0)
      CALL ADVECT_FULL(0_C_INT, HIPST%KIJS_INT - 1)
      CALL ADVECT_FULL(HIPST%KIJL_INT, NP)
    ENDIF
    IF (HIPST%IFRELFMAX > 0) THEN
      DTMP = HIPST%D_G1; HIPST%D_G1 = HIPST%D_G2; HIPST%D_G2 = DTMP
      HIPST%LGFAST_VALID = .TRUE.
    ENDIF
    ENDIF
    ENDIF
    IF (.NOT. LFUSE) THEN
      DTMP = HIPST%D_FL1; HIPST%D_FL1 = HIPST%D_FL3; HIPST%D_FL3 = DTMP
    ENDIF
    CDATE = HIP_CDTPRO
  ENDIF

  !* RETRIEVING NEW FORCING FIELDS IF NEEDED (wamintgr.F90:105, newwind.F90:107-174; wamintgr_loki_gpu.F90:121-131)
  CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=2)
  CALL FF_NOW%GET_DEVICE_DATA_RDWR(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
 & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
 & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE.)
  CALL FF_NEXT%GET_DEVICE_DATA_RDONLY(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
 & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
 & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE.)
  IF (CDTIMP >= CDATEWH) THEN
    IF (CDTIMP >= HIP_CDATEFL) HIP_LLNEWFILE = .TRUE.     ! newwind.F90:110-112
    IF (HIP_ICODE_WND /= 0) THEN
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_NEWWIND_ICODE(HIPST%CTX, NP, HIPST%D_FF, HIPST%D_FFN, HIP_ICODE_WND, S0), 'ECWAM_HIP_NEWWIND')
    ELSE
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_NEWWIND(HIPST%CTX, NP, HIPST%D_FF, HIPST%D_FFN, S0), 'ECWAM_HIP_NEWWIND')
    ENDIF
    IF (ASSOCIATED(HIP_INCDATE)) CALL HIP_INCDATE(CDATEWH, HIP_IDELWO)
  ENDIF

  ! IT IS TIME TO INTEGRATE THE SOURCE TERMS (wamintgr.F90:110-186)
  LSOURCE_NOW = (CDATE >= CDTIMPNEXT)
  IF (LSOURCE_NOW .AND. HIP_LLSOURCE) THEN
    ! wamintgr_loki_gpu.F90:139-157: the accumulating WAVE2OCEAN members travel up when the host holds newer values (after UPDNEMOSTRESS
    ! reset them, wamodel.F90:617-640), everything IMPLSCH writes becomes the device's
    CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=3)
    IF (HIPST%LWNEMOCOU .AND. (.NOT. HIPST%LWCOU)) THEN
      CALL WAM2NEMO%GET_DEVICE_DATA_RDWR(NEMOTAUICX=.TRUE., NEMOTAUICY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., &
 &      NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE., NEMOTAUX=.TRUE., NEMOTAUY=.TRUE.)
    ELSE
      CALL WAM2NEMO%GET_DEVICE_DATA_WRONLY(NEMOTAUICX=.TRUE., NEMOTAUICY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., &
 &      NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE., NEMOTAUX=.TRUE., NEMOTAUY=.TRUE.)
    ENDIF
    CALL WAM2NEMO%GET_DEVICE_DATA_WRONLY(NEMOUSTOKES=.TRUE., NEMOVSTOKES=.TRUE., NEMOSTRN=.TRUE.)
    CALL INTFLDS%GET_DEVICE_DATA_WRONLY(WSEMEAN=.TRUE., WSFMEAN=.TRUE., USTOKES=.TRUE., &
 &    VSTOKES=.TRUE., STRNMS=.TRUE., TAUXD=.TRUE., TAUYD=.TRUE., TAUOCXD=.TRUE., &
 &    TAUOCYD=.TRUE., TAUOC=.TRUE., PHIOCD=.TRUE., PHIEPS=.TRUE., PHIAW=.TRUE., &
 &    TAUICX=.TRUE., TAUICY=.TRUE.)
    CALL VARS_4D%GET_DEVICE_DATA_WRONLY(XLLWS=.TRUE.)
    CALL MIJ%GET_DEVICE_DATA_WRONLY()
    IF (LFUSE) THEN
      ! PROPAGS2 + IMPLSCH of the rows that read no halo row while the exchange runs, then the two ends of the band (propag_wam.F90:166)
      CALL ADVECT_AND_INTEGRATE(HIPST%KIJS_INT - 1, HIPST%KIJL_INT)
      IF (HIPST%LDECOMP) THEN
        CALL HIP_HALO_FINISH(S0)
        CALL ADVECT_AND_INTEGRATE(0_C_INT, HIPST%KIJS_INT - 1)
        CALL ADVECT_AND_INTEGRATE(HIPST%KIJL_INT, NP)
      ENDIF
      DTMP = HIPST%D_FL1; HIPST%D_FL1 = HIPST%D_FL3; HIPST%D_FL3 = DTMP
      IF (HIPST%IFRELFMAX > 0) THEN      ! D_G2 received the new fast waves: it is what the next advection step starts from
        DTMP = HIPST%D_G1; HIPST%D_G1 = HIPST%D_G2; HIPST%D_G2 = DTMP
        HIPST%LGFAST_VALID = .TRUE.
      ENDIF
    ELSE
    CALL FAST_SINK(.TRUE.)      ! IMPLSCH leaves the fast waves of its result in the compact rows as well
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_IMPLSCH(HIPST%CTX, 0_C_INT, NP, HIPST%D_FL1, HIPST%D_WVPRPT, HIPST%D_FF, HIPST%D_INTF, &
 &        HIPST%D_MIJ, HIPST%D_XLLWS, HIPST%D_W2N, C_NULL_PTR, S0), 'ECWAM_HIP_IMPLSCH')
    CALL FAST_SINK(.FALSE.)
    ENDIF
    IF (HIP_LSYNC_EAGER) THEN      ! wamintgr_loki_gpu.F90:197-200
      CALL VARS_4D%SYNC_HOST_RDONLY(FL1=.TRUE., QUEUE=4)
      CALL FF_NOW%SYNC_HOST_RDONLY(QUEUE=4)
      CALL WVENVI%SYNC_HOST_RDONLY(QUEUE=5)
      IF (HIPST%LWNEMOCOU .AND. (.NOT. HIPST%LWCOU)) CALL WAM2NEMO%SYNC_HOST_RDONLY(QUEUE=6)
    ENDIF
    IF (HIPST%LWNEMOCOU) HIP_NEMONTAU = HIP_NEMONTAU + 1      ! wamintgr.F90:150 / wamintgr_loki_gpu.F90:203
  ELSEIF (LSOURCE_NOW) THEN
    ! NO SOURCE TERM CONTRIBUTION (wamintgr.F90:152-160): MIJ = NFRE, FL1 = MAX(FL1, EPSMIN), XLLWS = 0 -- on the device copies
    CALL VARS_4D%GET_DEVICE_DATA_WRONLY(XLLWS=.TRUE.)
    CALL MIJ%GET_DEVICE_DATA_WRONLY()
    CALL FAST_SINK(.TRUE.)
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_NOSOURCE(HIPST%CTX, 0_C_INT, NP, HIPST%D_FL1, HIPST%D_MIJ, HIPST%D_XLLWS, S0), 'ECWAM_HIP_NOSOURCE')
    CALL FAST_SINK(.FALSE.)
  ENDIF
  IF (LSOURCE_NOW) THEN
    ! UPDATE FORCING FIELDS TIME COUNTER (wamintgr.F90:164-176)
    IF (HIP_LLNEWFILE) THEN
      HIP_LLNEWFILE = .FALSE.
      IF (ASSOCIATED(HIP_INCDATE)) THEN
        CALL HIP_INCDATE(HIP_CDAWIFL, MAX(HIP_IDELWI, HIP_IDELPRO))
        CALL HIP_INCDATE(HIP_CDATEFL, MAX(HIP_IDELWI, HIP_IDELPRO))
      ENDIF
    ENDIF
    HIP_CDATEWO = CDATEWH
    CDTIMP = CDTIMPNEXT
    IF (ASSOCIATED(HIP_INCDATE)) CALL HIP_INCDATE(CDTIMPNEXT, HIP_IDELT)
  ELSE
    ! not yet time for the source terms (wamintgr.F90:178-186): MIJ = NFRE, XLLWS = 0, the spectra as advected
    CALL VARS_4D%GET_DEVICE_DATA_WRONLY(XLLWS=.TRUE.)
    CALL MIJ%GET_DEVICE_DATA_WRONLY()
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_NOSOURCE(HIPST%CTX, 0_C_INT, NP, C_NULL_PTR, HIPST%D_MIJ, HIPST%D_XLLWS, S0), 'ECWAM_HIP_NOSOURCE')
  ENDIF

  ! ---- results back to the host-side derived types only when the host asks (SYNC_HOST_RDONLY of the OpenACC variant, :197-200)
  IF (HIP_LSYNC_FIELDS .OR. HIP_LSYNC_SPECTRA) CALL ECWAM_HIP_SYNC_HOST(FF_NOW, INTFLDS, MIJ, VARS_4D, HIP_LSYNC_FIELDS, HIP_LSYNC_SPECTRA)

CONTAINS
  ! PROPAGS2 with the weights rebuilt in the stencil on local rows [K0, K1) (0-based, half open): the full spectra (fast and slow waves
  ! in one pass, the fast waves also into the compact buffer), or one fast-wave sub-step on the compact buffer
  SUBROUTINE FAST_SINK(LON)
    LOGICAL, INTENT(IN) :: LON
    IF (HIPST%IFRELFMAX <= 0 .OR. HIPST%IREFRA /= 0 .OR. .NOT. C_ASSOCIATED(HIPST%D_G1)) RETURN
    IF (LON) THEN
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_SET_FASTWAVE_COPY(HIPST%CTX, HIPST%D_G1, HIPST%LFP), 'ECWAM_HIP_SET_FASTWAVE_COPY')
      HIPST%LGFAST_VALID = .TRUE.
    ELSE      ! the library keeps no pointer into a buffer this layer swaps
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_SET_FASTWAVE_COPY(HIPST%CTX, C_NULL_PTR, 0_C_INT), 'ECWAM_HIP_SET_FASTWAVE_COPY')
    ENDIF
  END SUBROUTINE
  ! the one-kernel step on local rows [K0, K1): PROPAGS2 from the rows of FL1 inside IMPLSCH's tile load, the new spectrum to the rows of FL3
  SUBROUTINE ADVECT_AND_INTEGRATE(K0, K1)
    INTEGER(C_INT), INTENT(IN) :: K0, K1
    IF (K1 <= K0) RETURN
    IF (HIPST%IFRELFMAX > 0) THEN
      ! the fast waves' last sub-step reads the compact rows D_G1 (after their sub-steps 1 .. NSTEP_LF-1); the new fast waves go to D_G2
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_SET_FASTWAVE_COPY(HIPST%CTX, HIPST%D_G2, HIPST%LFP), 'ECWAM_HIP_SET_FASTWAVE_COPY')
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_PROPAGS2_IMPLSCH(HIPST%CTX, HIPST%D_FL1, HIPST%D_FL3, NP, HIPST%NGY, REAL(HIPST%IDELPRO, C_DOUBLE), &
 &          HIPST%D_KXLT, HIPST%D_ZD, HIPST%XDELLA, HIPST%D_CP, HIPST%D_SP, HIPST%D_KLON, HIPST%D_KLAT, HIPST%D_KCOR, HIPST%D_WLAT, &
 &          HIPST%D_WCOR, HIPST%D_CG, HIPST%D_CM1, K0, K1, 1_C_INT, HIPST%NFRE_RED, HIPST%D_WVPRPT, HIPST%D_FF, HIPST%D_INTF, HIPST%D_MIJ, &
 &          HIPST%D_XLLWS, HIPST%D_W2N, HIPST%DELPRO_LF, HIPST%IFRELFMAX, HIPST%D_G1, HIPST%LFP, 0_C_INT, S0), 'ECWAM_HIP_PROPAGS2_IMPLSCH')
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_SET_FASTWAVE_COPY(HIPST%CTX, C_NULL_PTR, 0_C_INT), 'ECWAM_HIP_SET_FASTWAVE_COPY')
      RETURN
    ENDIF
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_PROPAGS2_IMPLSCH(HIPST%CTX, HIPST%D_FL1, HIPST%D_FL3, NP, HIPST%NGY, REAL(HIPST%IDELPRO, C_DOUBLE), &
 &        HIPST%D_KXLT, HIPST%D_ZD, HIPST%XDELLA, HIPST%D_CP, HIPST%D_SP, HIPST%D_KLON, HIPST%D_KLAT, HIPST%D_KCOR, HIPST%D_WLAT, &
 &        HIPST%D_WCOR, HIPST%D_CG, HIPST%D_CM1, K0, K1, 1_C_INT, HIPST%NFRE_RED, HIPST%D_WVPRPT, HIPST%D_FF, HIPST%D_INTF, HIPST%D_MIJ, &
 &        HIPST%D_XLLWS, HIPST%D_W2N, 0.0_C_DOUBLE, 0_C_INT, C_NULL_PTR, 0_C_INT, 0_C_INT, S0), 'ECWAM_HIP_PROPAGS2_IMPLSCH')
  END SUBROUTINE
  SUBROUTINE ADVECT_FULL(K0, K1)
    INTEGER(C_INT), INTENT(IN) :: K0, K1
    TYPE(C_PTR) :: GIN, GOUT
    INTEGER(C_INT) :: GK
    IF (K1 <= K0) RETURN
    GIN = C_NULL_PTR; GOUT = C_NULL_PTR; GK = 0
    IF (HIPST%IFRELFMAX > 0) THEN
      GIN = HIPST%D_G1; GOUT = HIPST%D_G2; GK = HIPST%LFP
    ENDIF
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_PROPAGS2_OTF_FAST(HIPST%CTX, HIPST%D_FL1, HIPST%D_FL3, NP, HIPST%NGY, &
 &        REAL(HIPST%IDELPRO, C_DOUBLE), HIPST%DELPRO_LF, HIPST%IFRELFMAX, 0_C_INT, GIN, GK, 0_C_INT, GOUT, GK, &
 &        HIPST%D_KXLT, HIPST%D_ZD, HIPST%XDELLA, HIPST%D_CP, HIPST%D_SP, HIPST%D_KLON, HIPST%D_KLAT, HIPST%D_KCOR, HIPST%D_WLAT, &
 &        HIPST%D_WCOR, HIPST%D_CG, HIPST%D_CM1, C_NULL_PTR, K0, K1, 1_C_INT, HIPST%NFRE_RED, 1_C_INT, S0), 'ECWAM_HIP_PROPAGS2_OTF')
  END SUBROUTINE
  ! one fast-wave sub-step, compact rows D_G1 -> D_G2 (COPY_REST carries the slow frequencies that fill the last 16-byte vector)
  SUBROUTINE ADVECT_FAST(K0, K1)
    INTEGER(C_INT), INTENT(IN) :: K0, K1
    IF (K1 <= K0) RETURN
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_PROPAGS2_OTF_FAST(HIPST%CTX, HIPST%D_G1, HIPST%D_G2, NP, HIPST%NGY, HIPST%DELPRO_LF, &
 &        HIPST%DELPRO_LF, 0_C_INT, HIPST%LFP, C_NULL_PTR, 0_C_INT, HIPST%LFP, C_NULL_PTR, 0_C_INT, &
 &        HIPST%D_KXLT, HIPST%D_ZD, HIPST%XDELLA, HIPST%D_CP, HIPST%D_SP, HIPST%D_KLON, HIPST%D_KLAT, HIPST%D_KCOR, HIPST%D_WLAT, &
 &        HIPST%D_WCOR, HIPST%D_CG, HIPST%D_CM1, C_NULL_PTR, K0, K1, 1_C_INT, HIPST%IFRELFMAX, 1_C_INT, S0), &
 &        'ECWAM_HIP_PROPAGS2_OTF (fast waves)')
  END SUBROUTINE
  ! PROPAGS2 with refraction on frequencies MS..ME with time step DELPRO; ICOPY: also copy the frequencies the stencil leaves alone;
  ! IRANGE: which CURMASK of the CTUWDRV checks applies (0: the only or the fast range, 1: the slow range of a split call)
  SUBROUTINE ADVECT_REFRA(DELPRO, MS, ME, ICOPY, IRANGE, K0, K1)
    REAL(C_DOUBLE), INTENT(IN) :: DELPRO
    INTEGER(C_INT), INTENT(IN) :: MS, ME, ICOPY, IRANGE, K0, K1
    IF (K1 <= K0) RETURN
    CALL ECWAM_HIP_CHECK(ECWAM_HIP_PROPAGS2_REFRA(HIPST%CTX, HIPST%D_FL1, HIPST%D_FL3, NP, HIPST%NGY, DELPRO, &
 &        HIPST%D_KXLT, HIPST%D_ZD, HIPST%XDELLA, HIPST%D_CP, HIPST%D_SP, HIPST%D_KLON, HIPST%D_KLAT, HIPST%D_KCOR, HIPST%D_WLAT, &
 &        HIPST%D_WCOR, HIPST%D_CG, HIPST%D_OM, HIPST%D_WN, HIPST%D_CM1, HIPST%D_REFR, IRANGE, K0, K1, MS, ME, &
 &        ICOPY, S0), 'ECWAM_HIP_PROPAGS2_REFRA')
  END SUBROUTINE
  ! MPEXCHNG of FL1, then the NPASS calls (PDEL, PMS, PME, PCP, PRG) on the rows that read no halo row while the exchange runs, then on the two ends
  SUBROUTINE EXCHANGE_AND_ADVECT_REFRA()
    INTEGER :: I
    CALL HIP_HALO_START(HIPST%D_FL1, NANG * NFRE, S0)
    DO I = 1, NPASS
      CALL ADVECT_REFRA(PDEL(I), PMS(I), PME(I), PCP(I), PRG(I), HIPST%KIJS_INT - 1, HIPST%KIJL_INT)
    ENDDO
    IF (HIPST%LDECOMP) THEN
      CALL HIP_HALO_FINISH(S0)
      DO I = 1, NPASS
        CALL ADVECT_REFRA(PDEL(I), PMS(I), PME(I), PCP(I), PRG(I), 0_C_INT, HIPST%KIJS_INT - 1)
        CALL ADVECT_REFRA(PDEL(I), PMS(I), PME(I), PCP(I), PRG(I), HIPST%KIJL_INT, NP)
      ENDDO
    ENDIF
  END SUBROUTINE
END SUBROUTINE WAMINTGR_HIP
