! seam_sequence.F90 -- harness program (this repository's own, not part of ecWAM): the call sequence of the reference's GPU build around
! its time-step driver, with WAMINTGR_HIP in the place of WAMINTGR_LOKI_GPU (the seam, wamodel.F90:289-299).  The FIELD_API calls are
! the reference's own lines, argument for argument:
!     wamodel.F90:207-226                      initial asynchronous host -> device copies on queues 1, 2, 3
!     wamintgr_loki_gpu.F90:100-105,121-131,141-157   the GET_DEVICE_DATA_* calls of the driver (WAMINTGR_HIP makes them itself; made here as
!                                              well, in front of the call: they are idempotent)
!     wamintgr_loki_gpu.F90:197-200            per-step asynchronous device -> host copies on queues 4, 5, 6
!     wamodel.F90:376-385, 435-470             output and restart steps: WAIT_FOR_ASYNC_QUEUE + GET_HOST_DATA_RDONLY
!     wamodel.F90:614-642                      NEMO coupling step: WAM2NEMO back on queue 6, UPDNEMOSTRESS on the host, back up on queue 3
!     wamodel.F90:651-671                      end of the run: GET_HOST_DATA_RDWR, DELETE_DEVICE_DATA
! compiled against this repository's YOWDRVTYPE / FIELD_ASYNC_MODULE / PARKIND_WAVE (yowdrvtype_hip.F90) -- the modules those lines USE in
! the reference.  What the host code between them does is restated in a few lines each: an output step reads FL1, UPDNEMOSTRESS averages
! and resets the accumulated stresses (updnemostress.F90:84-123), new winds arrive in FF_NEXT once.  Along the way the harness checks
! the intent tracking of SURVEY.md 8(b) -- no device -> host copy of anything the device did not write, no second copy of what the host
! already has, no host -> device copy of anything the host did not change -- with the copy counters of ECWAM_HIP_CAPI, and stops on a
! violation.  tests/test_gpu_fortran.py compares everything it writes with the Python host bit for bit.
PROGRAM SEAM_SEQUENCE
  USE, INTRINSIC :: ISO_C_BINDING
  USE PARKIND_WAVE, ONLY : JWIM, JWRB, JWRO
  USE YOWDRVTYPE  , ONLY : WVGRIDGLO, ENVIRONMENT, FREQUENCY, FORCING_FIELDS,  &
 &                       INTGT_PARAM_FIELDS, WAVE2OCEAN, MIJ_TYPE, TYPE_4D
  USE FIELD_ASYNC_MODULE, ONLY : WAIT_FOR_ASYNC_QUEUE
  USE ECWAM_HIP_CAPI, ONLY : HIPST, ECWAM_HIP_CHECK, ECWAM_HIP_SYNC
  USE ECWAM_HIP_MOD, ONLY : ECWAM_HIP_FINALIZE
  USE ECWAM_HIP_RESTART, ONLY : ECWAM_HIP_WRITEFL, ECWAM_HIP_READFL
  USE ECWAM_HIP_DRV
  USE HARNESS_CASE
  IMPLICIT NONE
  INTERFACE
    SUBROUTINE WAMINTGR_HIP (CDTPRA, CDATE, CDATEWH, CDTIMP, CDTIMPNEXT, BLK2GLO, WVENVI, WVPRPT, FF_NOW, FF_NEXT, INTFLDS, &
 &                           WAM2NEMO, MIJ, VARS_4D)
      USE YOWDRVTYPE
      CHARACTER(LEN=14), INTENT(IN)    :: CDTPRA
      CHARACTER(LEN=14), INTENT(INOUT) :: CDATE, CDATEWH, CDTIMP, CDTIMPNEXT
      TYPE(WVGRIDGLO), INTENT(IN)             :: BLK2GLO
      TYPE(ENVIRONMENT), INTENT(INOUT)        :: WVENVI
      TYPE(FREQUENCY), INTENT(INOUT)          :: WVPRPT
      TYPE(FORCING_FIELDS), INTENT(INOUT)     :: FF_NOW
      TYPE(FORCING_FIELDS), INTENT(IN)        :: FF_NEXT
      TYPE(INTGT_PARAM_FIELDS), INTENT(INOUT) :: INTFLDS
      TYPE(WAVE2OCEAN), INTENT(INOUT)         :: WAM2NEMO
      TYPE(MIJ_TYPE), INTENT(INOUT)           :: MIJ
      TYPE(TYPE_4D), INTENT(INOUT)            :: VARS_4D
    END SUBROUTINE
  END INTERFACE

  CHARACTER(LEN=14) :: CDTPRA, CDATE, CDATEWH, CDTIMP, CDTIMPNEXT, CDTPRO
  LOGICAL :: LWNEMOCOU, LWCOU, LSRC
  INTEGER(KIND=JWIM) :: KADV, NADV, ILOOP, NEMOWSTEP, NEMOFRCO, NEMONTAU_AVG, IASSI
  INTEGER(KIND=JWIM), PARAMETER :: KOUT = 2      ! output step: the host reads FL1 and FF_NOW after this advection step ...
  INTEGER(KIND=JWIM), PARAMETER :: KWIND = 3     ! ... and new winds are in FF_NEXT from this one on
  INTEGER(C_LONG_LONG) :: C0(4), C1(4)
  REAL(KIND=JWRB), ALLOCATABLE :: FL1_OUT(:,:,:,:), UFRIC_OUT(:,:)
  REAL(KIND=JWRO), ALLOCATABLE :: STRESS_AVG(:,:,:,:)
  INTEGER :: NCOUP, IU, NUP_EXPECTED, IJ
  CHARACTER(LEN=512) :: FRESTART
  TYPE(TYPE_4D) :: CHK_4D
  INTEGER(KIND=JWIM), ALLOCATABLE :: IJ2NEWIJ(:)

  CALL CASE_READ_AND_SETUP()
  LWNEMOCOU = (P%LWNEMOCOU /= 0); LWCOU = (P%LWCOU /= 0)
  NADV = NSTEP; NEMOFRCO = 2; NEMOWSTEP = 0; NCOUP = 0; IASSI = 1
  ALLOCATE(FL1_OUT(NPROMA,NANG,NFRE,NCHNK), UFRIC_OUT(NPROMA,NCHNK), STRESS_AVG(NPROMA,NCHNK,6,MAX(1,NADV/NEMOFRCO)))
  HIP_IDELT = IDELT; HIP_IDELPRO = IDELPRO; HIP_INCDATE => SIMPLE_INCDATE; HIP_LLSOURCE = (HDR(19) == 0)
  HIP_NEMONTAU = 0; NSOURCE = 0; NUP_EXPECTED = 0
  ! the next forcing fields: the ones the run starts with, until new winds arrive at step KWIND
  FF_NEXT%AIRD = FF_NOW%AIRD; FF_NEXT%WDWAVE = FF_NOW%WDWAVE; FF_NEXT%CICOVER = FF_NOW%CICOVER; FF_NEXT%WSWAVE = FF_NOW%WSWAVE
  FF_NEXT%WSTAR = FF_NOW%WSTAR; FF_NEXT%USTRA = FF_NOW%USTRA; FF_NEXT%VSTRA = FF_NOW%VSTRA; FF_NEXT%UFRIC = FF_NOW%UFRIC
  FF_NEXT%TAUW = FF_NOW%TAUW; FF_NEXT%TAUWDIR = FF_NOW%TAUWDIR; FF_NEXT%Z0M = FF_NOW%Z0M; FF_NEXT%Z0B = FF_NOW%Z0B
  FF_NEXT%CHRNCK = FF_NOW%CHRNCK; FF_NEXT%CITHICK = FF_NOW%CITHICK

  ! ---- wamodel.F90:207-226 ------------------------------------------------------------------------------------------------------------
  CALL COUNTERS(C0)
      CALL WVPRPT_LAND%SYNC_DEVICE_RDONLY(QUEUE=1)
      CALL VARS_4D%SYNC_DEVICE_RDWR(FL1=.TRUE., QUEUE=1)
      CALL BLK2GLO%SYNC_DEVICE_RDONLY(QUEUE=1)
      CALL WVPRPT%SYNC_DEVICE_RDONLY(QUEUE=1)
      CALL WVENVI%SYNC_DEVICE_RDONLY(DEPTH=.TRUE., DELLAM1=.TRUE., COSPHM1=.TRUE., UCUR=.TRUE., VCUR=.TRUE., &
      &                              EMAXDPT=.TRUE., IOBND=.TRUE., IODP=.TRUE., IBRMEM=.TRUE., QUEUE=1)
      CALL FF_NOW%SYNC_DEVICE_RDWR(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
      & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
      & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE., QUEUE=2)
      CALL FF_NEXT%SYNC_DEVICE_RDONLY(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
      & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
      & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE., QUEUE=2)
      IF (LWNEMOCOU .AND. (.NOT.LWCOU)) THEN
         CALL WAM2NEMO%SYNC_DEVICE_RDWR(NEMOTAUICX=.TRUE., NEMOTAUICY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., &
         & NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE., NEMOTAUX=.TRUE., NEMOTAUY=.TRUE., QUEUE=3)
      ENDIF
  ! (the FORCING_FIELDS objects are bound to their device rows by the first WAMINTGR_HIP: their copies start there, everything else is
  !  on its way now: FL1, 5 FREQUENCY members, DEPTH / EMAXDPT / IBRMEM, and -- LWNEMOCOU -- 10 WAVE2OCEAN members)
  CALL COUNTERS(C1)
  IF (C1(1) - C0(1) /= 9 + MERGE(10, 0, LWNEMOCOU .AND. .NOT. LWCOU) .OR. C1(2) /= C0(2)) CALL BAD('initial offload', C0, C1)

  CDTPRO = '00000000000000'; CDTIMP = CDTPRO; CDATEWH = '99999999999999'
  CDTIMPNEXT = CDTPRO; CALL SIMPLE_INCDATE(CDTIMPNEXT, IDELT)        ! wamodel.F90:182-185

  ADVECTION : DO KADV = 1, NADV                                       ! wamodel.F90:228-312
    CDTPRA = CDTPRO
    CALL SIMPLE_INCDATE(CDTPRO, IDELPRO)
    HIP_CDTPRO = CDTPRO

    IF (KADV == KWIND) THEN
      ! new winds (GETWND fills FF_NEXT on the host in the reference): the host says that it is going to write, writes, sends them up
      ! on queue 2; NEWWIND hands them over inside the next call (CDATEWH = now)
      CALL FF_NEXT%GET_HOST_DATA_RDWR()
      FF_NEXT%WSWAVE = 1.1_JWRB * FF_NEXT%WSWAVE; FF_NEXT%WDWAVE = FF_NEXT%WDWAVE + 0.3_JWRB
      FF_NEXT%CICOVER = 0.5_JWRB * FF_NEXT%CICOVER
      CALL COUNTERS(C0)
      CALL FF_NEXT%SYNC_DEVICE_RDONLY(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
      & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
      & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE., QUEUE=2)
      CALL COUNTERS(C1)
      IF (C1(1) - C0(1) /= 14 .OR. C1(2) /= C0(2)) CALL BAD('new winds', C0, C1)
      CDATEWH = CDTIMP
    ENDIF

    CDATE   = CDTPRA
    ILOOP = 1
    DO WHILE ( ILOOP == 1 .OR. CDTIMPNEXT <= CDTPRO)
      LSRC = (MERGE(CDTPRO, CDATE, CDATE == CDTPRA) >= CDTIMPNEXT)        ! this call integrates the source terms
      IF (LSRC .AND. HIP_LLSOURCE) NSOURCE = NSOURCE + 1
      CALL COUNTERS(C0)
      ! ---- wamintgr_loki_gpu.F90:100-105 ----------------------------------------------------------------------------------------------
CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=1)
CALL VARS_4D%GET_DEVICE_DATA_RDWR(FL1=.TRUE.)
CALL WVPRPT%GET_DEVICE_DATA_RDONLY()
CALL WVENVI%GET_DEVICE_DATA_RDONLY(DEPTH=.TRUE., DELLAM1=.TRUE., COSPHM1=.TRUE., UCUR=.TRUE., VCUR=.TRUE., &
&                                  EMAXDPT=.TRUE., IOBND=.TRUE., IODP=.TRUE., IBRMEM=.TRUE.)
CALL BLK2GLO%GET_DEVICE_DATA_RDONLY()
      ! ---- wamintgr_loki_gpu.F90:121-127 (unbound before the first WAMINTGR_HIP: nothing moves yet) -----------------------------------
CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=2)
CALL FF_NOW%GET_DEVICE_DATA_RDWR(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
& WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
& CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE.)
CALL FF_NEXT%GET_DEVICE_DATA_RDONLY(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
& WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
& CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE.)
      IF (LSRC .AND. HIP_LLSOURCE) THEN
      ! ---- wamintgr_loki_gpu.F90:141-157 ----------------------------------------------------------------------------------------------
      CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=3)
      IF (LWNEMOCOU .AND. (.NOT.LWCOU)) THEN
        CALL WAM2NEMO%GET_DEVICE_DATA_RDWR(NEMOTAUICX=.TRUE., NEMOTAUICY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., &
        & NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE., NEMOTAUX=.TRUE., NEMOTAUY=.TRUE.)
      ELSE
        CALL WAM2NEMO%GET_DEVICE_DATA_WRONLY(NEMOTAUICX=.TRUE., NEMOTAUICY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., &
        & NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE., NEMOTAUX=.TRUE., NEMOTAUY=.TRUE.)
      ENDIF
      CALL WAM2NEMO%GET_DEVICE_DATA_WRONLY(NEMOUSTOKES=.TRUE., NEMOVSTOKES=.TRUE., NEMOSTRN=.TRUE.)
      CALL INTFLDS%GET_DEVICE_DATA_WRONLY(WSEMEAN=.TRUE., WSFMEAN=.TRUE., USTOKES=.TRUE., &
      & VSTOKES=.TRUE., STRNMS=.TRUE., TAUXD=.TRUE., TAUYD=.TRUE., TAUOCXD=.TRUE., &
      & TAUOCYD=.TRUE., TAUOC=.TRUE., PHIOCD=.TRUE., PHIEPS=.TRUE., PHIAW=.TRUE., &
      & TAUICX=.TRUE., TAUICY=.TRUE.)
      CALL VARS_4D%GET_DEVICE_DATA_WRONLY(XLLWS=.TRUE.)
      CALL MIJ%GET_DEVICE_DATA_WRONLY()
      ENDIF

      CALL WAMINTGR_HIP (CDTPRA, CDATE, CDATEWH, CDTIMP, CDTIMPNEXT, &
 &                       BLK2GLO,                                    &
 &                       WVENVI, WVPRPT, FF_NOW, FF_NEXT, INTFLDS,   &
 &                       WAM2NEMO, MIJ, VARS_4D)

      CALL COUNTERS(C1)
      ! nothing travels down inside the driver; up: on the first call the two FORCING_FIELDS objects (bound now), later nothing
      IF (C1(2) /= C0(2)) CALL BAD('device -> host inside WAMINTGR_HIP', C0, C1)
      IF (KADV == 1 .AND. ILOOP == 1) THEN
        IF (C1(1) - C0(1) /= 28) CALL BAD('first WAMINTGR_HIP', C0, C1)
      ELSE      ! (after a coupling step: NEMOTAUICX / Y, which the host reset and wamodel.F90:638 does not name)
        IF (C1(1) - C0(1) /= NUP_EXPECTED) CALL BAD('host -> device in a later WAMINTGR_HIP', C0, C1)
      ENDIF
      NUP_EXPECTED = 0

      IF (LSRC .AND. HIP_LLSOURCE) THEN
      ! ---- wamintgr_loki_gpu.F90:197-200 ----------------------------------------------------------------------------------------------
      CALL COUNTERS(C0)
      CALL VARS_4D%SYNC_HOST_RDONLY(FL1=.TRUE., QUEUE=4)
      CALL FF_NOW%SYNC_HOST_RDONLY(QUEUE=4)
      CALL COUNTERS(C1)
      IF (C1(2) - C0(2) /= 15) CALL BAD('SYNC_HOST_RDONLY of FL1 and FF_NOW', C0, C1)      ! FL1 + the 14 members IMPLSCH / NEWWIND may write
      C0 = C1
      CALL WVENVI%SYNC_HOST_RDONLY(QUEUE=5)
      CALL COUNTERS(C1)
      IF (C1(2) /= C0(2)) CALL BAD('WVENVI%SYNC_HOST_RDONLY copied something the device never wrote', C0, C1)
      C0 = C1
      IF (LWNEMOCOU .AND. (.NOT.LWCOU)) CALL WAM2NEMO%SYNC_HOST_RDONLY(QUEUE=6)
      CALL COUNTERS(C1)
      IF (C1(2) - C0(2) /= MERGE(13, 0, LWNEMOCOU .AND. .NOT. LWCOU)) CALL BAD('WAM2NEMO%SYNC_HOST_RDONLY', C0, C1)
      ENDIF
      ILOOP = ILOOP +1
    ENDDO

    IF (KADV == KOUT) THEN
      ! ---- wamodel.F90:374-385: point / spectra output step -----------------------------------------------------------------------------
      CALL COUNTERS(C0)
            CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=4)
            CALL VARS_4D%GET_HOST_DATA_RDONLY(FL1=.TRUE.)
            CALL FF_NOW%GET_HOST_DATA_RDONLY(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
            & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
            & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE.)
      CALL COUNTERS(C1)
      IF (C1(2) /= C0(2)) CALL BAD('GET_HOST_DATA_RDONLY copied again what queue 4 had brought', C0, C1)
      FL1_OUT = VARS_4D%FL1; UFRIC_OUT = FF_NOW%UFRIC      ! (OUTWPSP / OUTSPEC read them in the reference)
    ENDIF

    IF (KADV == NADV) THEN
      ! ---- wamodel.F90:459-468: restart files ------------------------------------------------------------------------------------------
              CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=4)
              CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=5)

              CALL VARS_4D%GET_HOST_DATA_RDONLY(FL1=.TRUE.)
              CALL FF_NOW%GET_HOST_DATA_RDONLY(AIRD=.TRUE., WDWAVE=.TRUE., CICOVER=.TRUE., WSWAVE=.TRUE.,  &
              & WSTAR=.TRUE., UFRIC=.TRUE., TAUW=.TRUE., TAUWDIR=.TRUE., Z0M=.TRUE., Z0B=.TRUE.,  &
              & CHRNCK=.TRUE., CITHICK=.TRUE., USTRA=.TRUE., VSTRA=.TRUE.)
              CALL WVENVI%GET_HOST_DATA_RDONLY(DEPTH=.TRUE., DELLAM1=.TRUE., COSPHM1=.TRUE., UCUR=.TRUE., VCUR=.TRUE., &
              &                                EMAXDPT=.TRUE., IOBND=.TRUE., IODP=.TRUE.)
      ! SAVSPEC -> WRITEFL (writefl.F90:110-118): the restart spectra as one unformatted record, then a second record in the re-labelled
      ! order of a 2-D decomposition (here: the points reversed); both read back and compared (third command argument = the file)
      CALL GET_COMMAND_ARGUMENT(3, FRESTART)
      IF (LEN_TRIM(FRESTART) > 0) THEN
        CALL COUNTERS(C0)
        CALL ECWAM_HIP_WRITEFL(VARS_4D, TRIM(FRESTART), .TRUE.)
        ALLOCATE(IJ2NEWIJ(NPTS))
        IJ2NEWIJ = [(NPTS + 1 - IJ, IJ = 1, NPTS)]
        CALL ECWAM_HIP_WRITEFL(VARS_4D, TRIM(FRESTART), .FALSE., IJ2NEWIJ)
        CALL COUNTERS(C1)
        IF (C1(2) /= C0(2)) CALL BAD('ECWAM_HIP_WRITEFL copied spectra the host already had', C0, C1)
        CALL CHK_4D%ALLOC(UBOUNDS=[NPROMA, NANG, NFRE, NCHNK])
        CALL ECWAM_HIP_READFL(CHK_4D, TRIM(FRESTART))
        IF (ANY(CHK_4D%FL1 /= VARS_4D%FL1)) ERROR STOP 'seam_sequence: the restart record read back differs from the spectra written'
        CHK_4D%FL1 = 0.0_JWRB
        CALL ECWAM_HIP_READFL(CHK_4D, TRIM(FRESTART), IJ2NEWIJ, IREC=2)
        IF (ANY(CHK_4D%FL1 /= VARS_4D%FL1)) ERROR STOP 'seam_sequence: the re-labelled restart record read back differs'
        CALL CHK_4D%DEALLOC()
      ENDIF
    ENDIF

    ! ---- wamodel.F90:607-642: WAM-NEMO coupling without the atmospheric model --------------------------------------------------------------
        IF (LWNEMOCOU .AND. (.NOT.LWCOU)) THEN
          NEMOWSTEP=NEMOWSTEP+1

          IF (MOD(NEMOWSTEP,NEMOFRCO) == 0) THEN

            CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=6)
            CALL WAM2NEMO%GET_HOST_DATA_RDONLY(NEMOUSTOKES=.TRUE., NEMOVSTOKES=.TRUE., NEMOSTRN=.TRUE.,  &
            & NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE.)
            CALL WAM2NEMO%GET_HOST_DATA_RDWR(NEMOTAUX=.TRUE., NEMOTAUY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., &
            &                                NEMOTAUICX=.TRUE., NEMOTAUICY=.TRUE.)

            CALL UPDNEMOSTRESS_RESTATED()

            CALL COUNTERS(C0)
            CALL WAM2NEMO%SYNC_DEVICE_RDWR(NEMOUSTOKES=.TRUE., NEMOVSTOKES=.TRUE., NEMOSTRN=.TRUE.,  &
            & NPHIEPS=.TRUE., NTAUOC=.TRUE., NSWH=.TRUE., NMWP=.TRUE., NEMOTAUX=.TRUE.,  &
            & NEMOTAUY=.TRUE., NEMOWSWAVE=.TRUE., NEMOPHIF=.TRUE., QUEUE=3)
            CALL COUNTERS(C1)
            ! of the eleven members named, the host changed four that the call names (NEMOTAUX, NEMOTAUY, NEMOWSWAVE, NEMOPHIF): those travel;
            ! NEMOTAUICX / Y, changed as well and not named (as in the reference), travel with the next GET_DEVICE_DATA_RDWR of the driver
            IF (C1(1) - C0(1) /= 4 .OR. C1(2) /= C0(2)) CALL BAD('WAM2NEMO%SYNC_DEVICE_RDWR after the coupling step', C0, C1)
            NUP_EXPECTED = 2
          ENDIF
        ENDIF
  ENDDO ADVECTION

  ! ---- wamodel.F90:651-671 --------------------------------------------------------------------------------------------------------------
  CALL COUNTERS(C0)
      CALL WVPRPT%GET_HOST_DATA_RDWR()
      CALL WVENVI%GET_HOST_DATA_RDWR()
      CALL FF_NOW%GET_HOST_DATA_RDWR()
      CALL FF_NEXT%GET_HOST_DATA_RDWR()
      IF(IASSI == 1) CALL WAM2NEMO%GET_HOST_DATA_RDWR()
      CALL INTFLDS%GET_HOST_DATA_RDWR()
      CALL VARS_4D%GET_HOST_DATA_RDWR(FL1=.TRUE.)
      CALL BLK2GLO%GET_HOST_DATA_RDONLY()
  CALL COUNTERS(C1)
  ! what is still newer on the device: the 15 INTGT_PARAM_FIELDS members (nobody has asked for them so far); everything else has come
  ! back through the queues and the restart step above -- except, when the run ends on a coupling step, the eleven WAVE2OCEAN members
  ! wamodel.F90:638 handed back to the device for writing (SYNC_DEVICE_RDWR: FIELD_API's status rule, the device copy counts)
  IF (C1(2) - C0(2) /= 15 + MERGE(11, 0, LWNEMOCOU .AND. .NOT. LWCOU .AND. MOD(NADV, NEMOFRCO) == 0)) CALL BAD('final GET_HOST_DATA_RDWR', C0, C1)
  ! XLLWS and MIJ are not part of the reference's final list (outputs read on the device by OUTBS_LOKI_GPU); the test wants them
  CALL VARS_4D%GET_HOST_DATA_RDONLY(XLLWS=.TRUE.)
  CALL MIJ%GET_HOST_DATA_RDONLY()

      CALL WVPRPT%DELETE_DEVICE_DATA()
      CALL WVENVI%DELETE_DEVICE_DATA()
      CALL FF_NOW%DELETE_DEVICE_DATA()
      CALL FF_NEXT%DELETE_DEVICE_DATA()
      CALL WAM2NEMO%DELETE_DEVICE_DATA()
      CALL INTFLDS%DELETE_DEVICE_DATA()
      CALL VARS_4D%DELETE_DEVICE_DATA()
      CALL MIJ%DELETE_DEVICE_DATA()

  CALL CASE_WRITE(MERGE(NSOURCE, 0_JWIM, LWNEMOCOU))      ! (NEMONTAU itself was reset by the coupling steps: the count is checked there)
  OPEN(NEWUNIT=IU, FILE=TRIM(FOUT), ACCESS='STREAM', FORM='UNFORMATTED', STATUS='OLD', POSITION='APPEND')
  WRITE(IU) INT(NCOUP, C_INT)
  IF (NCOUP > 0) WRITE(IU) STRESS_AVG(:,:,:,1:NCOUP)
  WRITE(IU) FL1_OUT, UFRIC_OUT
  CALL COUNTERS(C1)
  WRITE(IU) C1
  CLOSE(IU)
  CALL ECWAM_HIP_FINALIZE()
  PRINT '(A)', 'seam_sequence: ok'
CONTAINS
  SUBROUTINE COUNTERS(C)
    INTEGER(C_LONG_LONG), INTENT(OUT) :: C(4)
    C = [HIPST%NH2D, HIPST%ND2H, HIPST%BH2D, HIPST%BD2H]
  END SUBROUTINE
  SUBROUTINE BAD(WHAT, CA, CB)
    CHARACTER(LEN=*), INTENT(IN) :: WHAT
    INTEGER(C_LONG_LONG), INTENT(IN) :: CA(4), CB(4)
    WRITE(0,'(A,A,A,2I6,A,2I6)') ' seam_sequence: intent tracking: ', WHAT, ': host->device / device->host copies before ', CA(1:2), ' after ', CB(1:2)
    ERROR STOP 1
  END SUBROUTINE
  ! UPDNEMOSTRESS (updnemostress.F90:84-123): the stresses accumulated over NEMONTAU source-term steps are averaged (and handed to NEMO in
  ! the reference: kept for the test here), then reset for the next accumulation
  SUBROUTINE UPDNEMOSTRESS_RESTATED()
    REAL(KIND=JWRO) :: Z
    IF (HIP_NEMONTAU /= NEMOFRCO) THEN      ! one source-term step per advection step in this harness
      WRITE(0,*) 'seam_sequence: NEMONTAU = ', HIP_NEMONTAU, ' at a coupling step after ', NEMOFRCO, ' source-term steps'
      ERROR STOP 1
    ENDIF
    NCOUP = NCOUP + 1
    Z = 1.0_JWRO / REAL(HIP_NEMONTAU, JWRO)
    STRESS_AVG(:,:,1,NCOUP) = WAM2NEMO%NEMOTAUX * Z; STRESS_AVG(:,:,2,NCOUP) = WAM2NEMO%NEMOTAUY * Z
    STRESS_AVG(:,:,3,NCOUP) = WAM2NEMO%NEMOWSWAVE * Z; STRESS_AVG(:,:,4,NCOUP) = WAM2NEMO%NEMOPHIF * Z
    STRESS_AVG(:,:,5,NCOUP) = WAM2NEMO%NEMOTAUICX * Z; STRESS_AVG(:,:,6,NCOUP) = WAM2NEMO%NEMOTAUICY * Z
    HIP_NEMONTAU = 0
    WAM2NEMO%NEMOTAUX = 0.0_JWRO; WAM2NEMO%NEMOTAUY = 0.0_JWRO; WAM2NEMO%NEMOWSWAVE = 0.0_JWRO; WAM2NEMO%NEMOPHIF = 0.0_JWRO
    WAM2NEMO%NEMOTAUICX = 0.0_JWRO; WAM2NEMO%NEMOTAUICY = 0.0_JWRO
  END SUBROUTINE
END PROGRAM SEAM_SEQUENCE
