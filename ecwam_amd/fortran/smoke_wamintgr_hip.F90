! smoke_wamintgr_hip.F90 -- harness program (this repository's own, not part of ecWAM): reads a case written by
! tests/test_gpu_fortran.py (harness_case.F90), calls NSTEP x WAMINTGR_HIP with the reference's WAMODEL date sequence
! (wamodel.F90:228-312: CDTPRA=CDTPRO; CDTPRO+=IDELPRO; loop while CDTIMPNEXT<=CDTPRO), fetches the results through the
! FIELD_API-shaped methods of the host types and writes the host arrays.  Dates are seconds counters in 14-character strings.
! (seam_sequence.F90 is the same run inside the complete call sequence of the reference's GPU build.)
PROGRAM SMOKE_WAMINTGR_HIP
  USE, INTRINSIC :: ISO_C_BINDING
  USE ECWAM_HIP_MOD
  USE ECWAM_HIP_DRV
  USE ECWAM_HIP_HOSTSYNC
  USE HARNESS_CASE
  IMPLICIT NONE
  INTERFACE
    SUBROUTINE WAMINTGR_HIP (CDTPRA, CDATE, CDATEWH, CDTIMP, CDTIMPNEXT, BLK2GLO, WVENVI, WVPRPT, FF_NOW, FF_NEXT, INTFLDS, &
 &                           WAM2NEMO, MIJ, VARS_4D)
      USE ECWAM_HIP_MOD
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
  INTEGER :: ISTEP, ILOOP
  INTEGER, PARAMETER :: NWARM = 2
  INTEGER(KIND=8) :: T0, T1, TRATE

  CALL CASE_READ_AND_SETUP()

  ! the steps leave everything on the device (no per-step copies); the host types are refreshed once, after the last step.
  ! HDR(19) = 1: LLSOURCE = .FALSE. (the no-source branch of WAMINTGR); a third command argument "time": print the time per step
  HIP_IDELT = IDELT; HIP_INCDATE => SIMPLE_INCDATE; HIP_LSYNC_SPECTRA = .FALSE.; HIP_LSYNC_FIELDS = .FALSE.
  HIP_LLSOURCE = (HDR(19) == 0)
  CALL GET_COMMAND_ARGUMENT(3, FMODE)
  CALL SYSTEM_CLOCK(T0, TRATE)
  CDTPRO = '00000000000000'; CDTIMP = CDTPRO; CDATEWH = '99999999999999'
  NSOURCE = 0; HIP_NEMONTAU = 0
  CDTIMPNEXT = CDTPRO; CALL SIMPLE_INCDATE(CDTIMPNEXT, IDELT)        ! wamodel.F90:182-185
  DO ISTEP = 1, NSTEP            ! wamodel.F90:228-312
    CDTPRA = CDTPRO
    CALL SIMPLE_INCDATE(CDTPRO, IDELPRO)
    HIP_CDTPRO = CDTPRO
    CDATE = CDTPRA
    ILOOP = 1
    DO WHILE (ILOOP == 1 .OR. CDTIMPNEXT <= CDTPRO)
      IF ((ILOOP > 1 .OR. CDTPRO >= CDTIMPNEXT) .AND. HIP_LLSOURCE) NSOURCE = NSOURCE + 1      ! this call integrates the source terms
      CALL WAMINTGR_HIP(CDTPRA, CDATE, CDATEWH, CDTIMP, CDTIMPNEXT, BLK2GLO, WVENVI, WVPRPT, FF_NOW, FF_NEXT, INTFLDS, WAM2NEMO, &
 &                      MIJ, VARS_4D)
      ILOOP = ILOOP + 1
    ENDDO
    IF (ISTEP == NWARM) THEN       ! timing mode: the first NWARM steps (first-call uploads, clocks) are not timed
      CALL ECWAM_HIP_CHECK(ECWAM_HIP_SYNC(HIPST%CTX, HIPST%QUEUE(0)), 'SYNC')
      CALL SYSTEM_CLOCK(T0, TRATE)
    ENDIF
  ENDDO
  CALL ECWAM_HIP_CHECK(ECWAM_HIP_SYNC(HIPST%CTX, HIPST%QUEUE(0)), 'SYNC')
  CALL SYSTEM_CLOCK(T1)
  IF (TRIM(FMODE) == 'time' .AND. NSTEP > NWARM) PRINT '(A,F10.3,A,I0,A,I0,A)', 'smoke_wamintgr_hip: ', &
 &   1000.0D0 * REAL(T1 - T0, 8) / REAL(TRATE, 8) / REAL(NSTEP - NWARM, 8), ' ms per WAMINTGR_HIP step (', NSTEP - NWARM, ' steps, ', NPTS, ' points)'
  ! the host types back, the way the reference's GPU build asks FIELD_API for them (wamintgr_loki_gpu.F90:197-200, wamodel.F90:376):
  ! member selectors, asynchronous queues, one wait per queue.  (ECWAM_HIP_SYNC_HOST is the whole-type, synchronous shorthand.)
  CALL VARS_4D%SYNC_HOST_RDONLY(FL1=.TRUE., QUEUE=4)
  CALL VARS_4D%SYNC_HOST_RDONLY(XLLWS=.TRUE., QUEUE=4)
  CALL FF_NOW%SYNC_HOST_RDONLY(QUEUE=4)
  CALL INTFLDS%SYNC_HOST_RDONLY(WSEMEAN=.TRUE., WSFMEAN=.TRUE., USTOKES=.TRUE., VSTOKES=.TRUE., STRNMS=.TRUE., TAUXD=.TRUE., TAUYD=.TRUE., &
 &                              TAUOCXD=.TRUE., TAUOCYD=.TRUE., TAUOC=.TRUE., TAUICX=.TRUE., TAUICY=.TRUE., PHIOCD=.TRUE., PHIEPS=.TRUE., &
 &                              PHIAW=.TRUE., QUEUE=5)
  CALL MIJ%SYNC_HOST_RDONLY(QUEUE=5)
  IF (P%LWNEMOCOU /= 0) CALL WAM2NEMO%SYNC_HOST_RDONLY(QUEUE=6)
  CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=4)
  CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=5)
  CALL WAIT_FOR_ASYNC_QUEUE(QUEUE=6)

  CALL CASE_WRITE(HIP_NEMONTAU)
  CALL ECWAM_HIP_FINALIZE()
  PRINT '(A)', 'smoke_wamintgr_hip: ok'
END PROGRAM SMOKE_WAMINTGR_HIP
