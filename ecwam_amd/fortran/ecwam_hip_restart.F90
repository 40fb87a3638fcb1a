! ecwam_hip_restart.F90 -- restart spectra in the reference's binary layout on the Fortran host side of the boundary (SURVEY.md 8f rank 3).
!
! WRITEFL (writefl.F90:110-118) writes ONE unformatted sequential record per file,
!     WRITE(IUNIT) (((FL(IJ,K,M), IJ=IJINF,IJSUP), K=KINF,KSUP), M=MINF,MSUP)
! -- the block-shaped spectra, IJ fastest, in the working precision -- and, on a 2-D decomposition, in the OLD sea-point numbering
! (FL(IJ2NEWIJ(IJ),K,M): :113-117); READFL / GETSPEC read the record back the same way.  The host type VARS_4D holds the spectra in chunks
! (NPROMA,NANG,NFRE,NCHNK), the device point-major: ECWAM_HIP_WRITEFL fetches what the device holds newer (GET_HOST_DATA_RDONLY: nothing
! moves if the host copy is current), un-chunks and writes the record; ECWAM_HIP_READFL reads one into VARS_4D%FL1 and leaves the
! host copy the valid one (GET_HOST_DATA_RDWR: the next WAMINTGR_HIP sends it up).  ecwam_amd/restart.py is the same layout for the
! Python host; tests/test_gpu_fortran.py reads the Fortran-written file with it.
MODULE ECWAM_HIP_RESTART
  USE ECWAM_HIP_CAPI, ONLY : JWIM, JWRB, HIPST, HIP_FATAL
  USE YOWDRVTYPE, ONLY : TYPE_4D
  IMPLICIT NONE
  PRIVATE
  PUBLIC :: ECWAM_HIP_WRITEFL, ECWAM_HIP_READFL
CONTAINS
  ! LOUNIT: .TRUE. = a new file ('w'), .FALSE. = append a record ('a'), as WRITEFL's IWAM_GET_UNIT modes.  IJ2NEWIJ(1:NPTS) (optional):
  ! the re-labelling of a 2-D decomposition (mpdecomp.F90:478-653) -- record position IJ holds point IJ2NEWIJ(IJ) of VARS_4D.
  SUBROUTINE ECWAM_HIP_WRITEFL(VARS_4D, FILENAME, LOUNIT, IJ2NEWIJ)
    TYPE(TYPE_4D), INTENT(INOUT) :: VARS_4D
    CHARACTER(LEN=*), INTENT(IN) :: FILENAME
    LOGICAL, INTENT(IN) :: LOUNIT
    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: IJ2NEWIJ(:)
    REAL(KIND=JWRB), ALLOCATABLE :: RFL(:,:,:)
    INTEGER :: IU, IJ, K, M, NPTS, NPROMA, NANG, NFRE, IOS
    CALL VARS_4D%GET_HOST_DATA_RDONLY(FL1=.TRUE.)
    CALL UNCHUNK(VARS_4D, RFL, IJ2NEWIJ)
    NPTS = SIZE(RFL, 1); NANG = SIZE(RFL, 2); NFRE = SIZE(RFL, 3)
    IF (LOUNIT) THEN
      OPEN(NEWUNIT=IU, FILE=TRIM(FILENAME), FORM='UNFORMATTED', ACCESS='SEQUENTIAL', STATUS='REPLACE', ACTION='WRITE', IOSTAT=IOS)
    ELSE
      OPEN(NEWUNIT=IU, FILE=TRIM(FILENAME), FORM='UNFORMATTED', ACCESS='SEQUENTIAL', STATUS='UNKNOWN', POSITION='APPEND', ACTION='WRITE', IOSTAT=IOS)
    ENDIF
    IF (IOS /= 0) CALL HIP_FATAL('ECWAM_HIP_WRITEFL: cannot open '//TRIM(FILENAME))
    WRITE(IU) (((RFL(IJ,K,M), IJ=1,NPTS), K=1,NANG), M=1,NFRE)
    CLOSE(IU)
  END SUBROUTINE ECWAM_HIP_WRITEFL

  ! Reads record number IREC (1-based; default 1) of FILENAME into VARS_4D%FL1 (pad lanes of the last chunk: the chunk's lane 1).
  SUBROUTINE ECWAM_HIP_READFL(VARS_4D, FILENAME, IJ2NEWIJ, IREC)
    TYPE(TYPE_4D), INTENT(INOUT) :: VARS_4D
    CHARACTER(LEN=*), INTENT(IN) :: FILENAME
    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: IJ2NEWIJ(:)
    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: IREC
    REAL(KIND=JWRB), ALLOCATABLE :: RFL(:,:,:)
    INTEGER :: IU, IJ, K, M, NPTS, NPROMA, NCHNK, NANG, NFRE, IOS, IR, ICH, IP, JN
    NPROMA = SIZE(VARS_4D%FL1, 1); NANG = SIZE(VARS_4D%FL1, 2); NFRE = SIZE(VARS_4D%FL1, 3); NCHNK = SIZE(VARS_4D%FL1, 4)
    NPTS = NPOINTS(NPROMA, NCHNK)
    ALLOCATE(RFL(NPTS, NANG, NFRE))
    OPEN(NEWUNIT=IU, FILE=TRIM(FILENAME), FORM='UNFORMATTED', ACCESS='SEQUENTIAL', STATUS='OLD', ACTION='READ', IOSTAT=IOS)
    IF (IOS /= 0) CALL HIP_FATAL('ECWAM_HIP_READFL: cannot open '//TRIM(FILENAME))
    IF (PRESENT(IREC)) THEN
      DO IR = 1, IREC - 1
        READ(IU)
      ENDDO
    ENDIF
    READ(IU, IOSTAT=IOS) (((RFL(IJ,K,M), IJ=1,NPTS), K=1,NANG), M=1,NFRE)
    IF (IOS /= 0) CALL HIP_FATAL('ECWAM_HIP_READFL: record too short for FL(NPTS,NANG,NFRE) in '//TRIM(FILENAME))
    CLOSE(IU)
    CALL VARS_4D%GET_HOST_DATA_RDWR(FL1=.TRUE.)      ! the host is about to write FL1: its copy is the valid one from here on
    DO M = 1, NFRE
      DO K = 1, NANG
        DO IJ = 1, NPTS
          JN = IJ
          IF (PRESENT(IJ2NEWIJ)) JN = IJ2NEWIJ(IJ)
          ICH = (JN - 1) / NPROMA + 1; IP = JN - (ICH - 1) * NPROMA
          VARS_4D%FL1(IP,K,M,ICH) = RFL(IJ,K,M)
        ENDDO
        DO IP = NPTS - (NCHNK - 1) * NPROMA + 1, NPROMA      ! pad lanes (propag_wam.F90:388-398)
          VARS_4D%FL1(IP,K,M,NCHNK) = VARS_4D%FL1(1,K,M,NCHNK)
        ENDDO
      ENDDO
    ENDDO
  END SUBROUTINE ECWAM_HIP_READFL

  INTEGER FUNCTION NPOINTS(NPROMA, NCHNK)
    INTEGER, INTENT(IN) :: NPROMA, NCHNK
    NPOINTS = NPROMA * NCHNK
    IF (HIPST%NPTS > 0 .AND. HIPST%NPROMA == NPROMA .AND. HIPST%NCHNK == NCHNK) NPOINTS = HIPST%NPTS      ! the ragged last chunk
  END FUNCTION

  SUBROUTINE UNCHUNK(VARS_4D, RFL, IJ2NEWIJ)
    TYPE(TYPE_4D), INTENT(IN) :: VARS_4D
    REAL(KIND=JWRB), ALLOCATABLE, INTENT(OUT) :: RFL(:,:,:)
    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: IJ2NEWIJ(:)
    INTEGER :: IJ, K, M, NPTS, NPROMA, NCHNK, NANG, NFRE, ICH, IP, JN
    NPROMA = SIZE(VARS_4D%FL1, 1); NANG = SIZE(VARS_4D%FL1, 2); NFRE = SIZE(VARS_4D%FL1, 3); NCHNK = SIZE(VARS_4D%FL1, 4)
    NPTS = NPOINTS(NPROMA, NCHNK)
    ALLOCATE(RFL(NPTS, NANG, NFRE))
    DO M = 1, NFRE
      DO K = 1, NANG
        DO IJ = 1, NPTS
          JN = IJ
          IF (PRESENT(IJ2NEWIJ)) JN = IJ2NEWIJ(IJ)
          ICH = (JN - 1) / NPROMA + 1; IP = JN - (ICH - 1) * NPROMA
          RFL(IJ,K,M) = VARS_4D%FL1(IP,K,M,ICH)
        ENDDO
      ENDDO
    ENDDO
  END SUBROUTINE UNCHUNK
END MODULE ECWAM_HIP_RESTART
