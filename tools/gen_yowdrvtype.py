#!/usr/bin/env python3
"""Writes ecwam_amd/fortran/yowdrvtype_hip.F90: the eleven host types of the reference's YOWDRVTYPE (yowdrvtype_config.yml:11-55) with
the method surface drvtype_mod.fypp:56-69 generates for its GPU build -- ALLOC / DEALLOC, SYNC_DEVICE_RDWR / _RDONLY, SYNC_HOST_RDWR /
_RDONLY, GET_DEVICE_DATA_RDWR / _RDONLY / _WRONLY, GET_HOST_DATA_RDWR / _RDONLY, DELETE_DEVICE_DATA, every one with the optional
member selectors and (SYNC_*) QUEUE= of the reference -- over the library's own device layout instead of FIELD_API images.

The table below is this repository's own schema: member names / kinds / ranks are the reference's (they must be: the seam passes these
types), the last column says where a member lives on the device (ecwam_hip_capi.F90::HIP_BLOCK): a block of packed per-point rows and
a column, or nothing (members no kernel of the hot path reads keep their host copy only).

Run:  python tools/gen_yowdrvtype.py      (the generated file is committed; the build does not depend on this script)"""
from __future__ import annotations

import os

I, R, O = "int", "real", "ocean"
DECL = {I: "INTEGER(KIND=JWIM)", R: "REAL(KIND=JWRB)", O: "REAL(KIND=JWRO)"}
ZERO = {I: "0_JWIM", R: "0.0_JWRB", O: "0.0_JWRO"}
EB = {I: "4", R: "INT(STORAGE_SIZE(1.0_JWRB) / 8)", O: "8"}

# FF rows (include/ecwam_hip.h): the 14 FORCING_FIELDS members IMPLSCH / NEWWIND use, then ENVIRONMENT%EMAXDPT, %DEPTH
FF_COL = {n: i for i, n in enumerate("AIRD WDWAVE CICOVER WSWAVE WSTAR USTRA VSTRA UFRIC TAUW TAUWDIR Z0M Z0B CHRNCK CITHICK".split())}
INTF_COL = {n: i for i, n in enumerate("WSEMEAN WSFMEAN USTOKES VSTOKES STRNMS TAUXD TAUYD TAUOCXD TAUOCYD TAUOC TAUICX TAUICY PHIOCD PHIEPS PHIAW".split())}
W2N_COL = {n: i for i, n in enumerate("NEMOUSTOKES NEMOVSTOKES NEMOSTRN NPHIEPS NTAUOC NSWH NMWP NEMOTAUX NEMOTAUY NEMOTAUICX NEMOTAUICY NEMOWSWAVE NEMOPHIF".split())}
WV_COL = {"WAVNUM": 0, "CGROUP": 1, "CINV": 2, "XK2CG": 3, "STOKFAC": 4}


def members(names, kind, where):
    return [(n.upper(), kind, where(n.upper())) for n in names.split()]


# type -> (rank, [(member, kind, (block, column) | None)])     block "ROLE": D_FF for FF_NOW, D_FFN for FF_NEXT
TYPES = {
    "ENVIRONMENT": (2, members("indep iodp iobnd", I, lambda n: ("HIP_BLK_ONES", 0) if n in ("IODP", "IOBND") else None)
                    + members("ibrmem dellam1 cosphm1 depth emaxdpt ucur vcur", R,
                              lambda n: {"IBRMEM": ("HIP_BLK_INTF", 15), "DEPTH": ("HIP_BLK_FF", 15), "EMAXDPT": ("HIP_BLK_FF", 14)}.get(n))),
    "FREQUENCY": (3, members("wavnum cinv cgroup xk2cg omosnh2kd stokfac ciwa", R, lambda n: ("HIP_BLK_WVPRPT", WV_COL[n]) if n in WV_COL else None)),
    "FORCING_FIELDS": (2, members("uwnd vwnd aird wstar cicover cithick lkfr ustra vstra ucur vcur wswave wdwave ufric tauw tauwdir z0m z0b chrnck xlon ylat",
                                  R, lambda n: ("ROLE", FF_COL[n]) if n in FF_COL else None)),
    "WAVE2OCEAN": (2, members("nswh nmwp nphieps nemophif ntauoc nemotaux nemotauy nemoustokes nemovstokes nemostrn nemowswave nemotauicx nemotauicy",
                              O, lambda n: ("HIP_BLK_W2N", W2N_COL[n]))),
    "INTGT_PARAM_FIELDS": (2, members("wsemean wsfmean ustokes vstokes phieps phiocd phiaw tauoc tauxd tauyd tauocxd tauocyd tauicx tauicy strnms altwh caltwh raltcor",
                                      R, lambda n: ("HIP_BLK_INTF", INTF_COL[n]) if n in INTF_COL else None)),
    "WVGRIDGLO": (1, members("ixlg kxlt", I, lambda n: None)),
    "WVGRIDLOC": (2, members("ifromij kfromij jfromij", I, lambda n: None)),
    "FREQUENCY_LAND": (1, members("wavnum cinv cgroup xk2cg omosnh2kd stokfac ciwa", R, lambda n: None)),
    "OCEAN2WAVE": (2, members("nemocicover nemocithick nemoucur nemovcur nemociibr", O, lambda n: None)),
    "TYPE_4D": (4, members("fl1 xllws", R, lambda n: ("HIP_BLK_" + n, 0))),
    "MIJ_TYPE": (2, members("ptr", I, lambda n: ("HIP_BLK_MIJ", 0))),
}

# method -> (operation code, has QUEUE)
METHODS = [("SYNC_DEVICE_RDWR", True), ("SYNC_DEVICE_RDONLY", True), ("SYNC_HOST_RDWR", True), ("SYNC_HOST_RDONLY", True),
           ("GET_DEVICE_DATA_RDWR", False), ("GET_DEVICE_DATA_RDONLY", False), ("GET_DEVICE_DATA_WRONLY", False),
           ("GET_HOST_DATA_RDWR", False), ("GET_HOST_DATA_RDONLY", False)]
OPCODE = {"SYNC_DEVICE_RDWR": "HIP_OP_SYNC_DEVICE_RDWR", "SYNC_DEVICE_RDONLY": "HIP_OP_SYNC_DEVICE_RDONLY", "SYNC_HOST_RDWR": "HIP_OP_SYNC_HOST_RDWR",
          "SYNC_HOST_RDONLY": "HIP_OP_SYNC_HOST_RDONLY", "GET_DEVICE_DATA_RDWR": "HIP_OP_GET_DEVICE_RDWR", "GET_DEVICE_DATA_RDONLY": "HIP_OP_GET_DEVICE_RDONLY",
          "GET_DEVICE_DATA_WRONLY": "HIP_OP_GET_DEVICE_WRONLY", "GET_HOST_DATA_RDWR": "HIP_OP_GET_HOST_RDWR", "GET_HOST_DATA_RDONLY": "HIP_OP_GET_HOST_RDONLY"}


def wrap(items, indent=""):
    """(flang takes free-form lines of any length)"""
    return ", ".join(items)


def gen_type(name, rank, mem):
    n = len(mem)
    dims = ",".join(":" * rank)
    names = [m for m, _, _ in mem]
    out = []
    A = out.append
    A(f"  ! ---- {name} (yowdrvtype_config.yml) " + "-" * max(4, 100 - len(name)))
    A(f"  TYPE {name}")
    for m, k, _ in mem:
        A(f"    {DECL[k]}, POINTER, CONTIGUOUS :: {m}({dims}) => NULL()")
    A("    LOGICAL :: LALLOC = .FALSE.")
    A("    ! status of the members' host and device copies (what the FIELD_API objects F_<member> hold in the reference): a pointer target,")
    A("    ! so that the methods may update it through an INTENT(IN) object, as the reference's own calls on FF_NEXT need")
    A("    TYPE(HIP_FIELD_STATE), POINTER :: HIP => NULL()")
    A("  CONTAINS")
    A(f"    PROCEDURE :: ALLOC => {name}_ALLOC")
    A(f"    PROCEDURE :: DEALLOC => {name}_DEALLOC")
    for meth, _ in METHODS:
        A(f"    PROCEDURE :: {meth} => {name}_{meth}")
    A(f"    PROCEDURE :: DELETE_DEVICE_DATA => {name}_DELETE_DEVICE_DATA")
    A(f"  END TYPE {name}")
    return out, names


def gen_procs(name, rank, mem):
    n = len(mem)
    dims = ",".join(":" * rank)
    names = [m for m, _, _ in mem]
    out = []
    A = out.append
    bnd = ", ".join(f"LL({i}):UBOUNDS({i})" for i in range(1, rank + 1))
    A(f"  SUBROUTINE {name}_ALLOC(SELF, UBOUNDS, LBOUNDS)")
    A(f"    CLASS({name}) :: SELF")
    A(f"    INTEGER(KIND=JWIM), INTENT(IN) :: UBOUNDS({rank})")
    A(f"    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: LBOUNDS({rank})")
    A(f"    INTEGER(KIND=JWIM) :: LL({rank})")
    A("    LL(:) = 1")
    A("    IF (PRESENT(LBOUNDS)) LL = LBOUNDS")
    for m, k, _ in mem:
        A(f"    ALLOCATE(SELF%{m}({bnd}))")
        A(f"    SELF%{m}({dims}) = {ZERO[k]}")
    A("    IF (.NOT. ASSOCIATED(SELF%HIP)) ALLOCATE(SELF%HIP)")
    A("    SELF%HIP%ST(:) = HIP_HOST_FRESH")
    A("    SELF%LALLOC = .TRUE.")
    A(f"  END SUBROUTINE {name}_ALLOC")
    A("")
    A(f"  SUBROUTINE {name}_DEALLOC(SELF)")
    A(f"    CLASS({name}) :: SELF")
    for m, _, _ in mem:
        A(f"    IF (ASSOCIATED(SELF%{m})) DEALLOCATE(SELF%{m})")
        A(f"    NULLIFY(SELF%{m})")
    A("    IF (ASSOCIATED(SELF%HIP)) DEALLOCATE(SELF%HIP)")
    A("    NULLIFY(SELF%HIP)")
    A("    SELF%LALLOC = .FALSE.")
    A(f"  END SUBROUTINE {name}_DEALLOC")
    A("")
    # selection: drvtype_mod.fypp:118-134 -- L_X = X where given, .FALSE. otherwise; no flag true = the entire structure
    arglist = wrap(names, "    ")
    A(f"  SUBROUTINE {name}_SELECT(L, {arglist})")
    A(f"    LOGICAL, INTENT(OUT) :: L({n})")
    A(f"    LOGICAL, INTENT(IN), OPTIONAL :: {wrap(names, '    ')}")
    A("    L(:) = .FALSE.")
    for i, m in enumerate(names, 1):
        A(f"    IF (PRESENT({m})) L({i}) = {m}")
    A("    IF (.NOT. ANY(L)) L(:) = .TRUE.")
    A(f"  END SUBROUTINE {name}_SELECT")
    A("")
    A(f"  SUBROUTINE {name}_XFER(SELF, L, IOP, QUEUE)")
    A(f"    CLASS({name}) :: SELF")
    A(f"    LOGICAL, INTENT(IN) :: L({n})")
    A("    INTEGER, INTENT(IN) :: IOP")
    A("    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: QUEUE")
    if any(w and w[0] == "ROLE" for _, _, w in mem):
        A("    INTEGER :: IB")
    if all(w is None for _, _, w in mem):
        A("    ! no member of this type has a device copy (nothing on the hot path reads them on the device): every method accepts its selectors")
        A("    ! and leaves the host copies, the only ones, alone")
    else:
        A(f"    IF (.NOT. ASSOCIATED(SELF%HIP)) CALL HIP_FATAL('{name}: ALLOC has not been called on this object')")
    if any(w and w[0] == "ROLE" for _, _, w in mem):
        A("    IB = HIP_BLK_NONE      ! an object ECWAM_HIP_BIND_FORCING / WAMINTGR_HIP has not seen yet keeps its host copy: the transfer happens at the")
        A("    IF (SELF%HIP%ROLE == 1) IB = HIP_BLK_FF      ! first GET_DEVICE_DATA_* inside WAMINTGR_HIP")
        A("    IF (SELF%HIP%ROLE == 2) IB = HIP_BLK_FFN")
    for i, (m, k, w) in enumerate(mem, 1):
        if w is None:
            continue      # no device copy: the host copy is always the valid one, every method leaves it alone
        blk, col = w
        if blk in ("HIP_BLK_FL1", "HIP_BLK_XLLWS"):
            A(f"    IF (L({i})) CALL HIP_SPECTRUM_XFER(SELF%HIP%ST({i}), IOP, {blk}, HIP_LOC{rank}{k[0].upper()}(SELF%{m}), QUEUE)")
        elif blk == "HIP_BLK_ONES":
            A(f"    IF (L({i})) CALL HIP_CHECK_ONES(IOP, HIP_LOC{rank}{k[0].upper()}(SELF%{m}), '{m}')")
        else:
            b = "IB" if blk == "ROLE" else blk
            A(f"    IF (L({i})) CALL HIP_MEMBER_XFER(SELF%HIP%ST({i}), IOP, {b}, {col}, {'.TRUE.' if rank == 3 else '.FALSE.'}, {EB[k]}, "
              f"HIP_LOC{rank}{k[0].upper()}(SELF%{m}), QUEUE)")
    A(f"  END SUBROUTINE {name}_XFER")
    A("")
    for meth, hasq in METHODS:
        q = ", QUEUE" if hasq else ""
        A(f"  SUBROUTINE {name}_{meth}(SELF, {wrap(names, '    ')}{q})")
        A(f"    CLASS({name}) :: SELF")
        A(f"    LOGICAL, INTENT(IN), OPTIONAL :: {wrap(names, '    ')}")
        if hasq:
            A("    INTEGER(KIND=JWIM), INTENT(IN), OPTIONAL :: QUEUE")
        A(f"    LOGICAL :: L({n})")
        A(f"    CALL {name}_SELECT(L, {wrap(names, '    ')})")
        A(f"    CALL {name}_XFER(SELF, L, {OPCODE[meth]}{', QUEUE' if hasq else ''})")
        A(f"  END SUBROUTINE {name}_{meth}")
        A("")
    A(f"  SUBROUTINE {name}_DELETE_DEVICE_DATA(SELF)")
    A(f"    CLASS({name}) :: SELF")
    A(f"    LOGICAL :: L({n})")
    A("    L(:) = .TRUE.")
    A(f"    CALL {name}_XFER(SELF, L, HIP_OP_DELETE_DEVICE)")
    A(f"  END SUBROUTINE {name}_DELETE_DEVICE_DATA")
    A("")
    return out


HEADER = '''! yowdrvtype_hip.F90 -- GENERATED by tools/gen_yowdrvtype.py; do not edit by hand.
!
! The host types of the WAMINTGR seam for a build outside ecWAM: member names, kinds and ranks of the reference's YOWDRVTYPE
! (yowdrvtype_config.yml:11-55) and the complete method surface its GPU build gives them (drvtype_mod.fypp:56-69,116-480): ALLOC /
! DEALLOC, SYNC_DEVICE_RDWR / _RDONLY, SYNC_HOST_RDWR / _RDONLY (optional member selectors + QUEUE), GET_DEVICE_DATA_RDWR / _RDONLY /
! _WRONLY, GET_HOST_DATA_RDWR / _RDONLY (member selectors), DELETE_DEVICE_DATA -- so that the reference's own call lines
! (wamodel.F90:207-226,376-385,435-470,614-642,651-671, wamintgr_loki_gpu.F90:100-157,197-200) compile and run as they stand.
!
! Differences from FIELD_API, all behind the same calls:
!   * the device copies are the library's packed per-point rows (include/ecwam_hip.h), not images of the host arrays: a transfer is a
!     plain copy of the member into a scratch image on the asked queue + ecwam_hip_member_scatter / _gather into / out of the member's
!     column (the spectra: ecwam_hip_chunks_to_points / _points_to_chunks);
!   * the member pointers always point at the HOST arrays (GET_DEVICE_DATA_* does not re-point them: no Fortran code touches device
!     memory here);
!   * members no kernel of the hot path reads (UWND, XLON, CIWA, ALTWH, WVGRIDLOC ...) have no device copy: the methods accept their
!     selectors and leave them alone;
!   * per member FIELD_API's status (host copy fresh / device copy fresh) decides whether a call copies anything: SYNC_HOST_* /
!     GET_HOST_DATA_* move only what the device wrote since the host last had it, SYNC_DEVICE_* / GET_DEVICE_DATA_* only what the host
!     changed (GET_HOST_DATA_RDWR / SYNC_HOST_RDWR say that it will).
! Inside ecWAM the real YOWDRVTYPE / FIELD_API are used instead (INTEGRATION.md).
MODULE PARKIND_WAVE      ! parkind_wave.F90:23-35
  USE ECWAM_HIP_CAPI, ONLY : JWIM, JWRB, JWRO
  USE, INTRINSIC :: ISO_C_BINDING, ONLY : C_DOUBLE
  IMPLICIT NONE
  INTEGER, PARAMETER :: JWRU = C_DOUBLE
END MODULE PARKIND_WAVE

MODULE FIELD_ASYNC_MODULE      ! field_api: WAIT_FOR_ASYNC_QUEUE(QUEUE) = the end of everything posted on that queue
  USE ECWAM_HIP_CAPI, ONLY : WAIT_FOR_ASYNC_QUEUE
  IMPLICIT NONE
END MODULE FIELD_ASYNC_MODULE

MODULE YOWDRVTYPE
  USE, INTRINSIC :: ISO_C_BINDING
  USE ECWAM_HIP_CAPI
  IMPLICIT NONE
  PRIVATE
'''


def main():
    out = [HEADER.rstrip("\n")]
    out.append("  PUBLIC :: " + ", ".join(TYPES) )
    out.append("  PUBLIC :: ECWAM_HIP_BIND_FORCING")
    out.append("")
    for name, (rank, mem) in TYPES.items():
        t, _ = gen_type(name, rank, mem)
        out += t
        out.append("")
    out.append("CONTAINS")
    out.append("")
    out.append("  ! C_LOC of a member (C_NULL_PTR when it is not associated: an object whose members were never allocated)")
    for rank in (1, 2, 3, 4):
        for k in (I, R, O):
            dims = ",".join(":" * rank)
            fn = f"HIP_LOC{rank}{k[0].upper()}"
            out += [f"  FUNCTION {fn}(A) RESULT(P)",
                    f"    {DECL[k]}, POINTER, CONTIGUOUS, INTENT(IN) :: A({dims})",
                    "    TYPE(C_PTR) :: P",
                    "    P = C_NULL_PTR",
                    "    IF (ASSOCIATED(A)) THEN",
                    "      IF (SIZE(A) > 0) P = C_LOC(A)",
                    "    ENDIF",
                    f"  END FUNCTION {fn}", ""]
    out += ["  ! Which device rows the two FORCING_FIELDS objects of the seam use: FF_NOW the rows IMPLSCH works on, FF_NEXT the rows NEWWIND reads.",
            "  ! WAMINTGR_HIP calls it on entry; a host that wants its first SYNC_DEVICE_* (wamodel.F90:215-220) to start the copies at once calls it",
            "  ! after ECWAM_HIP_SETUP -- before it the objects are unbound and a SYNC_DEVICE_* leaves the transfer to the first GET_DEVICE_DATA_*.",
            "  SUBROUTINE ECWAM_HIP_BIND_FORCING(FF_NOW, FF_NEXT)",
            "    TYPE(FORCING_FIELDS), INTENT(IN) :: FF_NOW, FF_NEXT",
            "    IF (.NOT. ASSOCIATED(FF_NOW%HIP) .OR. .NOT. ASSOCIATED(FF_NEXT%HIP)) CALL HIP_FATAL('FORCING_FIELDS: ALLOC has not been called on this object')",
            "    CALL REBIND(FF_NOW%HIP, 1); CALL REBIND(FF_NEXT%HIP, 2)",
            "  CONTAINS",
            "    ! An object that changes its role (the host swapped FF_NOW and FF_NEXT, or passes another object): the status of its members",
            "    ! describes the rows of the OLD role.  A member whose only valid copy is on the device under the old role cannot be carried over",
            "    ! (the rows now belong to the other object): the host has to GET_HOST_DATA_* it first.  Otherwise the host copies are the valid",
            "    ! ones and the next device access uploads them into the rows of the new role.",
            "    SUBROUTINE REBIND(H, IROLE)",
            "      TYPE(HIP_FIELD_STATE), INTENT(INOUT) :: H",
            "      INTEGER, INTENT(IN) :: IROLE",
            "      IF (H%ROLE /= 0 .AND. H%ROLE /= IROLE) THEN",
            "        IF (ANY(IAND(H%ST, HIP_HOST_FRESH) == 0)) CALL HIP_FATAL('FORCING_FIELDS: the object changes its role (FF_NOW <-> FF_NEXT) while ' // &",
            "   &      'members are valid on the device only: call GET_HOST_DATA_* on it before WAMINTGR_HIP')",
            "        H%ST(:) = HIP_HOST_FRESH",
            "      ENDIF",
            "      H%ROLE = IROLE",
            "    END SUBROUTINE REBIND",
            "  END SUBROUTINE ECWAM_HIP_BIND_FORCING", ""]
    for name, (rank, mem) in TYPES.items():
        out += gen_procs(name, rank, mem)
    out.append("END MODULE YOWDRVTYPE")
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ecwam_amd", "fortran", "yowdrvtype_hip.F90")
    with open(path, "w") as fh:
        fh.write("\n".join(out) + "\n")
    print(os.path.normpath(path), len(out), "lines")


if __name__ == "__main__":
    main()
