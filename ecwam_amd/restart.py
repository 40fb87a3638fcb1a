"""Restart spectra in the reference's binary layout.

`writefl.F90:110-118` writes ONE Fortran unformatted sequential record per rank:
``WRITE(IUNIT) (((FL(IJ,K,M),IJ=IJINF,IJSUP),K=KINF,KSUP),M=MINF,MSUP)`` -- IJ fastest, then K, then M, reals in the
working precision -- and `getspec`/`readfl` read it back the same way.  The device keeps spectra point-major
(`FL[ij][K][M]`, include/ecwam_hip.h), so the conversion is a transpose of the three axes.

Record framing is the 4-byte length-marker convention of gfortran/flang/ifort (`-assume byterecl` not needed for
sequential files): ``<int32 nbytes> payload <int32 nbytes>``.  Payloads beyond 2^31-9 bytes use gfortran's sub-record
scheme (every sub-record framed, a negative head marker = "continued", a negative tail marker = "has a predecessor"),
which is what a gfortran-built ecWAM produces for the 2.18 GB single-rank O320 record.
"""
from __future__ import annotations

import numpy as np

_MAX_SUB = 2147483639  # gfortran: GFC_MAX_SUBRECORD_LENGTH


def write_fl(path: str, fl_points: np.ndarray, append: bool = False) -> None:
    """fl_points: [n][NANG][NFRE] (owned rows only, device layout).  Appends like IWAM_GET_UNIT(...,'a',...) when asked."""
    a = np.ascontiguousarray(np.transpose(fl_points, (2, 1, 0)))  # C order [M][K][IJ] == Fortran FL(IJ,K,M) element order
    raw = a.reshape(-1).view(np.uint8)
    nb = raw.size
    with open(path, "ab" if append else "wb") as f:
        if nb <= _MAX_SUB:
            m = np.array([nb], dtype="<i4").tobytes()
            f.write(m); f.write(raw.tobytes() if nb < (1 << 26) else memoryview(raw)); f.write(m)
            return
        off, first = 0, True
        while off < nb:
            ln = min(_MAX_SUB, nb - off)
            last = off + ln >= nb
            head = np.array([ln if last else -ln], dtype="<i4").tobytes()
            tail = np.array([ln if first else -ln], dtype="<i4").tobytes()
            f.write(head); f.write(memoryview(raw[off:off + ln])); f.write(tail)
            off += ln
            first = False


def read_fl(path: str, n: int, nang: int, nfre: int, dtype, record: int = 0) -> np.ndarray:
    """Reads record number `record` (0-based) and returns [n][NANG][NFRE]."""
    dt = np.dtype(dtype)
    want = n * nang * nfre * dt.itemsize
    with open(path, "rb") as f:
        for rec in range(record + 1):
            chunks, total = [], 0
            while True:
                h = f.read(4)
                if len(h) != 4:
                    raise EOFError(f"{path}: record {rec} not found")
                head = int(np.frombuffer(h, "<i4")[0])
                ln = abs(head)
                if rec == record:
                    chunks.append(f.read(ln))
                else:
                    f.seek(ln, 1)
                total += ln
                tail = int(np.frombuffer(f.read(4), "<i4")[0])
                if abs(tail) != ln:
                    raise ValueError(f"{path}: corrupt record markers ({head} / {tail})")
                if head >= 0:  # last (or only) sub-record
                    break
            if rec == record:
                if total != want:
                    raise ValueError(f"{path}: record holds {total} bytes, expected {want} for FL({n},{nang},{nfre}) {dt}")
                a = np.frombuffer(b"".join(chunks), dtype=dt).reshape(nfre, nang, n)
                return np.ascontiguousarray(np.transpose(a, (2, 1, 0)))
    raise AssertionError("unreachable")
