"""Synthetic model grid for the hot path: octahedral reduced grid, sea-point list, neighbour tables.

The reference gets these tables from PREPROC (``wam_grid_tables``), which is out of scope
(SURVEY.md section 2 row 18); the hot path only *reads* them (YOWUBUF KLAT/KLON/KCOR/WLAT/WCOR,
YOWMAP ZDELLO/XDELLA, YOWGRID COSPH/SINPH, BLK2GLO IXLG/KXLT).  This module rebuilds them for a
synthetic grid with the reference's rules so that tests and the benchmark have valid inputs:

  grid definition   share/ecwam/scripts/ecwam_grids.py:71-114  (O<N>: rows 20+4j points, mirrored)
  row geometry      readmdlconf.F90:136-164 (COSPH/SINPH, pole clamp XLATMAX=87.5, ZDELLO=360/NLONRGG)
  aqua mask         programs/preproc.F90:337-346 (all sea, first/last row land)
  neighbours        propconnect.F90:69-200 (KLAT), :165-200 (KLON), :205-420 (KCOR), :655-900 (WLAT/WCOR)
  land index        "NSUP+1" (mpdecomp.F90:89-100): 0-based index ``nland`` = number of local points

Point indices are 0-based; sea points are numbered south->north, west->east.  The nearest-point
rule NINT((i-1)*ZDELLO(k)/ZDELLO(k')) is evaluated in exact integer arithmetic.
"""
from __future__ import annotations

import dataclasses

import numpy as np

LAT0 = {32: 87.863798839233, 48: 88.572168514007, 64: 88.927735352296, 96: 89.284227532514, 320: 89.784876907219,
        640: 89.892396445590, 1280: 89.946187715666}


def _lat0(n: int) -> float:
    if n in LAT0:
        return LAT0[n]
    # first Gaussian latitude is not tabulated for this N in ecwam_grids.py: use the asymptotic formula
    return 90.0 - 180.0 / np.pi * 2.404825557695773 / (2 * n + 0.5)  # first zero of J0 / (N+1/2)


def _nint_div(num: np.ndarray, den: np.ndarray) -> np.ndarray:
    """NINT(num/den) for integer arrays, den > 0 (round half away from zero)."""
    a = np.abs(num)
    q = (2 * a + den) // (2 * den)
    return np.where(num < 0, -q, q)


@dataclasses.dataclass
class Grid:
    name: str
    ngy: int
    nlonrgg: np.ndarray       # [ngy]
    xdella: float             # latitude increment [deg]
    zdello: np.ndarray        # [ngy] longitude increment per row [deg]
    amosop: float
    cosph: np.ndarray         # [ngy]
    sinph: np.ndarray         # [ngy]
    nsea: int
    ixlg: np.ndarray          # [nsea] 0-based longitude index in its row
    kxlt: np.ndarray          # [nsea] 0-based row index
    klon: np.ndarray          # [nsea][2]      west, east
    klat: np.ndarray          # [nsea][2][2]   (south, north) x (closest, second closest)
    kcor: np.ndarray          # [nsea][4][2]   (NE, SE, SW, NW) x (closest, second closest)
    wlat: np.ndarray          # [nsea][2]
    wcor: np.ndarray          # [nsea][4]
    row_start: np.ndarray     # [ngy+1] first sea index of each row (rows without sea: empty range)

    @property
    def nland(self) -> int:
        return self.nsea

    @property
    def cosphm1_ext(self) -> np.ndarray:
        """1/COSPH per point plus the land slot (proenvhalo.F90:81,103: land value 0)."""
        out = np.zeros(self.nsea + 1)
        out[: self.nsea] = 1.0 / self.cosph[self.kxlt]
        return out

    @property
    def lat_deg(self) -> np.ndarray:
        return self.amosop + self.kxlt * self.xdella


def build_grid(n: int, mask: str = "aqua", seed: int = 20230101, land_fraction: float = 0.35) -> Grid:
    """O<n> grid.  mask = "aqua" (reference LAQUA) or "continents" (seeded pseudo-continents)."""
    ngy = 2 * n
    j = np.arange(ngy)
    nlon = np.where(j < n, 20 + 4 * j, 20 + 4 * (ngy - 1 - j)).astype(np.int64)
    north = _lat0(n)
    south = -north
    xdella = (north - south) / (ngy - 1)
    zdello = 360.0 / nlon
    lat = (south + j * xdella) * (np.pi / 180.0)
    cosph, sinph = np.cos(lat), np.sin(lat)
    cmin = np.cos(87.5 * np.pi / 180.0)
    clamp = cosph <= cmin
    cosph = np.where(clamp, np.cos(87.5 * np.pi / 180.0), cosph)
    sinph = np.where(clamp, np.sin(87.5 * np.pi / 180.0), sinph)

    # ---- ocean mask per (row, lon)
    off = np.concatenate([[0], np.cumsum(nlon)])  # offsets into the flattened full grid
    ntot = int(off[-1])
    row_of = np.repeat(j, nlon)
    i_of = np.arange(ntot) - off[row_of]
    ocean = np.ones(ntot, dtype=bool)
    ocean[row_of == 0] = False
    ocean[row_of == ngy - 1] = False
    if mask == "continents":
        rng = np.random.default_rng(seed)
        # low-pass random field on the sphere from a few random plane waves, thresholded
        lam = 2 * np.pi * (i_of + 0.0) / nlon[row_of]
        phi = lat[row_of]
        xyz = np.stack([np.cos(phi) * np.cos(lam), np.cos(phi) * np.sin(lam), np.sin(phi)], 1)
        fld = np.zeros(ntot)
        for _ in range(24):
            kvec = rng.normal(size=3) * 2.5
            fld += np.cos(xyz @ kvec + rng.uniform(0, 2 * np.pi))
        thr = np.quantile(fld, 1.0 - land_fraction)
        ocean &= fld < thr
    elif mask != "aqua":
        raise ValueError(mask)

    sea_full = np.flatnonzero(ocean)
    nsea = sea_full.size
    full2sea = np.full(ntot + 1, nsea, dtype=np.int64)  # land -> nland ; slot ntot is a dummy "no such row"
    full2sea[sea_full] = np.arange(nsea)
    k = row_of[sea_full]
    i = i_of[sea_full]  # 0-based
    row_start = np.searchsorted(k, np.arange(ngy + 1))

    def pt(row, ii):
        """sea index of (row, lon index 0-based); rows outside the grid give land."""
        ok = (row >= 0) & (row < ngy)
        r = np.clip(row, 0, ngy - 1)
        ii = np.clip(ii, 0, nlon[r] - 1)
        return np.where(ok, full2sea[off[r] + ii], nsea)

    # ---- KLON (propconnect.F90:165-200), periodic
    nl = nlon[k]
    klon = np.stack([pt(k, (i - 1) % nl), pt(k, (i + 1) % nl)], 1)

    # ---- KLAT (propconnect.F90:69-160): closest and second closest in rows k-1, k+1; second one clamped (no wrap)
    klat = np.empty((nsea, 2, 2), dtype=np.int64)
    wlat = np.ones((nsea, 2))
    kcor = np.empty((nsea, 4, 2), dtype=np.int64)
    wcor = np.ones((nsea, 4))
    for ic, dk in ((0, -1), (1, +1)):
        kk = k + dk
        ok = (kk >= 0) & (kk < ngy)
        kr = np.clip(kk, 0, ngy - 1)
        n2 = nlon[kr]
        num = i * n2                       # XMIN = (I-1)*ZDELLO(K)/ZDELLO(K') = (I-1)*n'/n
        imin = _nint_div(num, nl)          # 0-based closest index
        left = num <= imin * nl            # XMIN <= IMIN-1 (1-based) <=> (I-1)n' <= (IMIN-1) n
        imin2 = np.where(left, np.maximum(imin - 1, 0), np.minimum(imin + 1, n2 - 1))
        klat[:, ic, 0] = np.where(ok, pt(kr, imin), nsea)
        klat[:, ic, 1] = np.where(ok, pt(kr, imin2), nsea)
        # WLAT (propconnect.F90:675-705, 780-805), in degrees
        zd, zd2 = zdello[k], zdello[kr]
        d0 = i * zd
        d3, d5 = d0 - 0.5 * zd, d0 + 0.5 * zd
        xp = imin * zd2
        d4, d6 = xp - 0.5 * zd2, xp + 0.5 * zd2
        w_left = np.where((d4 <= d3) | (d6 <= d5), 1.0, np.minimum(1.0, (zd - (d4 - d3)) / zd))
        w_right = np.where((d4 >= d3) | (d6 >= d5), 1.0, np.minimum(1.0, (zd - (d5 - d6)) / zd))
        wlat[:, ic] = np.where(ok, np.where(d0 <= xp, w_left, w_right), 1.0)
        # ---- KCOR / WCOR: west (-1) and east (+1) corners in this row (propconnect.F90:205-420, 706-760)
        for dx, icr in ((-1, 2 if dk < 0 else 3), (+1, 1 if dk < 0 else 0)):  # SW=3(idx2) SE=2(idx1) NW=4(idx3) NE=1(idx0)
            numc = (i + dx) * n2
            xmin_i = _nint_div(numc, nl)
            # periodic shift of the closest point
            wrap_lo = xmin_i < 0
            wrap_hi = xmin_i > n2 - 1
            iminc = np.where(wrap_lo, xmin_i + n2, np.where(wrap_hi, xmin_i - n2, xmin_i))
            numw = np.where(wrap_lo, numc + n2 * nl, np.where(wrap_hi, numc - n2 * nl, numc))
            leftc = numw <= iminc * nl
            imin2c = np.where(leftc, np.where(iminc <= 0, n2 - 1, iminc - 1), np.where(iminc >= n2 - 1, 0, iminc + 1))
            kcor[:, icr, 0] = np.where(ok, pt(kr, iminc), nsea)
            kcor[:, icr, 1] = np.where(ok, pt(kr, imin2c), nsea)
            xl = d0 + dx * zd
            xll, xlr = xl - 0.5 * zd, xl + 0.5 * zd
            xpc = _nint_div(numc, nl) * zd2          # the reference does not wrap XP here (propconnect.F90:712-716)
            xpl, xpr = xpc - 0.5 * zd2, xpc + 0.5 * zd2
            d1 = np.where((xpl > xll) & (xpr < xlr), zd, np.minimum(xlr, xpr) - np.maximum(xll, xpl))
            wcor[:, icr] = np.where(ok, np.minimum(1.0, d1 / zd), 1.0)

    return Grid(name=f"O{n}", ngy=ngy, nlonrgg=nlon, xdella=float(xdella), zdello=zdello, amosop=float(south), cosph=cosph,
                sinph=sinph, nsea=int(nsea), ixlg=i.astype(np.int32), kxlt=k.astype(np.int32), klon=klon.astype(np.int32),
                klat=klat.astype(np.int32), kcor=kcor.astype(np.int32), wlat=wlat, wcor=wcor, row_start=row_start)


def nsea_aqua(n: int) -> int:
    """Sea-point count of the all-ocean O<n> grid (4N(N+9)-40)."""
    return 4 * n * (n + 9) - 40


# ---- the tables on disk: an N-rank run builds the grid ONCE (rank 0) and hands it to the other ranks through a file instead of N ranks
#      building the whole grid side by side on a node's CPU share (54 s per rank at O1280; bench.py).  Plain .npy members in one .npz.
_ARRAYS = ("nlonrgg", "zdello", "cosph", "sinph", "ixlg", "kxlt", "klon", "klat", "kcor", "wlat", "wcor", "row_start")


def save_grid(g: Grid, path: str) -> None:
    """Write the tables to `path` (atomically: a reader never sees half a file)."""
    import os

    tmp = f"{path}.{os.getpid()}.tmp.npz"
    np.savez(tmp, name=np.array(g.name), scalars=np.array([g.ngy, g.nsea], dtype=np.int64), reals=np.array([g.xdella, g.amosop]),
             **{k: getattr(g, k) for k in _ARRAYS})
    os.replace(tmp, path)


def load_grid(path: str) -> Grid:
    with np.load(path) as z:
        ngy, nsea = (int(x) for x in z["scalars"])
        xdella, amosop = (float(x) for x in z["reals"])
        return Grid(name=str(z["name"]), ngy=ngy, nsea=nsea, xdella=xdella, amosop=amosop, **{k: z[k] for k in _ARRAYS})
