"""Sea-point block decomposition for the multi-GPU path (one process per GPU).

Mirrors the reference's 1-D latitude-band decomposition (mpdecomp.F90:58-100, LL1D): the
south->north sea-point list is cut into contiguous ranges of (almost) equal point counts; each
rank renumbers its neighbour tables into local indices [0,n) owned, [n,n+nh) halo (sorted by global
index, hence grouped by owner), nland = n+nh for land ("NSUP+1").  The advection halo exchange
(mpexchng.F90:141-206) then is: pack the owned points each neighbouring rank needs, exchange
point-to-point, receive straight into the contiguous halo segment of that rank.
"""
from __future__ import annotations

import dataclasses

import numpy as np


def split_points(nsea: int, nranks: int) -> np.ndarray:
    """bounds[r]..bounds[r+1] = range of rank r (mpdecomp.F90: equal counts, remainder to the first ranks)."""
    base, rem = divmod(nsea, nranks)
    counts = np.full(nranks, base, dtype=np.int64)
    counts[:rem] += 1
    return np.concatenate([[0], np.cumsum(counts)])


def _halo_global(grid, lo: int, hi: int) -> np.ndarray:
    nb = np.concatenate([grid.klon[lo:hi].ravel(), grid.klat[lo:hi].ravel(), grid.kcor[lo:hi].ravel()]).astype(np.int64)
    nb = nb[(nb != grid.nland) & ((nb < lo) | (nb >= hi))]
    return np.unique(nb)


@dataclasses.dataclass
class LocalDomain:
    rank: int
    nranks: int
    lo: int
    hi: int
    n: int                      # owned points
    nh: int                     # halo points
    halo_global: np.ndarray     # [nh] global indices, sorted
    klon: np.ndarray            # local numbering
    klat: np.ndarray
    kcor: np.ndarray
    kxlt: np.ndarray            # [n]
    cosphm1_ext: np.ndarray     # [n+nh+1]
    send: dict                  # peer -> local owned indices to send (int32, ordered by global index)
    recv: dict                  # peer -> (dst0 local row, count)

    @property
    def nland(self) -> int:
        return self.n + self.nh

    @property
    def nrows(self) -> int:
        return self.n + self.nh + 1

    def ext_global(self) -> np.ndarray:
        """global index of every local row except land"""
        return np.concatenate([np.arange(self.lo, self.hi), self.halo_global])

    def interior(self) -> tuple:
        """[a, b): the longest run of owned rows whose advection stencil reads no halo row.  The owned range is a
        latitude band, so the rows that do read one sit at its two ends: [0, a) and [b, n).  Lets the halo exchange run
        while the interior is advected."""
        n, nl = self.n, self.nland
        nb = np.concatenate([self.klon.reshape(n, -1), self.klat.reshape(n, -1), self.kcor.reshape(n, -1)], axis=1)
        needs = np.flatnonzero(((nb >= n) & (nb < nl)).any(axis=1))
        if needs.size == 0:
            return 0, n
        edges = np.concatenate([[-1], needs, [n]])
        g = int(np.argmax(np.diff(edges)))
        return int(edges[g] + 1), int(edges[g + 1])


def local_domain(grid, rank: int, nranks: int) -> LocalDomain:
    bounds = split_points(grid.nsea, nranks)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    n = hi - lo
    halo = _halo_global(grid, lo, hi)
    nh = halo.size
    nland = n + nh

    def renum(a):
        a = a[lo:hi].astype(np.int64)
        out = np.full(a.shape, nland, dtype=np.int64)
        own = (a >= lo) & (a < hi)
        out[own] = a[own] - lo
        ish = ~own & (a != grid.nland)
        out[ish] = n + np.searchsorted(halo, a[ish])
        return out.astype(np.int32)

    cos_ext = np.zeros(nland + 1)
    full = grid.cosphm1_ext
    cos_ext[:n] = full[lo:hi]
    cos_ext[n:nland] = full[halo]
    owner = np.searchsorted(bounds, halo, side="right") - 1
    recv, send = {}, {}
    for p in np.unique(owner):
        idx = np.flatnonzero(owner == p)
        assert np.all(np.diff(idx) == 1)
        recv[int(p)] = (n + int(idx[0]), int(idx.size))
    # only a band within one latitude row of this one can read its rows (KLAT / KCOR reach the adjacent latitude rows, KLON stays in the
    # row): the others are skipped -- at 8 ranks on O1280 that is two bands' neighbour tables scanned per rank instead of seven
    kx = np.asarray(grid.kxlt)
    my_rows = (int(kx[lo]), int(kx[hi - 1])) if n else (0, -1)
    for p in range(nranks):
        if p == rank or not n or bounds[p + 1] <= bounds[p]:
            continue
        if int(kx[bounds[p]]) > my_rows[1] + 1 or int(kx[bounds[p + 1] - 1]) < my_rows[0] - 1:
            continue
        ph = _halo_global(grid, int(bounds[p]), int(bounds[p + 1]))
        mine = ph[(ph >= lo) & (ph < hi)]
        if mine.size:
            send[p] = (mine - lo).astype(np.int32)
    return LocalDomain(rank=rank, nranks=nranks, lo=lo, hi=hi, n=n, nh=nh, halo_global=halo, klon=renum(grid.klon),
                       klat=renum(grid.klat), kcor=renum(grid.kcor), kxlt=grid.kxlt[lo:hi].astype(np.int32),
                       cosphm1_ext=cos_ext, send=send, recv=recv)


def strip_order(grid, dom, width: int = 256) -> np.ndarray:
    """Processing order of the owned points for the advection kernel: longitude strips of about `width` points on the
    longest owned latitude row, each strip walked row by row.  A point's latitude neighbours (KLAT/KCOR, the nearest
    longitudes of the adjacent rows) then lie ~`width` points away in the processing sequence instead of a whole row
    (~4*N points on an O-N grid), so that they are still in the XCD's L2 when they are needed.  Pure work ordering:
    the storage order of the spectra is the reference's (block order of mpdecomp.F90)."""
    ix = np.asarray(grid.ixlg[dom.lo:dom.hi], dtype=np.int64)        # 0-based longitude index in the row
    ky = np.asarray(grid.kxlt[dom.lo:dom.hi], dtype=np.int64)        # latitude row
    nlon = np.asarray(grid.nlonrgg, dtype=np.int64)[ky]
    lon = (ix + 0.5) / nlon                                          # longitude as a fraction of the circle
    rows = np.unique(ky)
    nmax = int(max(np.asarray(grid.nlonrgg)[rows].max(), 1))
    nstrip = max(1, int(round(nmax / float(width))))
    strip = np.minimum((lon * nstrip).astype(np.int64), nstrip - 1)
    key = (strip * (int(ky.max()) + 1) + ky) * (nmax + 1) + ix
    return np.argsort(key, kind="stable").astype(np.int32)


def tile2d_order(grid, dom, rows: int = 4, tile: int = 16) -> np.ndarray:
    """Processing order of the owned points as 2-D tiles for k_propags2_otf: groups of `rows` consecutive latitude rows are cut
    into longitude segments of at most tile/rows points per row; a tile lists its points as entry t = rows*g + w = the g-th point
    of the segment of row w, padded with -1 (skipped by the kernel) where a row of the group is shorter.  The four wavefronts of
    a workgroup then work on latitude neighbours at the same time (see the kernel).  Pure work ordering, like strip_order."""
    per = tile // rows
    ix = np.asarray(grid.ixlg[dom.lo:dom.hi], dtype=np.int64)
    ky = np.asarray(grid.kxlt[dom.lo:dom.hi], dtype=np.int64)
    out = []
    urows = np.unique(ky)
    for r0 in range(0, len(urows), rows):
        grp = urows[r0:r0 + rows]
        idx = []
        for r in grp:
            sel = np.nonzero(ky == r)[0]
            idx.append(sel[np.argsort(ix[sel], kind="stable")])
        nmax = max(len(a) for a in idx)
        nbins = -(-nmax // per)
        t = np.full((nbins, per, rows), -1, dtype=np.int64)
        for w, a in enumerate(idx):
            n = len(a)
            edges = (np.arange(nbins + 1) * n) // nbins         # <= per points per bin since n <= nmax <= per*nbins
            b = np.searchsorted(edges, np.arange(n), side="right") - 1
            g = np.arange(n) - edges[b]
            t[b, g, w] = a
        out.append(t.reshape(-1))
    return np.concatenate(out).astype(np.int32)
