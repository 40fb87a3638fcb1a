"""Synthetic inputs for the hot path: cold-start spectra, forcing, depth.

The reference's inputs come from PRESET (JONSWAP x cos^2 cold start: jonswap.F90:70-92, spr.F90,
namelist ALFA=0.018 GAMMA=3.0 SA=0.07 SB=0.09, ecwam_run_preset.sh:200-206) and from GRIB forcing
files that cannot be obtained offline; this module produces inputs of the same shape and value
range from a seed.  Per-point parameters are drawn for the *global* point index so that every
shard of a decomposed run sees identical values.

Layouts (C order):  FL1[ij][k][m];  per-frequency [ij][m];  FF[ij][14];  INTF[ij][15];  ENV[ij][2].
FF members  : AIRD WDWAVE CICOVER WSWAVE WSTAR USTRA VSTRA UFRIC TAUW TAUWDIR Z0M Z0B CHRNCK CITHICK
              (FORCING_FIELDS members IMPLSCH reads/writes, implsch.F90:13-14)
INTF members: WSEMEAN WSFMEAN USTOKES VSTOKES STRNMS TAUXD TAUYD TAUOCXD TAUOCYD TAUOC TAUICX TAUICY
              PHIOCD PHIEPS PHIAW   (INTGT_PARAM_FIELDS members of implsch.F90:19-22)
"""
from __future__ import annotations

import numpy as np

FF_NAMES = ["AIRD", "WDWAVE", "CICOVER", "WSWAVE", "WSTAR", "USTRA", "VSTRA", "UFRIC", "TAUW", "TAUWDIR", "Z0M", "Z0B",
            "CHRNCK", "CITHICK"]
INTF_NAMES = ["WSEMEAN", "WSFMEAN", "USTOKES", "VSTOKES", "STRNMS", "TAUXD", "TAUYD", "TAUOCXD", "TAUOCYD", "TAUOC",
              "TAUICX", "TAUICY", "PHIOCD", "PHIEPS", "PHIAW"]
NFF, NINTF = len(FF_NAMES), len(INTF_NAMES)


def point_params(nglobal: int, seed: int = 12345, shallow_fraction: float = 0.1, ice_fraction: float = 0.1) -> dict:
    """Per-point scalars for all global points (cheap: O(10) floats per point)."""
    rng = np.random.default_rng(seed)
    p = {}
    p["FP"] = rng.uniform(0.08, 0.35, nglobal)
    p["THETAQ"] = rng.uniform(0.0, 2 * np.pi, nglobal)
    p["WSWAVE"] = np.clip(8.0 * np.sqrt(-np.log1p(-rng.uniform(0, 1, nglobal))) * 1.0, 1.0, 40.0)  # Weibull k=2, scale 8
    p["WDWAVE"] = rng.uniform(0.0, 2 * np.pi, nglobal)
    p["WSTAR"] = rng.uniform(0.0, 1.0, nglobal)
    ice = rng.uniform(0, 1, nglobal) < ice_fraction
    p["CICOVER"] = np.where(ice, rng.uniform(0.0, 1.0, nglobal), 0.0)
    shallow = rng.uniform(0, 1, nglobal) < shallow_fraction
    p["DEPTH"] = np.where(shallow, 10.0 ** rng.uniform(0.3, 3.0, nglobal), 998.999)
    p["DEPTH"] = np.minimum(p["DEPTH"], 998.999)
    return p


def jonswap_spectra(fr, th, fp, thq, dtype, epsmin=0.1e-32, alfa=0.018, gamma=3.0, sa=0.07, sb=0.09):
    """FL1[ij][k][m] = JONSWAP(fr; fp) * (2/pi) max(0, cos(th-thq))^2, floored at EPSMIN."""
    fr = np.asarray(fr, dtype=np.float64)[None, :]
    fp = np.asarray(fp, dtype=np.float64)[:, None]
    g, zpi = 9.806, 2 * np.pi
    sig = np.where(fr <= fp, sa, sb)
    arg = -0.5 * ((fr - fp) / (sig * fp)) ** 2
    et = alfa * g * g / zpi ** 4 * fr ** -5.0 * np.exp(-1.25 * (fp / fr) ** 4) * gamma ** np.exp(arg)
    c = np.maximum(0.0, np.cos(np.asarray(th, dtype=np.float64)[None, :] - np.asarray(thq)[:, None]))
    st = (2.0 / np.pi) * c * c
    fl = st[:, :, None] * et[:, None, :]
    return np.maximum(fl, epsmin).astype(dtype)


def forcing(p: dict, sl: slice, tables, dtype) -> np.ndarray:
    """FF[ij][14] first guess as in SURVEY.md 8d (UFRIC=sqrt(ACD+BCD*U)*U, TAUW=0.1*UFRIC^2, Z0M=1e-4...)."""
    n = p["WSWAVE"][sl].size
    ff = np.zeros((n, NFF))
    u = p["WSWAVE"][sl]
    us = np.sqrt(float(tables.ACD) + float(tables.BCD) * u) * u
    ff[:, 0] = 1.225
    ff[:, 1] = p["WDWAVE"][sl]
    ff[:, 2] = p["CICOVER"][sl]
    ff[:, 3] = u
    ff[:, 4] = p["WSTAR"][sl]
    ff[:, 7] = us
    ff[:, 8] = 0.1 * us * us
    ff[:, 9] = p["WDWAVE"][sl]
    ff[:, 10] = 0.0001
    ff[:, 11] = 0.0001
    ff[:, 12] = 0.0185
    return ff.astype(dtype)


def depth_props(depth, tables, dtype) -> dict:
    """Per-point FREQUENCY fields + EMAXDPT (depthprpt.F90:60-82, aki.F90:71-91, initdpthflds.F90:64-75),
    evaluated in ``dtype`` with the reference's operation order."""
    T = np.dtype(dtype).type
    t = tables
    d = np.asarray(depth, dtype=T)
    n, nfre = d.size, t.cfg.nfre
    G = T(t.G)
    out = {k: np.zeros((n, nfre), dtype=T) for k in ("WAVNUM", "CINV", "CGROUP", "XK2CG", "OMOSNH2KD", "STOKFAC")}
    gh = G / (T(4.0) * T(t.PI))
    ebs, dkmax = T(0.0001), T(40.0)
    for m in range(nfre):
        om = T(t.ZPIFR[m])
        akm1 = om * om / (T(4.0) * G)
        akm2 = om / (T(2.0) * np.sqrt(G * d))
        ao = np.maximum(akm1, akm2).astype(T)
        ak = np.zeros(n, dtype=T)
        done = np.zeros(n, dtype=bool)
        for _ in range(200):
            akp = ao
            bo = d * ao
            deep = (bo > dkmax) & ~done
            ak[deep] = om * om / G
            done |= deep
            act = ~done
            if not act.any():
                break
            with np.errstate(over="ignore", invalid="ignore"):
                thv = G * ao * np.tanh(bo)
                sth = np.sqrt(thv)
                ch = np.cosh(np.minimum(bo, T(45.0)))
                new = ao + (om - sth) * sth * T(2.0) / (thv / ao + G * bo / (ch * ch))
            new = np.where(act, new, ao).astype(T)
            conv = act & ~(np.abs(akp - new) > ebs * new)
            ak[conv] = new[conv]
            done |= conv
            ao = new
        akd = ak * d
        shallow = akd <= T(10.0)
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            akds = np.where(shallow, akd, T(1.0))
            cg_s = T(0.5) * np.sqrt(G * np.tanh(akds) / ak) * (T(1.0) + T(2.0) * akds / np.sinh(T(2.0) * akds))
            om_s = om / np.sinh(T(2.0) * akds)
            st_s = T(2.0) * G * ak * ak / (om * np.tanh(T(2.0) * akds))
        out["WAVNUM"][:, m] = ak
        out["CGROUP"][:, m] = np.where(shallow, cg_s, gh / T(t.FR[m]))
        out["OMOSNH2KD"][:, m] = np.where(shallow, om_s, T(0.0))
        out["STOKFAC"][:, m] = np.where(shallow, st_s, T(2.0) / G * om * om * om)
        out["CINV"][:, m] = ak / om
        out["XK2CG"][:, m] = ak * ak * out["CGROUP"][:, m]
    gd = T(t.GAM_B_J) * d
    out["EMAXDPT"] = T(0.0625) * (gd * gd)
    return out


def currents(grid, amp: float = 0.8, undefined_band: float = 10.0):
    """Synthetic surface current (UCUR, VCUR) [m/s] on the sea points of `grid`: a smooth rotational pattern of
    amplitude ~`amp`, exactly zero ("current field not defined", gradi.F90:170-180) poleward of 90-`undefined_band`
    degrees and in one mid-latitude box.  Function of the geographic position only: identical for every decomposition."""
    lat = np.deg2rad(np.asarray(grid.lat_deg, dtype=np.float64))
    lon = np.deg2rad(np.asarray(grid.ixlg, dtype=np.float64) * np.asarray(grid.zdello)[grid.kxlt])
    u = amp * np.cos(lat) * (0.6 * np.sin(2 * lon + 0.3) * np.cos(3 * lat) + 0.4 * np.cos(5 * lon) * np.sin(4 * lat))
    v = amp * np.cos(lat) * (0.5 * np.cos(3 * lon - 0.7) * np.sin(2 * lat) + 0.3 * np.sin(7 * lon) * np.cos(5 * lat))
    undef = (np.abs(np.rad2deg(lat)) > 90.0 - undefined_band) | ((np.abs(np.rad2deg(lat) - 20.0) < 8.0) & (np.abs(np.rad2deg(lon) - 200.0) < 15.0))
    u[undef] = 0.0
    v[undef] = 0.0
    return u, v


def obstructions(grid, nfre: int, seed: int = 99, fraction: float = 0.25):
    """Synthetic sub-grid obstruction (transmission) coefficients OBS[n][8][NFRE] = OBSLAT(IJ,M,1:2), OBSLON(IJ,M,1:2),
    OBSCOR(IJ,M,1:4) (getbobstrct.F90 reads the real ones from the grid file): 1 (open) at most points, 0.3 .. 1 at a
    seeded fraction, weaker blocking for the longer waves (low frequencies diffract around small islands), 0 in a few
    directions.  Function of the global point index only."""
    rng = np.random.default_rng(seed)
    n = grid.nsea
    obs = np.ones((n, 8, nfre))
    hit = rng.uniform(size=(n, 8)) < fraction
    base = rng.uniform(0.3, 1.0, (n, 8))
    base[rng.uniform(size=(n, 8)) < 0.02] = 0.0
    ramp = np.linspace(0.0, 1.0, nfre)[None, None, :]            # short waves are blocked fully, long ones half as much
    val = 1.0 - (1.0 - base[:, :, None]) * (0.5 + 0.5 * ramp)
    obs[hit] = val[hit]
    return obs
