"""ctypes binding of libecwam_hip.so (include/ecwam_hip.h).  There is no CPU fallback: if the HIP
library is missing this module raises, and nothing in the product path imports the oracle."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .tables import JTOT_TAUHF, Tables

HERE = os.path.dirname(os.path.abspath(__file__))
# ECWAM_HIP_LIB: A/B timing of two builds of the same extension (tools/); never a fallback: the file must exist
LIBPATH = os.environ.get("ECWAM_HIP_LIB") or os.path.join(HERE, "lib", "libecwam_hip.so")

_INT_FIELDS_1 = ["nang", "nfre", "nfre_red", "nfre_odd", "idelt"]
_PARAM_LAYOUT = (
    [(n, C.c_int) for n in _INT_FIELDS_1]
    + [("ximp", C.c_double)]
    + [(n, C.c_int) for n in ["iphys", "isnonlin", "irefra", "icode", "llgcbz0", "llnormagam", "llcapchnk", "lbiwbk", "licerun",
                               "lmaskice", "lwamrsetci", "lciwa1", "lciwa2", "lciwa3", "lciscal", "lwvflx_snl", "lwflux",
                               "lwfluxout", "lwnemocou", "lwcou", "lwcouast", "lwnemocouwrs", "lwnemotauoc", "lwnemocousend",
                               "lwnemocoustk", "lwnemocoustrn"]]
    + [(n, C.c_double) for n in ["g", "gm1", "pi", "zpi", "zpi4gm1", "zpi4gm2", "epsmin", "rowater", "rowaterm1", "epsus",
                                  "epsu10", "acd", "bcd", "acdlin", "bcdlin", "cdmax", "tauocmin", "tauocmax", "phiepsmin",
                                  "phiepsmax", "wsemean_min", "circ", "r_earth", "fratio", "wetail", "frtail", "wp1tail", "fric",
                                  "delth", "flogsprdm1", "xkappa", "xnlev", "rnu", "rnum", "betamaxoxkappa2", "bmaxokap",
                                  "gamnconst", "zalp", "alpha", "alphamin", "alphamax", "chnkmin_u", "alphapmax", "tauwshelter", "dthrn_a",
                                  "dthrn_u", "tailfactor", "tailfactor_pm", "ang_gc_a", "ang_gc_b", "ang_gc_c", "rn1_rn", "swellf",
                                  "swellf2", "swellf3", "swellf4", "swellf5", "swellf6", "swellf7", "swellf7m1", "z0rat",
                                  "z0tubmax", "abmin", "abmax", "sdsbr", "ssdsc2", "ssdsc3", "ssdsc4", "ssdsc5", "ssdsc6", "miche"]]
    + [("nsdsnth", C.c_int), ("ipsat", C.c_int)]
    + [(n, C.c_double) for n in ["egrcrv", "afcrv", "bfcrv", "x0tauhf", "eps1", "flmin", "cithrsh", "ciblock", "cithrsh_tail",
                                  "zalpwrs", "bathymax", "wspmin", "wspmin_reset_tauw", "cdis", "delta_sdis", "cdisvis"]]
    + [("idamping", C.c_int)]
    + [(n, C.c_double) for n in ["cdicwa", "zalpfacb", "zalpfacx"]]
    + [("lwnemocouibr", C.c_int), ("zibrw_thrsh", C.c_double), ("nict", C.c_int), ("nich", C.c_int)]
    + [(n, C.c_double) for n in ["ticmin", "dtic", "dhic", "hicmin"]]
    + [("mfrstlw", C.c_int), ("mlsthg", C.c_int), ("kfrh", C.c_int), ("dal1", C.c_double), ("dal2", C.c_double),
       ("nwav_gc", C.c_int), ("xlogkratiom1_gc", C.c_double), ("sqrtgosurft", C.c_double)]
)


class Params(C.Structure):
    _fields_ = _PARAM_LAYOUT


_TABLE_NAMES = ["fr", "dfim", "dfimofr", "dfimfr", "dfim_sim", "rhowg_dfim", "zpifr", "fr5", "cofrm4", "flmax", "th", "costh",
                "sinth", "wtauhf", "swellft", "ikp", "ikp1", "ikm", "ikm1", "af11", "k1w", "k2w", "k11w", "k21w", "inlcoef",
                "rnlcoef", "indicessat", "satweights", "kpm", "jxo", "jyo", "kcr", "xk_gc", "xkm_gc", "omega_gc", "omxkm3_gc",
                "cm_gc", "c2osqrtvg_gc", "xkmsqrtvgoc2_gc", "om3gmkm_gc", "delkcc_gc_ns", "delkcc_omxkm3_gc", "cideac"]
_INT_TABLES = {"ikp", "ikp1", "ikm", "ikm1", "k1w", "k2w", "k11w", "k21w", "inlcoef", "indicessat", "kpm", "jxo", "jyo", "kcr"}


class TablePtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _TABLE_NAMES]


EXPORTS = ["ecwam_hip_last_error", "ecwam_hip_abi_version", "ecwam_hip_selftest", "ecwam_hip_create", "ecwam_hip_destroy", "ecwam_hip_set_obstructions", "ecwam_hip_propags2",
           "ecwam_hip_ctuw", "ecwam_hip_propags2_otf", "ecwam_hip_propags2_otf_split", "ecwam_hip_propags2_otf_fast", "ecwam_hip_set_fastwave_copy", "ecwam_hip_copy_freq_range", "ecwam_hip_propdot", "ecwam_hip_ctuw_refra", "ecwam_hip_propags2_refra", "ecwam_hip_implsch", "ecwam_hip_implsch_generation_used", "ecwam_hip_device_tables", "ecwam_hip_implsch_reserve", "ecwam_hip_propags2_implsch_supported", "ecwam_hip_propags2_implsch", "ecwam_hip_outbs", "ecwam_hip_outwnorm", "ecwam_hip_newwind", "ecwam_hip_newwind_icode", "ecwam_hip_nosource", "ecwam_hip_host_register", "ecwam_hip_host_unregister", "ecwam_hip_chunks_to_points",
           "ecwam_hip_points_to_chunks", "ecwam_hip_member_scatter", "ecwam_hip_member_gather", "ecwam_hip_pack_rows", "ecwam_hip_unpack_rows", "ecwam_hip_halo_setup", "ecwam_hip_halo_counts", "ecwam_hip_comm_unique_id",
           "ecwam_hip_comm_init", "ecwam_hip_comm_count", "ecwam_hip_halo_start", "ecwam_hip_halo_finish", "ecwam_hip_halo_pack_host", "ecwam_hip_halo_unpack_host", "ecwam_hip_proenvhalo_pack", "ecwam_hip_proenvhalo_unpack", "ecwam_hip_malloc", "ecwam_hip_free",
           "ecwam_hip_memcpy_h2d", "ecwam_hip_memcpy_d2h", "ecwam_hip_memset", "ecwam_hip_sync", "ecwam_hip_queue_create", "ecwam_hip_queue_destroy",
           "ecwam_hip_queue_wait_for"]

ABI_VERSION = 6        # include/ecwam_hip.h ECWAM_HIP_ABI_VERSION
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise RuntimeError(f"{LIBPATH} not found: the HIP extension is not built (run `python -m ecwam_amd.build` or "
                           "__graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm bundles its own HIP runtime (same SONAME as the system one the extension was linked against).  Whichever is
    # mapped first serves the whole process, and two different copies cannot both own the devices: load torch's first so
    # that device buffers, streams and our kernels live in ONE runtime, whatever order the caller imports things in.
    import torch  # noqa: F401
    lib = C.CDLL(LIBPATH)
    lib.ecwam_hip_last_error.restype = C.c_char_p
    lib.ecwam_hip_abi_version.restype = C.c_int
    if lib.ecwam_hip_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIBPATH}: ABI version {lib.ecwam_hip_abi_version()} but this binding is written for {ABI_VERSION}; rebuild the extension")
    for name in EXPORTS[1:]:
        getattr(lib, name).restype = C.c_int
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    lib.ecwam_hip_create.argtypes = [C.POINTER(Params), C.POINTER(TablePtrs), ci, ci, C.POINTER(vp)]
    lib.ecwam_hip_destroy.argtypes = [vp]
    lib.ecwam_hip_selftest.argtypes = [ci]
    lib.ecwam_hip_set_obstructions.argtypes = [vp, vp, ci]
    lib.ecwam_hip_propags2.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_ctuw.argtypes = [vp, ci, ci, ci, cd, ci, ci, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ecwam_hip_propags2_otf.argtypes = [vp, vp, vp, ci, ci, cd, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_propags2_otf_split.argtypes = [vp, vp, vp, ci, ci, cd, cd, ci, ci, vp, ci, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_propags2_otf_fast.argtypes = [vp, vp, vp, ci, ci, cd, cd, ci, ci, vp, ci, ci, vp, ci, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_set_fastwave_copy.argtypes = [vp, vp, ci]
    lib.ecwam_hip_propags2_implsch_supported.argtypes = [vp]
    lib.ecwam_hip_propags2_implsch.argtypes = [vp, vp, vp, ci, ci, cd, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, vp,
                                               cd, ci, vp, ci, ci, vp]
    lib.ecwam_hip_copy_freq_range.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
    lib.ecwam_hip_propdot.argtypes = [vp, ci, ci, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ecwam_hip_ctuw_refra.argtypes = [vp, ci, ci, ci, cd, ci, ci, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp]
    lib.ecwam_hip_propags2_refra.argtypes = [vp, vp, vp, ci, ci, cd, vp, vp, cd, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_implsch.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.ecwam_hip_device_tables.argtypes = [vp]
    lib.ecwam_hip_device_tables.restype = vp
    lib.ecwam_hip_implsch_reserve.argtypes = [vp, ci]
    lib.ecwam_hip_implsch_generation_used.argtypes = [vp]
    lib.ecwam_hip_outbs.argtypes = [vp, ci, ci, vp, cd, vp, vp]
    lib.ecwam_hip_outwnorm.argtypes = [vp, vp, ci, ci, cd, C.POINTER(C.c_double), vp]
    lib.ecwam_hip_newwind.argtypes = [vp, ci, vp, vp, vp]
    lib.ecwam_hip_newwind_icode.argtypes = [vp, ci, vp, vp, ci, vp]
    lib.ecwam_hip_nosource.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    lib.ecwam_hip_host_register.argtypes = [vp, vp, C.c_ulonglong]
    lib.ecwam_hip_host_unregister.argtypes = [vp, vp]
    lib.ecwam_hip_chunks_to_points.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_points_to_chunks.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ecwam_hip_member_scatter.argtypes = [vp, vp, vp, ci, ci, ci, ci, C.c_longlong, C.c_longlong, ci, vp]
    lib.ecwam_hip_member_gather.argtypes = [vp, vp, vp, ci, ci, ci, ci, C.c_longlong, C.c_longlong, ci, vp]
    lib.ecwam_hip_pack_rows.argtypes = [vp, vp, vp, ci, vp, vp]
    lib.ecwam_hip_unpack_rows.argtypes = [vp, vp, ci, vp, ci, vp]
    lib.ecwam_hip_halo_setup.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp]
    lib.ecwam_hip_halo_counts.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
    lib.ecwam_hip_comm_unique_id.argtypes = [vp]
    lib.ecwam_hip_comm_init.argtypes = [vp, vp]
    lib.ecwam_hip_comm_count.argtypes = [vp, C.POINTER(ci)]
    lib.ecwam_hip_halo_start.argtypes = [vp, vp, ci, vp]
    lib.ecwam_hip_halo_finish.argtypes = [vp, vp]
    lib.ecwam_hip_halo_pack_host.argtypes = [vp, vp, ci, vp, vp]
    lib.ecwam_hip_halo_unpack_host.argtypes = [vp, vp, ci, vp, vp]
    lib.ecwam_hip_proenvhalo_pack.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.ecwam_hip_proenvhalo_unpack.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib


def make_params(t: Tables) -> Params:
    c = t.cfg
    p = Params()
    for name, typ in _PARAM_LAYOUT:
        if hasattr(c, name) and name not in ("wspmin", "rnu", "rnum"):
            v = getattr(c, name)
        elif name == "r_earth":
            v = t.R
        else:
            v = getattr(t, name.upper())
        setattr(p, name, int(v) if typ is C.c_int else float(v))
    return p


def make_tables(t: Tables):
    """Returns (TablePtrs, keepalive list).  Index tables are passed 1-based as the reference holds them."""
    T = t.dtype
    keep = []
    tp = TablePtrs()
    one_based = {"indicessat": 1, "kpm": 1}  # ours are stored 0-based in Tables
    for name in _TABLE_NAMES:
        a = getattr(t, name.upper())
        if name in _INT_TABLES:
            a = np.ascontiguousarray(np.asarray(a) + one_based.get(name, 0), dtype=np.int32)
        else:
            a = np.ascontiguousarray(a, dtype=T)
        keep.append(a)
        setattr(tp, name, a.ctypes.data_as(C.c_void_p))
    assert t.WTAUHF.size == JTOT_TAUHF
    return tp, keep
