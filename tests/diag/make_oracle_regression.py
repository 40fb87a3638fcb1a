#!/usr/bin/env python3
"""Writes tests/golden/oracle_regression_dp.npz: outputs of THIS repository's CPU oracle (double precision) for small seeded
cases of IMPLSCH (flag set A, flag set B, sea-ice attenuation) and of CTUW + PROPAGS2 (plain and with currents).  These are
regression vectors -- they freeze the oracle's behaviour so that a later edit cannot change it unnoticed; they are NOT outputs
of the reference (which cannot be built here) and pin nothing against it."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def cases():
    out = {}
    for name, flags in (("A", {}), ("B", dict(llgcbz0=True, llnormagam=True)), ("ICE", dict(lciwa1=True, lciwa3=True, lciscal=True)),
                        ("JAN", dict(iphys=0)), ("SNL2", dict(isnonlin=2))):
        cfg = Config(nang=12, nfre=36, nfre_red=25, **flags)
        case = H.make_point_case(8, cfg, "dp", spectra="mixed", seed=101)
        case["FF"][:, 2] = np.linspace(0.0, 0.9, 8)
        case["FF"][:, 13] = np.linspace(0.0, 2.1, 8)
        r = H.oracle_implsch(case, Oracle(cfg, "dp"))
        out[f"implsch_{name}_FL1"] = r["FL1"]
        out[f"implsch_{name}_MIJ"] = r["MIJ"]
        out[f"implsch_{name}_FF"] = r["FF"]
        out[f"implsch_{name}_INTF"] = r["INTF"]
    import test_gpu_refraction as R
    for ir in (0, 2):
        c = R._case("dp", ir, n_oct=8, nang=12, nred=25)
        o = Oracle(c["cfg"], "dp")
        if ir:
            dot = o.propdot(c["g"], ir, c["dep"], c["u"], c["v"], c["wn"], c["cg"], c["om"])
            w = o.ctu_weights_gen(c["g"], ir, c["cg"], c["om"], c["u"], c["v"], dot, float(c["cfg"].idelpro))
            f3 = o.propags2_gen(c["g"], c["f1"], w)
        else:
            w = o.ctu_weights(c["g"], c["cg"], float(c["cfg"].idelpro))
            f3 = o.propags2(c["g"], c["f1"], w)
        out[f"propags2_irefra{ir}_F3"] = f3[: c["n"], :, :25]
    return out


if __name__ == "__main__":
    dst = os.path.join(ROOT, "tests", "golden", "oracle_regression_dp.npz")
    np.savez_compressed(dst, **cases())
    print("wrote", dst, os.path.getsize(dst), "bytes")
