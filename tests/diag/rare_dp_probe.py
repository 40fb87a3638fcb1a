"""Diagnostics (not a test): the double precision RARE builds of k_implsch4 (profiles/r04_rare_dp_note.txt) against k_implsch2 on the
configuration of test_implsch_fluxes_without_the_nonlinear_transfer[dp] -- every rare switch but LWVFLX_SNL off.  Prints WHERE the two
generations differ: by point position inside the wavefront, by output, by frequency row and direction.

ECWAM_HIP_LIB=<variant library> python tests/diag/rare_dp_probe.py [nang=24] [n=512] [prec=dp]
One process per variant: a faulting kernel ends the process (HSA queue abort)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H  # noqa: E402
from ecwam_amd import api  # noqa: E402
from ecwam_amd.tables import Config  # noqa: E402

nang = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
prec = sys.argv[3] if len(sys.argv) > 3 else "dp"
nred = {36: 36, 24: 29, 12: 25}[nang]
pp = {("dp", 36): 3, ("dp", 24): 4, ("dp", 12): 5, ("sp", 36): 3, ("sp", 24): 5, ("sp", 12): 10}[(prec, nang)]
cfg = Config(nang=nang, nfre=36, nfre_red=nred, lwvflx_snl=False)
case = H.make_point_case(n, cfg, prec, spectra="mixed", seed=71)
ctx = api.HipContext(case["tables"])
ctx.set_implsch_generation(2)
old = H.gpu_implsch(case, ctx)
print("lib", os.environ.get("ECWAM_HIP_LIB", "(product)"), "generation 2 ran:", ctx.implsch_generation_used(), flush=True)
ctx.set_implsch_generation(0)
new = H.gpu_implsch(case, ctx)
gen = ctx.implsch_generation_used()
print("automatic choice ran generation", gen, flush=True)
st = H.compare_implsch(old, new, case["tables"])
print({k: st[k] for k in ("mij_flips", "xllws_bins_diff", "fl1_max_rel_peak_all", "ff_max_rel_all", "intf_max_rel_all", "swh_max_rel")})
if gen != 4:
    print("RESULT: the variant does not carry the double precision RARE builds")
    sys.exit(0)
peak = np.abs(old["FL1"]).max(axis=(1, 2), keepdims=True)
e = np.abs(new["FL1"] - old["FL1"]) / np.maximum(peak, 1e-300)
bad = ~np.isfinite(new["FL1"]) | (e > 1e-9)
badpt = bad.any(axis=(1, 2))
print("points with a wrong spectrum:", int(badpt.sum()), "of", n)
if badpt.any():
    pos = np.arange(n) % pp
    print("  by position in the wavefront (point index mod PP=%d):" % pp, [int(badpt[pos == q].sum()) for q in range(pp)])
    print("  by wavefront: first bad waves", sorted(set((np.nonzero(badpt)[0] // pp).tolist()))[:20])
    print("  wrong bins by frequency row M:", bad.sum(axis=(0, 1)).tolist())
    print("  wrong bins by direction K   :", bad.sum(axis=(0, 2)).tolist())
    i = int(np.nonzero(badpt)[0][0])
    print("  first bad point", i, "MIJ old/new", int(old["MIJ"][i]), int(new["MIJ"][i]))
    print("   old FF", np.array2string(old["FF"][i], precision=5))
    print("   new FF", np.array2string(new["FF"][i], precision=5))
for name in ("FF", "INTF"):
    a, b = old[name].astype(float), new[name].astype(float)
    sc = np.maximum(np.abs(a).max(axis=0), 1e-300)
    d = np.where(np.isfinite(b), np.abs(b - a) / sc, np.inf)
    cols = {int(c): int((d[:, c] > 1e-9).sum()) for c in range(a.shape[1]) if (d[:, c] > 1e-9).any()}
    print(name, "columns with differences (column: points):", cols)
print("MIJ differences:", int((old["MIJ"] != new["MIJ"]).sum()), " XLLWS point differences:", st["xllws_pts_diff"])
print("RESULT:", "PASS" if (not badpt.any() and st["mij_flips"] == 0 and st["ff_max_rel_all"] < 1e-9) else "WRONG NUMBERS")
ctx.close()
