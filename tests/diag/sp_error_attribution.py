#!/usr/bin/env python3
"""Where the single-precision difference between k_implsch4 and the oracle comes from (diagnostic, not a test).

For every build variant of the IMPLSCH translation units (ecwam_amd/build.py VARIANTS: hardware reciprocal / square root, hardware
exp2 / log2 with a one-product argument scaling, FMA contraction) one fresh process loads that library (ECWAM_HIP_LIB) and reports
  * the parity statistics of one IMPLSCH call against the oracle (1 536 points, 36 x 36, mixed sea and swell, IDELT = 900 and 450 s),
  * the per-point swh difference after 12 full WAMINTGR steps on a small grid with land (24 x 29, IDELT = 900 s),
  * the time of an IMPLSCH launch on 131 072 points.
python tests/diag/sp_error_attribution.py [variant ...]        (builds what is missing; prints one line per variant and a table)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


ORACLE_ROW = "--oracle-row" in sys.argv


def worker() -> None:
    import numpy as np
    import torch

    import harness as H
    from ecwam_amd import api, grid as G
    from ecwam_amd.tables import Config
    from ecwam_amd.wamintgr import Wamintgr
    from oracle.oracle import Oracle

    out = {}
    prec = "sp"
    for idelt in (900, 450):
        cfg = Config(nang=36, nfre=36, nfre_red=36, idelt=idelt, idelpro=idelt)
        case = H.make_point_case(1536, cfg, prec, spectra="mixed", seed=777)
        ref = H.oracle_implsch(case, Oracle(cfg, prec))
        # the same single-precision inputs through the double-precision oracle: what single precision itself loses
        cd = dict(case, prec="dp", FL1=case["FL1"].astype(np.float64), FF=case["FF"].astype(np.float64), INTF=case["INTF"].astype(np.float64),
                  ENV=case["ENV"].astype(np.float64), props={k: v.astype(np.float64) for k, v in case["props"].items()})
        truth = H.oracle_implsch(cd, Oracle(cfg, "dp"))
        keys = ("mij_flips", "xllws_pts_diff", "fl1_max_rel_peak_clean", "swh_max_rel", "ff_max_rel_clean", "intf_max_rel_clean",
                "fl1_frac_sig_bins_gt_1e-5")
        if ORACLE_ROW:
            st = H.compare_implsch(truth, ref, case["tables"])
            out[f"idelt{idelt}"] = {k: st[k] for k in keys}
            out[f"idelt{idelt}_vs_dp"] = out[f"idelt{idelt}"]
            out["ms_131072"] = float("nan")
            continue
        ctx = api.HipContext(case["tables"])
        got = H.gpu_implsch(case, ctx)
        st = H.compare_implsch(ref, got, case["tables"])
        out[f"idelt{idelt}"] = {k: st[k] for k in keys}
        st2 = H.compare_implsch(truth, got, case["tables"])
        out[f"idelt{idelt}_vs_dp"] = {k: st2[k] for k in keys}
        if idelt == 900:      # launch time on 131 072 points
            n, nc = 131072, 1536
            dev = ctx.device
            wv, ff, intf = H.pack_device_inputs(case)
            rep = (n + nc - 1) // nc
            fl0 = torch.from_numpy(case["FL1"]).to(dev).repeat(rep, 1, 1)[:n].contiguous()
            twv = torch.from_numpy(wv).to(dev).repeat(rep, 1, 1)[:n].contiguous()
            tff0 = torch.from_numpy(ff).to(dev).repeat(rep, 1)[:n].contiguous()
            tin0 = torch.from_numpy(intf).to(dev).repeat(rep, 1)[:n].contiguous()
            ts = []
            for _ in range(6):
                fl, tff, tin = fl0.clone(), tff0.clone(), tin0.clone()
                mij = torch.zeros(n, dtype=torch.int32, device=dev)
                xl = torch.zeros_like(fl0)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.implsch(0, n, fl, twv, tff, tin, mij, xl)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out["ms_131072"] = min(ts[1:])
        ctx.close()
    if ORACLE_ROW:
        out["steps12"] = {"swh_pt_max": float("nan"), "swh_pt_p99": float("nan"), "swh_avg": float("nan"), "swh_max": float("nan")}
        print("RESULT " + json.dumps(out), flush=True)
        return
    # twelve full steps, per-point swh and the global norms
    cfg = Config(nang=24, nfre=36, nfre_red=29)
    g = G.build_grid(16, mask="continents")
    m = Wamintgr(cfg, g, prec)
    m.init_synthetic(seed=3)
    n = g.nsea
    o = Oracle(cfg, prec)
    fl = m.fl1.cpu().numpy().copy()
    wv = m.wvprpt.cpu().numpy()
    ff = m.ff.cpu().numpy()[:, :14].copy()
    env = m.ff.cpu().numpy()[:, 14:16].copy()
    intf = np.zeros((n, 15), np.float32)
    wref = o.ctu_weights(g, m.cgroup_ext.cpu().numpy(), float(cfg.idelpro))
    for _ in range(12):
        m.step()
        f3 = o.propags2(g, fl, wref)
        f3[:, :, cfg.nfre_red:] = fl[:, :, cfg.nfre_red:]
        r = o.implsch(f3[:n], wv[:, 0], wv[:, 1], wv[:, 2], wv[:, 3], wv[:, 4], env, ff, intf)
        fl[:n], ff, intf = r["FL1"], r["FF"], r["INTF"]
    hs_g = m.outbs().cpu().numpy()[:, 0].astype(float)
    hs_w = o.outbs(fl[:n])[:, 0].astype(float)
    rel = np.abs(hs_g - hs_w) / np.maximum(hs_w, 0.05)
    out["steps12"] = {"swh_pt_max": float(rel.max()), "swh_pt_p99": float(np.percentile(rel, 99)),
                      "swh_avg": float(abs(hs_g.mean() - hs_w.mean()) / hs_w.mean()), "swh_max": float(abs(hs_g.max() - hs_w.max()) / hs_w.max())}
    m.ctx.close()
    print("RESULT " + json.dumps(out), flush=True)


def main() -> None:
    from ecwam_amd import build as B

    variants = [a for a in sys.argv[1:] if not a.startswith("--")] or list(B.VARIANTS)
    rows = []
    for v in ["oracle-sp"] + variants:
        v = "" if v in ("default", "product") else v
        env = dict(os.environ)
        if v != "oracle-sp":
            lib = B.lib_path(v)
            if not os.path.exists(lib):
                B.build(variant=v)
            env["ECWAM_HIP_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"] + (["--oracle-row"] if v == "oracle-sp" else []), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if r.returncode != 0 or not line:
            print(f"variant {v or 'product'}: FAILED\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}", flush=True)
            continue
        d = json.loads(line[-1][7:])
        rows.append((v or "product", "the sp oracle against the dp oracle on the same sp inputs" if v == "oracle-sp" else " ".join(B.VARIANTS[v]), d))
        print(f"{v or 'product'}: {json.dumps(d)}", flush=True)
    print()
    print("variant  | flags | ms/131072 pts | IDELT=900 vs sp oracle: bins/peak  swh  forcing  fluxes  MIJ flips | IDELT=900 vs DP oracle: bins/peak  swh  forcing | "
          "IDELT=450 vs sp oracle: bins/peak  swh | 12 steps vs sp oracle: swh point max / p99  norm avg  norm max")
    for name, flags, d in rows:
        a, b, s, t = d["idelt900"], d["idelt450"], d["steps12"], d["idelt900_vs_dp"]
        print(f"{name:9s} | {flags} | {d['ms_131072']:.3f} | {a['fl1_max_rel_peak_clean']:.2e} {a['swh_max_rel']:.2e} {a['ff_max_rel_clean']:.2e} "
              f"{a['intf_max_rel_clean']:.2e} {a['mij_flips']} | {t['fl1_max_rel_peak_clean']:.2e} {t['swh_max_rel']:.2e} {t['ff_max_rel_clean']:.2e} | "
              f"{b['fl1_max_rel_peak_clean']:.2e} {b['swh_max_rel']:.2e} | "
              f"{s['swh_pt_max']:.2e} {s['swh_pt_p99']:.2e} {s['swh_avg']:.2e} {s['swh_max']:.2e}")


if __name__ == "__main__":
    if "--worker" in sys.argv:
        worker()
    else:
        main()
