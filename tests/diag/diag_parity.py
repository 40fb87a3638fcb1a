"""Diagnostic (not a test): prints parity statistics of the HIP kernels against the oracle."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import harness as H  # noqa: E402
from ecwam_amd import api, grid as G  # noqa: E402
from ecwam_amd.tables import Config, Tables  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402


def diag_implsch(n=2048):
    for nang, nred in ((36, 36), (24, 29), (12, 25)):
        for prec in ("dp", "sp"):
            cfg = Config(nang=nang, nfre=36, nfre_red=nred)
            case = H.make_point_case(n, cfg, prec, spectra="mixed")
            o = Oracle(cfg, prec)
            t0 = time.time()
            ref = H.oracle_implsch(case, o, want_dbg=True)
            t1 = time.time()
            ctx = api.HipContext(case["tables"])
            got = H.gpu_implsch(case, ctx, want_dbg=True)
            st = H.compare_implsch(ref, got, case["tables"])
            d = np.abs(got["DBG"][:, :7].astype(float) - ref["DBG"][:, :7].astype(float)) / np.maximum(np.abs(ref["DBG"][:, :7]), 1e-30)
            st["dbg_max_rel"] = [float(x) for x in d.max(0)]
            st["cpu_s"] = t1 - t0
            print(f"IMPLSCH nang={nang} prec={prec}", json.dumps(st), flush=True)
            ctx.close()


def diag_propag(n_oct=24):
    g = G.build_grid(n_oct, mask="continents")
    for prec in ("dp", "sp"):
        cfg = Config(nang=24, nfre=36, nfre_red=29, idelpro=900)
        dt = H.np_dtype(prec)
        t = Tables(cfg, dt)
        o = Oracle(cfg, prec)
        rng = np.random.default_rng(5)
        from ecwam_amd import synthetic as syn
        depth = np.where(rng.uniform(0, 1, g.nsea) < 0.3, 10 ** rng.uniform(0.5, 3, g.nsea), 998.999)
        props = syn.depth_props(depth, t, dt)
        cg_ext = np.zeros((g.nsea + 1, cfg.nfre), dt)
        cg_ext[: g.nsea] = props["CGROUP"]
        cg_ext[g.nsea] = syn.depth_props(np.array([998.999]), t, dt)["CGROUP"][0]
        wref = o.ctu_weights(g, cg_ext, float(cfg.idelpro))
        f1 = np.zeros((g.nsea + 1, cfg.nang, cfg.nfre), dt)
        f1[: g.nsea] = rng.uniform(0, 1, (g.nsea, cfg.nang, cfg.nfre)) ** 4
        f3ref = o.propags2(g, f1, wref)
        ctx = api.HipContext(t)
        dev = ctx.device
        gd = api.grid_to_device(g, ctx.dtype, dev)
        w = torch.zeros((g.nsea, 8, cfg.nang * cfg.nfre_red), dtype=ctx.dtype, device=dev)
        fail = torch.zeros(g.nsea, dtype=torch.int32, device=dev)
        ctx.ctuw(gd, torch.from_numpy(cg_ext).to(dev), w, fail, float(cfg.idelpro))
        tf1 = torch.from_numpy(f1).to(dev)
        tf3 = torch.zeros_like(tf1)
        ctx.propags2(tf1, tf3, gd["klon"], gd["klat"], gd["kcor"], w, 0, g.nsea, check_indices=True)
        torch.cuda.synchronize()
        wg = w.cpu().numpy().reshape(g.nsea, 8, cfg.nang, cfg.nfre_red)
        # reference-shaped -> stored-8 selection
        jx = t.JXO[:, 0] - 1; jy = t.JYO[:, 0] - 1
        K = np.arange(cfg.nang)
        sel = [wref["SUMWN"], wref["WLONN"][:, K, :, jx].transpose(1, 0, 2), wref["WLATN"][:, K, :, jy, 0].transpose(1, 0, 2),
               wref["WLATN"][:, K, :, jy, 1].transpose(1, 0, 2), wref["WCORN"][:, :, :, 0, 0], wref["WCORN"][:, :, :, 0, 1],
               wref["WKPMN"][:, :, :, 0], wref["WKPMN"][:, :, :, 2]]
        errs = [float(np.max(np.abs(wg[:, i].astype(float) - sel[i].astype(float)))) for i in range(8)]
        f3 = tf3.cpu().numpy()
        e3 = float(np.max(np.abs(f3[: g.nsea, :, : cfg.nfre_red].astype(float) - f3ref[: g.nsea, :, : cfg.nfre_red].astype(float))))
        rest = float(np.max(np.abs(f3[: g.nsea, :, cfg.nfre_red:] - f1[: g.nsea, :, cfg.nfre_red:])))
        print(f"PROPAG prec={prec} nsea={g.nsea} weight_abs_err={errs} cflfail gpu={int(fail.sum())} ref={wref['NFAIL']} "
              f"f3_abs_err={e3} rest_copy_err={rest} wlat_mut_equal={bool(np.array_equal(gd['wlat'].cpu().numpy(), wref['WLAT']))}", flush=True)
        ctx.close()


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    diag_propag()
    diag_implsch()
