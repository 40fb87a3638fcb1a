"""The largest error of every gated statistic over a statistics log of the GPU suite (ECWAM_TEST_STATS_LOG, tests/test_gpu_parity.py::_log_stats),
grouped by precision and source-term time step: what the gates of _assert_implsch_stats are set from.
python tools/gate_report.py gpurun_out/r05s1/stats.jsonl"""
import collections
import json
import sys

KEYS = ("fl1_max_rel_peak_clean", "fl1_max_rel_peak_all", "swh_max_rel", "ff_max_rel_clean", "intf_max_rel_clean", "fl1_frac_sig_bins_gt_1e-5",
        "fl1_frac_bins_gt_1e-5", "mij_flips", "xllws_pts_diff")
grp = collections.defaultdict(lambda: collections.defaultdict(lambda: (0.0, "")))
cnt = collections.Counter()
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    g = (r["prec"], r.get("idelt", 0))
    cnt[g] += 1
    for k in KEYS:
        v = r.get(k, 0.0)
        if k in ("mij_flips", "xllws_pts_diff"):
            v = v / max(r.get("n", 1), 1)
        if v >= grp[g][k][0]:
            grp[g][k] = (v, r["test"].split("::")[-1].replace(" (call)", ""))
for g in sorted(grp):
    print(f"== precision {g[0]}, IDELT {g[1]} s: {cnt[g]} comparisons")
    for k in KEYS:
        v, t = grp[g][k]
        print(f"   {k:28s} {v:10.3e}   {t}")
