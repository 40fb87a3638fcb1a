"""The largest error of every gated statistic over a statistics log of the GPU suite (ECWAM_TEST_STATS_LOG, tests/test_gpu_parity.py::_log_stats),
grouped by precision and source-term time step: what the gates of _assert_implsch_stats are set from.
python tools/gate_report.py gpurun_out/r05s1/stats.jsonl [more logs ...]"""
import collections
import json
import sys

KEYS = ("fl1_rob_rel_peak", "fl1_max_rel_peak_clean", "fl1_max_rel_peak_all", "swh_rob_rel", "swh_max_rel", "ff_rob_rel", "ff_max_rel_clean", "ff_max_rel_all",
        "intf_rob_rel", "intf_max_rel_clean", "intf_max_rel_all", "fl1_frac_sig_bins_gt_1e-5", "fl1_frac_bins_gt_1e-5", "mij_flips", "xllws_pts_diff")
grp = collections.defaultdict(lambda: collections.defaultdict(lambda: (0.0, "")))
cnt = collections.Counter()
for ln in (ln for f in sys.argv[1:] for ln in open(f)):      # several logs (e.g. one per seed offset): the maxima over all of them
    r = json.loads(ln)
    if "implsch2" in r["test"] or "generations_agree" in r["test"] or "older_kernel" in r["test"]:
        continue      # comparisons of the two device implementations with each other, or of the tests' second implementation: other gates
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
