#!/usr/bin/env python3
"""Counter summary of the bench kernels, taken on the bench command itself (so that the summary describes the grid, spectrum and precision the
bench line is quoted on): rocprofv3 --pmc passes over `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline <args>`, each counter set in its
own pass (MI355X_MICROARCH.md: 8 SQ slots per pass, FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only beside --pmc.

python3 tools/pmc_bench.py out.json [bench.py arguments ...]

out.json: {"workload": what bench.py printed under config / dtype (+ grid, nang, nfre, prec, points), "kernels": {"implsch": {"name": the main
kernel's instantiation, "launches", per-launch sums of every counter, "per_point": SQ counters per sea point, "valu_busy", "lds_busy",
"waitcnt_fraction", "hbm_bytes": 1024 (2 FETCH_SIZE + WRITE_SIZE) -- FETCH_SIZE doubled as the guide prescribes for gfx950 --}, "propags2": ...}}.
bench.py --pmc-file reads it and refuses a summary whose workload or kernel differs from the run's."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1]
bench_args = sys.argv[2:]
SETS = {"sq_a": "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS",
        "sq_b": "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAVES",
        "fetch": "FETCH_SIZE", "write": "WRITE_SIZE"}
work = os.path.join(ROOT, "gpurun_out", "pmc_bench")
shutil.rmtree(work, ignore_errors=True)
os.makedirs(work)
os.chdir(ROOT)
env = dict(os.environ, TMPDIR="/tmp")
cmd = ["python3", "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *bench_args]
r = subprocess.run(cmd, capture_output=True, text=True, env=env)
line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
if r.returncode != 0 or not line:
    sys.exit("bench.py failed: " + r.stderr[-500:])
bj = json.loads(line[-1])
npts = bj["config"]["points_per_gpu"]
split_build = "split" in os.path.basename(os.environ.get("ECWAM_HIP_LIB", ""))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
names = {}
for tag, counters in SETS.items():
    d = os.path.join(work, tag)
    p = subprocess.run(["rocprofv3", "--kernel-include-regex", "implsch|propags2", "--pmc", *counters.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "--", *cmd],
                       capture_output=True, text=True, env=env)
    files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    if not files:
        print(f"pass {tag} failed: {p.stderr[-300:]}", file=sys.stderr)
        continue
    seen = collections.defaultdict(set)
    for row in csv.DictReader(open(files[0])):
        n = row["Kernel_Name"]
        k = "implsch" if "implsch" in n else ("propags2" if "propags2" in n else None)
        if not k:
            continue
        main = not ("implsch4_pre" in n or "implsch4_fin" in n)
        if k == "implsch" and split_build:      # the two-kernel split: one entry per part, the two scalar kernels on their own
            part = n.split(">(")[0].split(",")[-1].strip()
            k = f"implsch_part{part}" if main else "implsch_scalar_kernels"
            main = True
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if main:
            seen[k].add(row["Dispatch_Id"])
            names.setdefault(k, n.split("(")[0].replace("void ", ""))
    for k in seen:
        launches[(k, tag)] = len(seen[k])
import hashlib
sys.path.insert(0, ROOT)
from ecwam_amd import lib as _L
# what ties the summary to the code object that ran: bench.py attaches it only to runs of the same library file
lib_id = {"file": os.path.relpath(_L.LIBPATH, ROOT), "sha256": hashlib.sha256(open(_L.LIBPATH, "rb").read()).hexdigest(),
          "bench_py_sha256": hashlib.sha256(open(os.path.join(ROOT, "bench.py"), "rb").read()).hexdigest()}
try:
    lib_id["git_head"] = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or None      # (no .git on a GPU box)
except OSError:
    lib_id["git_head"] = None
out = {"workload": {"metric": bj["metric"], "dtype": bj["dtype"], "workload": bj["config"]["workload"], "points": npts, "bench_args": bench_args},
       "library": lib_id,
       "command": " ".join(cmd), "counter_sets": SETS, "kernels": {}}
for k in agg:
    per_launch = {}
    for tag, counters in SETS.items():
        nl = launches.get((k, tag), 0)
        for c in counters.split():
            if nl and c in agg[k]:
                per_launch[c] = agg[k][c] / nl
    e = {"name": names.get(k), "launches_per_pass": {t: launches.get((k, t), 0) for t in SETS}, "per_launch": per_launch}
    pp = {c: v / npts for c, v in per_launch.items() if c.startswith("SQ_")}
    e["per_point"] = pp
    if "FETCH_SIZE" in per_launch and "WRITE_SIZE" in per_launch:
        e["hbm_bytes"] = 1024.0 * (2.0 * per_launch["FETCH_SIZE"] + per_launch["WRITE_SIZE"])
        e["hbm_read_bytes"], e["hbm_write_bytes"] = 2048.0 * per_launch["FETCH_SIZE"], 1024.0 * per_launch["WRITE_SIZE"]
    if k.startswith("implsch") and k != "implsch_scalar_kernels" and "SQ_WAVE_CYCLES" in pp:
        wps = 2.0 if bj["dtype"] == "f32" else 1.0      # resident waves per SIMD of k_implsch4 (LDS: 8 / 4 waves per CU)
        e["resident_waves_per_simd"] = wps
        e["valu_busy"] = pp["SQ_ACTIVE_INST_VALU"] / (pp["SQ_WAVE_CYCLES"] / wps)
        e["waitcnt_fraction"] = pp["SQ_WAIT_ANY"] / pp["SQ_WAVE_CYCLES"]
        e["cycles_per_valu_instruction_while_active"] = 4.0 * pp["SQ_ACTIVE_INST_VALU"] / pp["SQ_INSTS_VALU"]
        if "SQ_LDS_IDX_ACTIVE" in pp:
            e["lds_busy"] = pp["SQ_LDS_IDX_ACTIVE"] / (pp["SQ_WAVE_CYCLES"] * 4.0 / (4.0 * wps))
            e["lds_bank_conflict_share"] = pp["SQ_LDS_BANK_CONFLICT"] / pp["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][k] = e
with open(out_path, "w") as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: {x: v[x] for x in ("name", "valu_busy", "lds_busy", "hbm_bytes") if x in v} for k, v in out["kernels"].items()}, indent=1))
