#!/bin/bash
# round 5, GPU session 13: instruction-cache counters of k_implsch4 (sp: 61.6 KB of code, dp: 102.7 KB; 64 KB of instruction cache per two CUs)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s13; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for prec in sp dp; do
  timeout -k 10 300 rocprofv3 --kernel-include-regex "k_implsch4<" --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$R/$O/ic_$prec" -- python3 "$R/tools/prof_implsch.py" $prec 131072 > "$R/$O/ic_$prec.log" 2>&1 || { echo "pmc $prec failed"; tail -5 "$R/$O/ic_$prec.log"; exit 1; }
done
cd "$R"
python3 - <<'PY'
import csv, glob, collections
for prec in ("sp", "dp"):
    f = glob.glob(f"gpurun_out/r05s13/ic_{prec}/*/*counter_collection.csv")
    agg = collections.defaultdict(float); n = set()
    for row in csv.DictReader(open(f[0])):
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); n.add(row["Dispatch_Id"])
    print(prec, "launches", len(n), {k: v / len(n) for k, v in agg.items()})
PY
