#!/bin/bash
# L2 hit/miss and EA read requests of the bench kernels for given bench options
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
rm -rf gpurun_out/tcc_x
timeout 100 rocprofv3 --kernel-include-regex "implsch|propags2" --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/tcc_x -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/tcc_x/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    k="implsch" if "implsch" in n else ("propags2" if "propags2" in n else None)
    if not k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in agg: print(k, {c: round(v/len(cnt[k])) for c,v in agg[k].items()})
PY
