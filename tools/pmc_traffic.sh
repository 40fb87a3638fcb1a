#!/bin/bash
# HBM traffic of the bench kernels (MI355X_MICROARCH.md, "HBM traffic with rocprofv3"): separate --pmc passes on
# `bench.py --steps 3 --warmup 1`, per-launch averages, gfx950 FETCH_SIZE correction applied in the summary.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/traf_*
timeout 300 rocprofv3 --kernel-include-regex 'implsch|propags2' --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/traf_a -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-include-regex 'implsch|propags2' --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d gpurun_out/traf_b -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections,json
out=collections.defaultdict(dict)
for d in ("a","b"):
    f=glob.glob(f"gpurun_out/traf_{d}/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="implsch" if "implsch" in n else ("propags2" if "propags2" in n else None)
        if not k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in agg:
        for c,v in agg[k].items(): out[k][c]=v/len(cnt[k])
print(json.dumps(out,indent=1))
PY
