#!/bin/bash
# HBM traffic of the bench kernels (MI355X_MICROARCH.md "HBM traffic"): FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (3 + 2 TCC counters do not fit one pass), per-launch averages in bytes; FETCH_SIZE doubled for the 16 B/lane streams (gfx950).
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
rm -rf gpurun_out/traf_*
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  d=gpurun_out/traf_$(echo $c | cut -c1-5)
  timeout 150 rocprofv3 --kernel-include-regex "implsch|propags2" --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv,glob,collections,json
out=collections.defaultdict(dict)
for f in glob.glob("gpurun_out/traf_*/*/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="implsch" if "implsch" in n else ("propags2" if "propags2" in n else None)
        if not k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if "implsch4_pre" not in n and "implsch4_fin" not in n: cnt[k].add(r["Dispatch_Id"])   # one IMPLSCH call = three kernels
    for k in agg:
        for c,v in agg[k].items(): out[k][c]=v/len(cnt[k])
print(json.dumps(out,indent=1))
PY
