#!/bin/bash
# Round-2 measurement set on one MI355X (writes gpurun_out/r02/*; the summaries are copied to profiles/ by hand):
#   bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes, IMPLSCH SQ counter sets, FETCH_SIZE calibration
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
O=gpurun_out/r02; mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/bench_O320_sp.json 2> $O/bench_O320_sp.err || echo "bench failed"
echo "bench done"; tail -c 600 $O/bench_O320_sp.json
rm -rf $O/stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 || echo "stats failed"
cp $O/stats/*/*kernel_stats.csv $O/bench_O320_sp_kernel_stats.csv 2>/dev/null; head -8 $O/bench_O320_sp_kernel_stats.csv
bash tools/pmc_traffic.sh > $O/hbm_traffic_pmc.json 2>&1; cat $O/hbm_traffic_pmc.json
GEN=4 bash tools/pmc_implsch_sets.sh > $O/implsch_pmc_gen4.txt 2>&1; cat $O/implsch_pmc_gen4.txt
hipcc --offload-arch=gfx950 -O3 -o /tmp/calib_copy tools/calib_copy.hip 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  d=$O/calib_$(echo $c | cut -c1-5); rm -rf $d
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- /tmp/calib_copy > /dev/null 2>&1 || echo "calib pass $c failed"
done
python3 - <<'PY' > gpurun_out/r02/fetch_size_calibration.json
import csv,glob,collections,json
out=collections.defaultdict(dict)
for f in glob.glob("gpurun_out/r02/calib_*/*/*counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "calib" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k in agg:
        for c,v in agg[k].items(): out[k][c]=v/len(cnt[k])
out["bytes_per_launch"]={"read":421080*36*36*4,"written":421080*36*36*4}
print(json.dumps(out,indent=1))
PY
cat $O/fetch_size_calibration.json
