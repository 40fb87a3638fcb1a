#!/bin/bash
# diagnostics: SQ-level counters of the advection kernel in bench.py (what the waves wait on), one --pmc pass per counter group
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
rm -rf gpurun_out/sqp_*
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); d=gpurun_out/sqp_$i
  timeout 150 rocprofv3 --kernel-include-regex "propags2" --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv,glob,collections,json
out={}
for f in glob.glob("gpurun_out/sqp_*/*/*counter_collection.csv"):
    agg=collections.defaultdict(float); cnt=set()
    for r in csv.DictReader(open(f)):
        if "propags2" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt.add(r["Dispatch_Id"])
    for c,v in agg.items(): out[c]=v/max(len(cnt),1)
print(json.dumps(out,indent=1))
PY
