#!/bin/bash
# diagnostics: memory-path counters of the advection kernel in bench.py, one --pmc pass per group
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
rm -rf gpurun_out/memp_*
i=0
for c in "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"; do
  i=$((i+1)); d=gpurun_out/memp_$i
  timeout 150 rocprofv3 --kernel-include-regex "propags2" --pmc $c --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1 || echo "pass $c failed"
done
python3 - <<'PY'
import csv,glob,collections,json
out={}
for f in glob.glob("gpurun_out/memp_*/*/*counter_collection.csv"):
    agg=collections.defaultdict(float); cnt=set()
    for r in csv.DictReader(open(f)):
        if "propags2" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt.add(r["Dispatch_Id"])
    for c,v in agg.items(): out[c]=v/max(len(cnt),1)
print(json.dumps(out,indent=1))
PY
