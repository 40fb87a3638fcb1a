#!/bin/bash
# round 6, session 2: the one-kernel step after the instruction diet of its advecting load; gather depth 2 / 3 / 4
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s2; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_fused.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_fused.log
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest_fused.log; then echo "GPU fault"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3"
for i in 1 2; do
  timeout -k 10 300 $B > $O/bench_two_$i.json 2> $O/bench_two_$i.err; echo "two rc=$?"
  timeout -k 10 300 $B --fused on > $O/bench_one_$i.json 2> $O/bench_one_$i.err; echo "one rc=$?"
  timeout -k 10 300 $B --fused on --fused-flags 1 > $O/bench_one_nat_$i.json 2> $O/bench_one_nat_$i.err; echo "one-nat rc=$?"
  for v in advd2 advd4; do
    ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_$v.so timeout -k 10 300 $B --fused on --fused-flags 1 > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err; echo "$v rc=$?"
  done
  ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_advprobe.so timeout -k 10 300 $B --fused on --fused-flags 3 > $O/bench_probe_$i.json 2> $O/bench_probe_$i.err; echo "probe rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06s2/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d["ms_per_step"],3), {k:round(v["ms"],3) for k,v in d["kernels"].items()}, d["finite"])
    except Exception as e: print(f, "ERR", e)
PY
