#!/bin/bash
# round 6, session 1: the one-kernel step -- parity tests, then A/B of the two-kernel step, the one-kernel step and its go / no-go probe
cd "${GRAFT_REPO_ROOT:?}" || exit 2
mkdir -p gpurun_out/r06s1
O=gpurun_out/r06s1
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py -x -q -m gpu > $O/pytest_fused.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest_fused.log
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest_fused.log; then echo "GPU fault"; exit 99; fi
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_two_$i.json 2> $O/bench_two_$i.err; echo "two rc=$?"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --fused on > $O/bench_one_$i.json 2> $O/bench_one_$i.err; echo "one rc=$?"
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --fused on --fused-flags 1 > $O/bench_one_nat_$i.json 2> $O/bench_one_nat_$i.err; echo "one-nat rc=$?"
  ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_advprobe.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --fused on --fused-flags 2 > $O/bench_probe_$i.json 2> $O/bench_probe_$i.err; echo "probe rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06s1/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d["ms_per_step"],3), {k:round(v["ms"],3) for k,v in d["kernels"].items()}, d["finite"])
    except Exception as e: print(f, "ERR", e)
PY
