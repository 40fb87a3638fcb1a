#!/bin/bash
# round 6, session 3: fused step in dp and from Fortran; dp bench at O320 two / one
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s3; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fortran.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest.log; then echo "GPU fault"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
B="python bench.py --no-cpu-baseline --steps 10 --warmup 2"
timeout -k 10 300 $B --prec dp > $O/bench_dp_two.json 2> $O/bench_dp_two.err; echo "dp two rc=$?"
timeout -k 10 300 $B --prec dp --fused on > $O/bench_dp_one.json 2> $O/bench_dp_one.err; echo "dp one rc=$?"
timeout -k 10 300 $B > $O/bench_sp_two.json 2> $O/bench_sp_two.err; echo "sp two rc=$?"
timeout -k 10 300 $B --fused on > $O/bench_sp_one.json 2> $O/bench_sp_one.err; echo "sp one rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06s3/bench_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d["ms_per_step"],3), {k:round(v["ms"],3) for k,v in d["kernels"].items()}, d["finite"])
    except Exception as e: print(f, "ERR", e)
PY
