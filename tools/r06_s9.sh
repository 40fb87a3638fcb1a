#!/bin/bash
# round 6, session 9: the refraction stencil as two launches (hoisted form on every lane, the general kernel on the tiles with an upwind switch)
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s9; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_refraction.py tests/test_gpu_fortran.py tests/test_gpu_propag_wam.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log | cut -c1-600
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest.log; then echo "GPU fault"; exit 99; fi
[ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift; timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
run O320_sp_irefra2 --irefra 2 --steps 10 --warmup 2 --repeats 3
ECWAM_HIP_GEN_FAST_VW=2 run O320_sp_irefra2_vw2 --irefra 2 --steps 10 --warmup 2 --repeats 3
run O320_sp_irefra3 --irefra 3 --steps 10 --warmup 2 --repeats 3
run O320_dp_irefra2 --irefra 2 --prec dp --steps 6 --warmup 2 --repeats 3
exit 0
