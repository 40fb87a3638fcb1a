#!/bin/bash
# round 6, session 7: the native O1280 cycle with the last advection step inside the source-term kernel; gates on the default inputs
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s7; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -x -q -m gpu -s -k "fused or multirank or one_kernel or strict or bench_ or benchmark_time_step or iphys_0 or edge_cases or implsch2 or refuses" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep "benchmark step" $O/pytest.log | cut -c1-400; tail -4 $O/pytest.log | cut -c1-300
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest.log; then echo "GPU fault"; exit 99; fi
run() { tag=$1; shift; timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
run O1280_sp_native --grid 1280 --ifrelfmax 5 --adv-per-source 2 --steps 4 --warmup 1 --repeats 3
run O1280_sp_native_two --grid 1280 --ifrelfmax 5 --adv-per-source 2 --steps 4 --warmup 1 --repeats 3 --fused off
run O1280_dp_native --grid 1280 --prec dp --ifrelfmax 5 --adv-per-source 2 --steps 3 --warmup 1 --repeats 3
run O320_sp --steps 20 --warmup 3
exit 0
