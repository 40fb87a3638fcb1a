#!/bin/bash
# round 6, session 13: LSUBGRID cost in the one-kernel step and in the two kernels; then the whole GPU suite on the final library
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s13; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
run subgrid_one --subgrid --steps 20 --warmup 3
run subgrid_two --subgrid --steps 20 --warmup 3 --fused off
run plain_one --steps 20 --warmup 3
bash tools/gpu_suite.sh r06final3
