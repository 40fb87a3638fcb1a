#!/bin/bash
# Round-6 measurement set on ONE MI355X (writes gpurun_out/r06/*; the summaries are copied to profiles/r06_* by hand):
#   the counter summary of the bench command (tools/pmc_bench.py: --pmc passes, tied to the library's sha256), the bench line with it attached
#   (5 windows, median), rocprofv3 kernel stats of the same command; the same for the two-kernel step (--fused off: the A/B partner);
#   the other BASELINE shapes.  usage: bash tools/r06_profile.sh [quick]
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r06; mkdir -p $O
set_of() { tag=$1; shift      # counters, bench line, kernel stats of `bench.py "$@"`
  timeout -k 10 600 python3 tools/pmc_bench.py $O/bench_${tag}_pmc.json "$@" > $O/pmc_bench_$tag.log 2>&1 || { echo "pmc_bench $tag failed"; tail -5 $O/pmc_bench_$tag.log; }
  tail -12 $O/pmc_bench_$tag.log
  python3 bench.py --steps 20 --warmup 5 --pmc-file $O/bench_${tag}_pmc.json "$@" > $O/bench_$tag.json 2> $O/bench_$tag.err || { echo "bench $tag failed"; tail -5 $O/bench_$tag.err; return 1; }
  python3 -c "import json; d=json.load(open('$O/bench_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['kernels'], d['roofline'], d.get('cpu_baseline'))"
  rm -rf $O/stats_$tag; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/bench_${tag}_under_rocprof.json 2>/dev/null || echo "stats $tag failed"
  cp $O/stats_$tag/*/*kernel_stats.csv $O/bench_${tag}_kernel_stats.csv 2>/dev/null; head -7 $O/bench_${tag}_kernel_stats.csv | cut -c1-260; }
set_of O320_sp
set_of O320_sp_two_kernels --fused off --no-cpu-baseline
[ "$1" = quick ] && exit 0
run() { tag=$1; shift; timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
run O640_sp --grid 640 --steps 5 --warmup 1 --repeats 3
run O1280_sp --grid 1280 --steps 5 --warmup 1 --repeats 3
run O1280_dp --grid 1280 --prec dp --steps 4 --warmup 1 --repeats 3
run O1280_dp_two_kernels --grid 1280 --prec dp --steps 4 --warmup 1 --repeats 3 --fused off
run O1280_sp_native --grid 1280 --ifrelfmax 5 --adv-per-source 2 --steps 4 --warmup 1 --repeats 3
run O320_sp_irefra2 --irefra 2 --steps 10 --warmup 2 --repeats 3
exit 0
