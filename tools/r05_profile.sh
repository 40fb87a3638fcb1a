#!/bin/bash
# Round-5 measurement set on ONE MI355X (writes gpurun_out/r05/*; the summaries are copied to profiles/r05_* by hand):
#   the counter summary of the bench command (tools/pmc_bench.py), the bench line with it attached (5 windows, median), rocprofv3 kernel
#   stats of the same command, the same for the "split" build of the library (the round's prototype), the other BASELINE shapes.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 600 python3 tools/pmc_bench.py $O/bench_O320_sp_pmc.json > $O/pmc_bench.log 2>&1 || { echo "pmc_bench failed"; tail -5 $O/pmc_bench.log; }
tail -12 $O/pmc_bench.log
python3 bench.py --steps 20 --warmup 5 --pmc-file $O/bench_O320_sp_pmc.json > $O/bench_O320_sp.json 2> $O/bench_O320_sp.err || { echo "bench failed"; tail -5 $O/bench_O320_sp.err; exit 1; }
echo "bench done"; python3 -c "import json; d=json.load(open('$O/bench_O320_sp.json')); print(d['value'], d['ms_per_step'], d['kernels'], d['roofline'])"
rm -rf $O/stats; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null || echo "stats failed"
cp $O/stats/*/*kernel_stats.csv $O/bench_O320_sp_kernel_stats.csv 2>/dev/null; head -6 $O/bench_O320_sp_kernel_stats.csv | cut -c1-260
# the split build: counters and kernel stats of the same command
if [ -f "$PWD/ecwam_amd/lib/libecwam_hip_split.so" ]; then      # (python -m ecwam_amd.build --variant=split; not kept in the tree)
export ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip_split.so"
timeout -k 10 600 python3 tools/pmc_bench.py $O/bench_O320_sp_split_pmc.json > $O/pmc_bench_split.log 2>&1 || { echo "pmc_bench (split) failed"; tail -5 $O/pmc_bench_split.log; }
rm -rf $O/stats_split; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_split -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_split_under_rocprof.json 2>/dev/null || echo "stats (split) failed"
cp $O/stats_split/*/*kernel_stats.csv $O/bench_O320_sp_split_kernel_stats.csv 2>/dev/null; head -6 $O/bench_O320_sp_split_kernel_stats.csv | cut -c1-260
unset ECWAM_HIP_LIB
fi
run() { tag=$1; shift; timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
run O640_sp --grid 640 --steps 5 --warmup 1 --repeats 3
run O1280_sp --grid 1280 --steps 5 --warmup 1 --repeats 3
run O1280_dp --grid 1280 --prec dp --steps 4 --warmup 1 --repeats 3
run O1280_sp_native --grid 1280 --ifrelfmax 5 --adv-per-source 2 --steps 4 --warmup 1 --repeats 3
run O320_sp_irefra2 --irefra 2 --steps 10 --warmup 2 --repeats 3
exit 0
