#!/bin/bash
# Round-4 measurement set on ONE MI355X (writes gpurun_out/r04/*; the summaries are copied to profiles/ by hand):
#   the bench line (5 windows, median), rocprofv3 kernel stats of the same command, the HBM traffic PMC passes, the SQ counter sets of IMPLSCH.
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out
O=gpurun_out/r04; mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/bench_O320_sp.json 2> $O/bench_O320_sp.err || { echo "bench failed"; tail -5 $O/bench_O320_sp.err; exit 1; }
echo "bench done"; tail -c 600 $O/bench_O320_sp.json; echo
rm -rf $O/stats; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null || echo "stats failed"
cp $O/stats/*/*kernel_stats.csv $O/bench_O320_sp_kernel_stats.csv 2>/dev/null; head -8 $O/bench_O320_sp_kernel_stats.csv
bash tools/pmc_traffic.sh > $O/hbm_traffic_pmc.json 2>&1; cat $O/hbm_traffic_pmc.json
GEN=4 bash tools/pmc_implsch_sets.sh > $O/implsch_pmc_gen4.txt 2>&1; cat $O/implsch_pmc_gen4.txt
