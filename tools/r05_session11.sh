#!/bin/bash
# round 5, GPU session 11: the bench line with the CPU baseline on the benchmark's own grid (O320: SURVEY 8d "same inputs, same step count")
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s11; mkdir -p "$O"
( while true; do sleep 60; echo "[progress] $(date +%T)"; done ) &
TICK=$!
timeout -k 10 1000 python3 bench.py --steps 20 --warmup 5 --cpu-baseline-grid 320 > "$O/bench_O320_sp_cpu320.json" 2> "$O/bench.err"; rc=$?
kill $TICK
tail -3 "$O/bench.err"; python3 -c "
import json; d=json.load(open('$O/bench_O320_sp_cpu320.json')); c=d['cpu_baseline']; print(d['value'], d['ms_per_step']); print({k: c[k] for k in ('value','cores','implsch_only','propags2_only','implsch_variant')}); print(c['sample']); print({k: c['dp'][k] for k in ('value','implsch_only','propags2_only','steps')})"
exit $rc
