#!/bin/bash
# Round-2 measurement set, part 2 (one MI355X): the other BASELINE shapes through bench.py and the IMPLSCH generations side by side.
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out; O=gpurun_out/r02; mkdir -p $O
for cfg in "640 sp" "1280 sp" "640 dp" "1280 dp"; do
  set -- $cfg
  timeout -k 10 280 python3 bench.py --grid $1 --prec $2 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_O$1_$2.json 2> $O/bench_O$1_$2.err || echo "bench $cfg failed"
  python3 -c "import json,sys; d=json.load(open('$O/bench_O$1_$2.json')); print('O$1 $2', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"
done
timeout -k 10 250 python3 bench.py --grid 1280 --ifrelfmax 5 --adv-per-source 2 --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_O1280_sp_native.json 2> $O/bench_O1280_sp_native.err || echo "native failed"
timeout -k 10 120 python3 bench.py --irefra 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_O320_sp_irefra2.json 2> $O/bench_O320_sp_irefra2.err || echo "irefra failed"
for f in A B; do timeout -k 10 300 python3 tests/diag/implsch_gens.py 131072 sp,dp 36,24,12 $f > $O/gens_$f.txt 2>&1 || echo "gens $f failed"; grep -h "ms" $O/gens_$f.txt | head -20; done
