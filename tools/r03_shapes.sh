#!/bin/bash
# Round-3 measurement set, part 2 (one MI355X): the other BASELINE shapes through bench.py, advection work orders at O1280, the IMPLSCH
# kernel generations side by side (flag sets A and B, IPHYS = 0, ISNONLIN = 1).
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out; O=gpurun_out/r03; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 280 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
run O640_sp --grid 640 --steps 5 --warmup 1
run O1280_sp --grid 1280 --steps 5 --warmup 1
run O1280_sp_tiles2d --grid 1280 --steps 5 --warmup 1 --strip -1
run O1280_dp --grid 1280 --prec dp --steps 4 --warmup 1
run O1280_sp_native --grid 1280 --ifrelfmax 5 --adv-per-source 2 --steps 4 --warmup 1
run O320_sp_irefra2 --irefra 2 --steps 10 --warmup 2
for f in A B J E; do timeout -k 10 300 python3 tests/diag/implsch_gens.py 131072 sp,dp 36,24,12 $f > $O/gens_$f.txt 2>&1 || echo "gens $f failed"; grep -h "ms" $O/gens_$f.txt | head -20; done
