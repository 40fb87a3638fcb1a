#!/bin/bash
# diagnostics: PROPAGS2 at O320 in natural order and with the 2-D tiles of decomp.tile2d_order (bench.py --strip -1); the spectra
# after 12 steps must be bit-identical
mkdir -p gpurun_out/r02
for W in 0 -1; do
  echo -n "strip $W: "
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --strip $W --no-cpu-baseline --dump gpurun_out/r02/dump_s$W 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('propags2 ms', round(d['kernels']['propags2']['ms'],3), 'step ms', round(d['ms_per_step'],3))"
done
python3 -c "
import numpy as np
a=np.load('gpurun_out/r02/dump_s0.0.npy'); b=np.load('gpurun_out/r02/dump_s-1.0.npy'); print('identical', np.array_equal(a,b))
"
rm -f gpurun_out/r02/dump_s*
