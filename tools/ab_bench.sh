#!/bin/bash
# diagnostics: A/B of two builds of the library through bench.py (ecwam_amd/lib/libecwam_hip_base.so vs the current one): ms per step and per kernel
for i in 1 2 3; do
  for w in base new; do
    if [ $w = base ]; then export ECWAM_HIP_LIB=$GRAFT_REPO_ROOT/ecwam_amd/lib/libecwam_hip_base.so; else unset ECWAM_HIP_LIB; fi
    echo -n "$w: "; python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})"
  done
done
