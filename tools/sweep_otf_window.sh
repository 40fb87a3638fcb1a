#!/bin/bash
# diagnostics: PROPAGS2 (k_propags2_otf) time at O320 against the processing order (longitude strips of W points) and the number of
# workgroups in flight (the window of spectra an XCD touches at a time).  Needs tools/build_diag.sh.  usage: bash tools/sweep_otf_window.sh
export ECWAM_HIP_LIB=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}/ecwam_amd/lib/libecwam_hip_diag.so
for W in 0 32 64 128; do
  for G in 256 512 1024 2048 4096; do
    echo -n "strip $W grid $G: "
    ECWAM_HIP_OTF_GRID=$G python3 bench.py --steps 10 --warmup 2 --strip $W --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('propags2 ms', round(d['kernels']['propags2']['ms'],3), 'step ms', round(d['ms_per_step'],3))"
  done
done
