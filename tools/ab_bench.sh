#!/bin/bash
# diagnostics: bench.py kernel times with several builds of the library on the same GPU.  usage: bash tools/ab_bench.sh "<bench args>" <lib suffix> ...
args=$1; shift
for i in 1 2; do
  for v in "$@"; do
    lib=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}/ecwam_amd/lib/libecwam_hip${v:+_$v}.so
    echo -n "${v:-product}: "; ECWAM_HIP_LIB=$lib python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v['ms'],3) for k,v in d['kernels'].items()})"
  done
done
