#!/bin/bash
# diagnostics: timing of several builds of the library on the same GPU, IMPLSCH on 421 080 points, minimum of the launches of each of
# four alternating runs.  usage: bash tools/ab3.sh sp <lib suffix> [<lib suffix> ...]   ("" = the product library, "base" = libecwam_hip_base.so ...)
prec=$1; shift
for i in 1 2 3 4; do
  for v in "$@"; do
    lib=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}/ecwam_amd/lib/libecwam_hip${v:+_$v}.so
    echo -n "${v:-product}: "; ECWAM_HIP_LIB=$lib python3 tools/prof_implsch.py $prec 421080 2>&1 | grep "implsch ms" | awk '{print $3}' | sort -n | head -1
  done
done
