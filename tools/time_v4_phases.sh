#!/bin/bash
# diagnostics: cumulative time of k_implsch4 up to each phase boundary (needs tools/build_diag.sh). usage: [N=131072] [PREC=sp] [FLAGSET=A|B] bash tools/time_v4_phases.sh
export ECWAM_HIP_LIB="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"/ecwam_amd/lib/libecwam_hip_diag.so
for m in 201 202 203 204 205 206 207 208 209 210 211 0; do
  echo -n "exit $m: "; ECWAM_HIP_DEBUG_SKIP=$m python3 tools/prof_implsch.py ${PREC:-sp} ${N:-131072} 4 ${FLAGSET:-A} 2>/dev/null | tail -1
done
