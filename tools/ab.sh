#!/bin/bash
# diagnostics: A/B timing of two builds of the library on the same GPU (ecwam_amd/lib/libecwam_hip_base.so vs the current one): IMPLSCH on
# 421 080 points, minimum of the launches of each of four alternating runs.  usage: bash tools/ab.sh [sp|dp]
for i in 1 2 3 4; do
  echo -n "base: "; ECWAM_HIP_LIB=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}/ecwam_amd/lib/libecwam_hip_base.so python3 tools/prof_implsch.py ${1:-sp} 421080 2>&1 | grep "implsch ms" | awk '{print $3}' | sort -n | head -1
  echo -n "new : "; python3 tools/prof_implsch.py ${1:-sp} 421080 2>&1 | grep "implsch ms" | awk '{print $3}' | sort -n | head -1
done
