#!/bin/bash
# diagnostics: A/B timing of two builds of the extension on the same GPU (baseline copy vs current)
for i in 1 2; do
  echo -n "base: "; ECWAM_HIP_LIB=$GRAFT_REPO_ROOT/ecwam_amd/lib/libecwam_hip_base.so python3 tools/prof_implsch.py ${1:-sp} 2>&1 | tail -1
  echo -n "new : "; python3 tools/prof_implsch.py ${1:-sp} 2>&1 | tail -1
done
