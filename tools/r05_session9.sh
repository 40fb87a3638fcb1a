#!/bin/bash
# round 5, GPU session 9: the staggered first generation of k_implsch4 waves against the product
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s9; mkdir -p "$O"
for v in "" stagger stagger6 "" stagger stagger6 "" stagger stagger6; do
  echo "== IMPLSCH O320 sp, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 421080 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -3
done | tee "$O/time_stagger.txt"
exit 0
