#!/bin/bash
# round 5, GPU session 12: -O2 against -O3 for the common builds in double precision (131 072 points)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s12; mkdir -p "$O"
for v in "" o2 "" o2; do
  echo "== IMPLSCH 131072 dp, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py dp 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee "$O/time_o2_dp.txt"
exit 0
