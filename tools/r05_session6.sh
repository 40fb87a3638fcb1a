#!/bin/bash
# round 5, GPU session 6: the Fortran host layer tests (with the RCCL self-peer case), then the measurement set
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s6; mkdir -p "$O"
timeout -k 10 900 python -m pytest tests/test_gpu_fortran.py tests/test_gpu_multirank.py -q -m gpu -x > "$O/pytest_fortran.log" 2>&1; rc=$?; grep -E "passed|failed|^FAILED|^ERROR|^E  " "$O/pytest_fortran.log" | cut -c1-500 | tail -20
[ $rc -eq 124 ] && exit 124
grep -q "Memory access fault\|HSA_STATUS_ERROR" "$O/pytest_fortran.log" && exit 99
bash tools/r05_profile.sh
for v in "" o2 "" o2; do
  echo "== IMPLSCH O320 sp, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 421080 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee gpurun_out/r05/time_o2_vs_o3_sp.txt
