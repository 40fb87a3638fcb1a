#!/bin/bash
# round 5, GPU session 8: the window form of the DIA gathers / scatters (V4_DIAWIN) against the separate rotated reads: parity, time, counters
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s8; mkdir -p "$O"
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py -q -m gpu > "$O/pytest.log" 2>&1; rc=$?; grep -E "passed|failed|^FAILED|^ERROR|^E  " "$O/pytest.log" | cut -c1-600 | tail -20
unset ECWAM_TEST_STATS_LOG
[ $rc -eq 124 ] && exit 124
grep -q "Memory access fault\|HSA_STATUS_ERROR" "$O/pytest.log" && exit 99
for v in "" diaold "" diaold "" diaold; do
  echo "== IMPLSCH O320 sp, library ${v:-product (window form, staged one interaction ahead)}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 421080 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -3
done | tee "$O/time_dia.txt"
for v in "" diaold; do
  echo "== IMPLSCH 131072 dp, library ${v:-product (window form, staged one interaction ahead)}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py dp 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee -a "$O/time_dia.txt"
for v in "" diaold; do
  export ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so"
  echo "== counters, library ${v:-product (window form, staged one interaction ahead)}"
  PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY" TAG=dia_${v:-win} N=131072 PREC=sp GEN=4 bash tools/pmc_run.sh
done | tee "$O/pmc_dia.txt"
exit 0
