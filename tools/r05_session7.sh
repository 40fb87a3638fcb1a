#!/bin/bash
# round 5, GPU session 7: the split row layout of the 36-direction tile (V4_ROWSPLIT) against the contiguous one: parity, time, LDS counters
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s7; mkdir -p "$O"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "36 or benchmark or generations or registered" > "$O/pytest.log" 2>&1; rc=$?; grep -E "passed|failed|^FAILED|^ERROR|^E  " "$O/pytest.log" | cut -c1-400 | tail
[ $rc -ne 0 ] && exit 1
for v in "" rowcontig "" rowcontig "" rowcontig; do
  echo "== IMPLSCH O320 sp, library ${v:-product (split rows)}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 421080 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -3
done | tee "$O/time_rows.txt"
for v in "" rowcontig; do
  export ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so"
  echo "== LDS counters, library ${v:-product (split rows)}"
  PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY" TAG=rows_${v:-split} N=131072 PREC=sp GEN=4 bash tools/pmc_run.sh
done | tee "$O/pmc_rows.txt"
exit 0
