#!/bin/bash
# round 5, GPU session 18: the whole GPU suite on OTHER random inputs than the ones its gates were set on (tests/conftest.py,
# ECWAM_TEST_SEED_OFFSET): evidence for profiles/r05_seed_robustness.txt, not part of the suite's contract
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s18; mkdir -p "$O"
for off in ${OFFSETS:-1000}; do
  export ECWAM_TEST_SEED_OFFSET=$off ECWAM_TEST_STATS_LOG="$PWD/$O/stats_$off.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
  timeout -k 10 1000 python -m pytest tests -q -m gpu > "$O/pytest_$off.log" 2>&1; rc=$?
  echo "== seed offset $off: rc $rc"; grep -E "passed|failed|^FAILED|^ERROR" "$O/pytest_$off.log" | cut -c1-300 | tail -25
  [ $rc -eq 124 ] && exit 124
  grep -q "Memory access fault\|HSA_STATUS_ERROR" "$O/pytest_$off.log" && exit 99
done
exit 0
