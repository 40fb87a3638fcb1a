#!/bin/bash
# round 5, final GPU session: the whole GPU suite + smoke on the final source, then the measurement set again (profiles/r05_* are copied from it)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05final; mkdir -p "$O"
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 1100 python -m pytest tests -q -m gpu --durations=8 > "$O/pytest.log" 2>&1; rc=$?; grep -E "passed|failed|^FAILED|^ERROR|^E  " "$O/pytest.log" | cut -c1-500 | tail -20
unset ECWAM_TEST_STATS_LOG
[ $rc -eq 124 ] && exit 124
grep -q "Memory access fault\|HSA_STATUS_ERROR" "$O/pytest.log" && exit 99
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/r05_profile.sh
