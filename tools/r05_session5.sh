#!/bin/bash
# round 5, GPU session 5: the whole GPU suite after k_implsch2 moved to tests/csrc (second implementation through tests/v2lib.py), smoke, bench
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s5; mkdir -p "$O"
fault() { grep -q "Memory access fault\|HSA_STATUS_ERROR" "$1" && { echo "GPU runtime fault in $1"; grep -m3 "Memory access fault\|HSA_STATUS_ERROR" "$1"; return 0; }; return 1; }
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 1100 python -m pytest tests -q -m gpu -x --durations=10 > "$O/pytest.log" 2>&1; rc=$?; grep -E "passed|failed|^FAILED|^ERROR|^E  " "$O/pytest.log" | cut -c1-600 | tail -30
[ $rc -eq 124 ] && exit 124; fault "$O/pytest.log" && exit 99
unset ECWAM_TEST_STATS_LOG
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
exit 0
