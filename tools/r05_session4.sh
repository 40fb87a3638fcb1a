#!/bin/bash
# round 5, GPU session 4: 48 directions and IPHYS 0 beside flag set B on k_implsch4 (the parity file), then the driver bench
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s4; mkdir -p "$O"
fault() { grep -q "Memory access fault\|HSA_STATUS_ERROR" "$1" && { echo "GPU runtime fault in $1"; grep -m3 "Memory access fault\|HSA_STATUS_ERROR" "$1"; return 0; }; return 1; }
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "48 or iphys_0 or benchmark_time_step or many_points or registered or alternate" > "$O/pytest.log" 2>&1; rc=$?; grep -E "passed|failed|^FAILED|^ERROR" "$O/pytest.log" | tail -30
[ $rc -eq 124 ] && exit 124; fault "$O/pytest.log" && exit 99
unset ECWAM_TEST_STATS_LOG
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$O/bench.json" 2> "$O/bench.err"; tail -c 1500 "$O/bench.json"
exit 0
