#!/bin/bash
# round 5, GPU session 3: the whole GPU suite on the product (new gates, the O320 four-step norm test, the many-point RARE tests in both
# precisions), host memory of the box, then the counters of the split build per kernel
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s3; mkdir -p "$O"
free -g | head -2; nproc
fault() { grep -q "Memory access fault\|HSA_STATUS_ERROR" "$1" && { echo "GPU runtime fault in $1"; grep -m3 "Memory access fault\|HSA_STATUS_ERROR" "$1"; return 0; }; return 1; }
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 1100 python -m pytest tests -q -m gpu -s --durations=15 > "$O/pytest.log" 2>&1; rc=$?; grep -E "swh norms|spectra:|passed|failed|^FAILED|^ERROR" "$O/pytest.log" | tail -40
[ $rc -eq 124 ] && exit 124; fault "$O/pytest.log" && exit 99
exit 0
