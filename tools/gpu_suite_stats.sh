#!/bin/bash
# the GPU suite with the statistics log of every IMPLSCH comparison (tests/harness.py: ECWAM_TEST_STATS_LOG), on the default inputs and on a
# second set of random inputs (ECWAM_TEST_SEED_OFFSET); usage: bash tools/gpu_suite_stats.sh <tag> <seed offset> [pytest args]
cd "${GRAFT_REPO_ROOT:?}" || exit 2
tag=${1:-stats}; off=${2:-0}; shift; shift
O=gpurun_out/$tag; mkdir -p $O; rm -f $O/stats_$off.jsonl
ECWAM_TEST_STATS_LOG=$PWD/$O/stats_$off.jsonl ECWAM_TEST_SEED_OFFSET=$off timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > $O/pytest_$off.log 2>&1; rc=$?
echo "pytest (seed offset $off) rc=$rc"; tail -12 $O/pytest_$off.log | cut -c1-600
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest_$off.log; then echo "GPU fault"; exit 99; fi
exit $rc
