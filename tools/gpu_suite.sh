#!/bin/bash
# the whole GPU suite in one process, full log under gpurun_out/<tag>/; usage: bash tools/gpu_suite.sh <tag> [pytest args]
cd "${GRAFT_REPO_ROOT:?}" || exit 2
tag=${1:-suite}; shift
O=gpurun_out/$tag; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider "$@" > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -25 $O/pytest.log | cut -c1-400
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest.log; then echo "GPU fault"; exit 99; fi
exit $rc
