#!/bin/bash
# round 6, session 12: the one-kernel step with sub-grid obstructions (LSUBGRID)
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s12; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_fortran.py tests/test_gpu_refraction.py -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest.log | cut -c1-600
if grep -q "HSA_STATUS_ERROR\|Memory access fault" $O/pytest.log; then echo "GPU fault"; exit 99; fi
exit $rc
