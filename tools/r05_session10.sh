#!/bin/bash
# round 5, GPU session 10: the O320 norm test over 24 steps (evidence run, not part of the suite), then the two new small tests
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s10; mkdir -p "$O"
ECWAM_NORM_STEPS=24 timeout -k 10 1000 python -m pytest tests/test_gpu_full_size.py -q -m gpu -s -k "swh_norms_after" > "$O/norms24.log" 2>&1; grep -E "step|swh norms|spectra|passed|failed|^E " "$O/norms24.log" | cut -c1-300
timeout -k 10 300 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity.py -q -m gpu -k "refuses" 2>&1 | tail -3
exit 0
