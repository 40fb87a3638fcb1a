#!/bin/bash
# round 5, GPU session 1: the GPU suite with the statistics log of every gated comparison, the occupancy sweep of k_implsch4, and the double
# precision RARE builds at -O3 / -O2 / -O1 / with index checks (one process each: a faulting kernel ends its process)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s1; mkdir -p "$O"
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$O/pytest.log" 2>&1; rc=$?; tail -3 "$O/pytest.log"; [ $rc -eq 124 ] && exit 124
unset ECWAM_TEST_STATS_LOG
timeout -k 10 300 python tools/occupancy_sweep4.py sp 131072 > "$O/occupancy_sp.txt" 2>&1; rc=$?; cat "$O/occupancy_sp.txt"; [ $rc -eq 124 ] && exit 124
for v in rdp rdpO2 rdpO1 rdpchk; do
  echo "== variant $v"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip_$v.so" timeout -k 10 180 python tests/diag/rare_dp_probe.py 24 512 dp > "$O/probe_$v.log" 2>&1; rc=$?
  echo "rc=$rc"; tail -25 "$O/probe_$v.log"
  [ $rc -eq 124 ] && exit 124
done
exit 0
