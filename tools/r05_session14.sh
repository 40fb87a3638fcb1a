#!/bin/bash
# round 5, GPU session 14: the straddling pairs of the DIA windows / saturation filter as one shuffle (product) against the vectoriser's
# two v_mov_b32 (variant noshuf): IMPLSCH time at 131 072 points sp and dp, then the IMPLSCH parity tests on the product
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s14; mkdir -p "$O"
for prec in sp dp; do
for v in "" noshuf "" noshuf; do
  echo "== IMPLSCH 131072 $prec, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py $prec 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done
done | tee "$O/time_shuf.txt"
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 | tee "$O/parity.txt"
grep -q "passed" "$O/parity.txt" && ! grep -q "failed" "$O/parity.txt"
