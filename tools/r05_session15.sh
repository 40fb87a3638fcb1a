#!/bin/bash
# round 5, GPU session 15: no straddling pair assembled (product: V4_WINSHUF = 2, vectorising passes off) against one shuffle per pair
# (shuf1) and the vectoriser's two moves (noshuf, rounds 2 - 4): time, bit comparison of the outputs, parity tests
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s15; mkdir -p "$O"
for prec in sp dp; do
for v in "" noshuf shuf1 "" noshuf shuf1; do
  echo "== IMPLSCH 131072 $prec, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py $prec 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done
done | tee "$O/time.txt"
for prec in sp dp; do
  for v in "" noshuf; do
    ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/implsch_dump.py $prec 8190 "$O/out_${prec}_${v:-product}.npz" > /dev/null 2>&1 || exit 1
  done
  echo "== outputs, $prec: product against noshuf"
  python3 tools/implsch_dump.py --compare "$O/out_${prec}_product.npz" "$O/out_${prec}_noshuf.npz"
done | tee "$O/bits.txt"
rm -f "$O"/out_*.npz
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 | tee "$O/parity.txt"
grep -q "passed" "$O/parity.txt" && ! grep -q "failed" "$O/parity.txt"
