#!/bin/bash
# round 5, GPU session 25: flag set B (LLGCBZ0 + LLNORMAGAM: SINPUT_ARD with the normalised growth rate) with the rows' constants fetched
# ahead (product) against V4_RECPF = 0 (norecpf): time, bits; then the IMPLSCH parity tests (IPHYS 0 builds included)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s25; mkdir -p "$O"
for v in "" norecpf "" norecpf "" norecpf; do
  echo "== IMPLSCH 131072 sp flag set B, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 131072 4 B 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee "$O/time.txt"
for v in "" norecpf; do
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/implsch_dump.py sp 8190 "$O/out_${v:-product}.npz" B > /dev/null 2>&1 || exit 1
done
python3 tools/implsch_dump.py --compare "$O/out_product.npz" "$O/out_norecpf.npz" | tee "$O/bits.txt"; rm -f "$O"/out_*.npz
timeout -k 10 1000 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 | tee "$O/parity.txt"
grep -q "passed" "$O/parity.txt" && ! grep -q "failed" "$O/parity.txt"
