#!/bin/bash
# round 5, GPU session 26: the coefficient records fetched ahead (V4_RECPF; single precision): loads of the sweep's record at the top of the
# interaction (recpf) or in front of the row update (recpf1) against the product: time, bits
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s26; mkdir -p "$O"
for v in "" noredn "" noredn "" noredn; do
  echo "== IMPLSCH 131072 sp, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee "$O/time.txt"
for v in "" noredn; do
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/implsch_dump.py sp 8190 "$O/out_sp_${v:-product}.npz" > /dev/null 2>&1 || exit 1
done
for v in noredn; do echo "== outputs: product against $v"; python3 tools/implsch_dump.py --compare "$O/out_sp_product.npz" "$O/out_sp_$v.npz"; done | tee "$O/bits.txt"; rm -f "$O"/out_*.npz
exit 0
