#!/bin/bash
# round 5, GPU session 24: the module's scalar constants in the lanes of two vector registers (V4_TBS, product) against scalar loads (notbs):
# time sp and dp, bits
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s24; mkdir -p "$O"
for prec in sp dp; do
for v in "" notbs "" notbs "" notbs; do
  echo "== IMPLSCH 131072 $prec, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py $prec 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done
done | tee "$O/time.txt"
for prec in sp dp; do
for v in "" notbs; do
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/implsch_dump.py $prec 8190 "$O/out_${prec}_${v:-product}.npz" > /dev/null 2>&1 || exit 1
done
echo "== outputs $prec: product against notbs"; python3 tools/implsch_dump.py --compare "$O/out_${prec}_product.npz" "$O/out_${prec}_notbs.npz"
done | tee "$O/bits.txt"; rm -f "$O"/out_*.npz
exit 0
