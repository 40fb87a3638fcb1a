#!/bin/bash
# round 5, GPU session 2: (a) the parity suite on the product library (double precision RARE builds at -O2 on generation 4), (b) the whole
# parity suite on the "split" build of the library (every k_implsch4 as PART 1 | PART 2), (c) its time and kernel trace beside the product's,
# (d) the double precision RARE builds against k_implsch2 in time, (e) last: the -O3 split of the double precision RARE builds (may fault)
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s2; mkdir -p "$O"
fault() { grep -q "Memory access fault\|HSA_STATUS_ERROR" "$1" && { echo "GPU runtime fault in $1"; grep -m3 "Memory access fault\|HSA_STATUS_ERROR" "$1"; return 0; }; return 1; }
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats_product.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > "$O/pytest_product.log" 2>&1; rc=$?; tail -3 "$O/pytest_product.log"
[ $rc -eq 124 ] && exit 124; fault "$O/pytest_product.log" && exit 99
export ECWAM_TEST_STATS_LOG="$PWD/$O/stats_split.jsonl"; rm -f "$ECWAM_TEST_STATS_LOG"
ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip_split.so" timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py -x -q -m gpu > "$O/pytest_split.log" 2>&1; rc=$?; tail -3 "$O/pytest_split.log"
[ $rc -eq 124 ] && exit 124; fault "$O/pytest_split.log" && exit 99
unset ECWAM_TEST_STATS_LOG
for v in "" split "" split; do
  echo "== IMPLSCH O320 sp, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py sp 421080 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee "$O/time_split_vs_product_sp.txt"
for v in "" split; do
  echo "== IMPLSCH 131072 points dp, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py dp 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done | tee "$O/time_split_vs_product_dp.txt"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip_split.so" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_split" -- python3 tools/prof_implsch.py sp 421080 4 > "$O/prof_split.log" 2>&1
find "$O/prof_split" -name "*kernel_stats.csv" -exec head -8 {} \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_product" -- python3 tools/prof_implsch.py sp 421080 4 > "$O/prof_product.log" 2>&1
find "$O/prof_product" -name "*kernel_stats.csv" -exec head -8 {} \;
for f in RW RI R2 RU RB RJ; do
  timeout -k 10 300 python3 tests/diag/implsch_gens.py 131072 dp 36 $f 2>&1 | grep -v amdgpu.ids | tail -4
done | tee "$O/rare_dp_generations.txt"
echo "== variant rdps (the -O3 two-kernel split of the double precision RARE builds)"
ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip_rdps.so" timeout -k 10 180 python tests/diag/rare_dp_probe.py 24 512 dp > "$O/probe_rdps.log" 2>&1; echo "rc=$?"; tail -12 "$O/probe_rdps.log"
exit 0
