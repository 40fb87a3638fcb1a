#!/bin/bash
# round 5, GPU session 21: the sweep's coefficient record fetched one interaction ahead (variant recpf) against the product: time, bits, counters
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2
O=gpurun_out/r05s21; mkdir -p "$O"
for prec in sp dp; do
for v in "" recpf "" recpf; do
  echo "== IMPLSCH 131072 $prec, library ${v:-product}"
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/prof_implsch.py $prec 131072 4 2>&1 | grep "implsch ms" | sort -n -k3 | head -2
done
done | tee "$O/time.txt"
for v in "" recpf; do
  ECWAM_HIP_LIB="$PWD/ecwam_amd/lib/libecwam_hip${v:+_$v}.so" timeout -k 10 200 python3 tools/implsch_dump.py sp 8190 "$O/out_sp_${v:-product}.npz" > /dev/null 2>&1 || exit 1
done
python3 tools/implsch_dump.py --compare "$O/out_sp_product.npz" "$O/out_sp_recpf.npz" | tee "$O/bits.txt"; rm -f "$O"/out_*.npz
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
for v in "" recpf; do
  export ECWAM_HIP_LIB="$R/ecwam_amd/lib/libecwam_hip${v:+_$v}.so"
  timeout -k 10 300 rocprofv3 --kernel-include-regex "k_implsch4<" --pmc SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES \
    --kernel-trace --output-format csv -d "$R/$O/pmc_${v:-product}" -- python3 "$R/tools/prof_implsch.py" sp 131072 > "$R/$O/pmc_${v:-product}.log" 2>&1 || { echo "pmc failed"; exit 1; }
done
unset ECWAM_HIP_LIB
cd "$R"
python3 - <<'PY'
import csv, glob, collections
for v in ("product", "recpf"):
    f = glob.glob(f"gpurun_out/r05s21/pmc_{v}/*/*counter_collection.csv")
    agg = collections.defaultdict(float); n = set()
    for row in csv.DictReader(open(f[0])):
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); n.add(row["Dispatch_Id"])
    print(v, "launches", len(n), {k: round(v_ / len(n) / 131072, 1) for k, v_ in agg.items()})
PY
