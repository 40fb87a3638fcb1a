#!/bin/bash
# round 6, session 6: the one-kernel step with the hoisted weights -- gather depth, wave priority, XCD grouping
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s6; mkdir -p $O
B="python bench.py --no-cpu-baseline --steps 20 --warmup 3 --fused on"
for i in 1 2; do
  timeout -k 10 300 $B > $O/b_base_$i.json 2> $O/b_base_$i.err; echo "base rc=$?"
  for v in advd2 advd4 advd5 advprio; do
    ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_$v.so timeout -k 10 300 $B > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err; echo "$v rc=$?"
  done
  for g in 4 16 32 64 256; do
    timeout -k 10 300 $B --fused-flags $((g*256)) > $O/b_xcd${g}_$i.json 2> $O/b_xcd${g}_$i.err; echo "xcd$g rc=$?"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06s6/b_*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d["ms_per_step"],3), d["finite"], d["swh_norm_rank0"]["avg"])
    except Exception as e: print(f, "ERR", e)
PY
