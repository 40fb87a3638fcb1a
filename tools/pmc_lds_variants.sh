#!/bin/bash
# diagnostics: LDS counters of IMPLSCH for several builds of the library on one box.  usage: bash tools/pmc_lds_variants.sh "" natlayout ...
for v in "$@"; do
  export ECWAM_HIP_LIB="${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}"/ecwam_amd/lib/libecwam_hip${v:+_$v}.so
  echo "== ${v:-product}"
  PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" TAG=lds_${v:-product} N=${N:-131072} PREC=${PREC:-sp} GEN=4 bash tools/pmc_run.sh
done
