#!/bin/bash
# diagnostics: IMPLSCH time with phases ablated at full residency and at one block per CU (latency-bound limit)
for pad in 0 45000; do
for mask in ${MASKS:-0 1 2 4 8 127}; do
  echo -n "pad $pad mask $mask: "
  ECWAM_HIP_IMPLSCH_PADLDS=$pad ECWAM_HIP_DEBUG_SKIP=$mask python3 tools/prof_implsch.py sp 131072 2>&1 | tail -1
done; done
