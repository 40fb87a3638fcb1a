#!/bin/bash
# diagnostics: cumulative time of k_implsch3 up to each phase boundary (ECWAM_HIP_DEBUG_SKIP = 101..107 returns early)
for m in 100 101 102 103 104 105 106 107 0; do
  echo -n "skip $m: "; ECWAM_HIP_DEBUG_SKIP=$m python tools/time_implsch_v3.py 131072 2>/dev/null | grep k_implsch3
done
