#!/bin/bash
# diagnostics build of the library (phase early-exits, ECWAM_HIP_DEBUG_SKIP honoured): ecwam_amd/lib/libecwam_hip_diag.so
cd "$(dirname "$0")/../ecwam_amd/lib" || exit 1
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -DECWAM_HIP_DIAGNOSTICS"
hipcc $F -c ../csrc/capi.hip -o /tmp/capi_diag.o &
hipcc $F -fno-hip-fp32-correctly-rounded-divide-sqrt -c ../csrc/implsch4.hip -o /tmp/implsch4_diag.o &
hipcc $F -c ../csrc/propag.hip -o /tmp/propag_diag.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o libecwam_hip_diag.so /tmp/capi_diag.o /tmp/propag_diag.o implsch.o /tmp/implsch4_diag.o implsch4x.o outbs.o
