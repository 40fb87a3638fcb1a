#!/bin/bash
# round 6, session 10: the refraction stencil: one launch with the hoisted form per lane; pinned to 3 / 4 waves per SIMD; 16-byte accesses
cd "${GRAFT_REPO_ROOT:?}" || exit 2
O=gpurun_out/r06s10; mkdir -p $O
run() { tag=$1; shift; timeout -k 10 500 python3 bench.py "$@" --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || echo "bench $tag failed";
  python3 -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('$tag', round(d['value']/1e6,2), 'M pt-steps/s', round(d['ms_per_step'],2), 'ms', {k:round(v['ms'],2) for k,v in d['kernels'].items()})"; }
A="--irefra 2 --steps 10 --warmup 2 --repeats 3"
run one $A
ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_genwpe3.so run wpe3 $A
ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_genwpe4.so run wpe4 $A
ECWAM_HIP_LIB=$PWD/ecwam_amd/lib/libecwam_hip_ctustrict.so run strict $A
ECWAM_HIP_GEN_TWO_LAUNCHES=1 run two $A
exit 0
