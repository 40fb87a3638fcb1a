#!/bin/bash
# one GPU step of a development session: run the command, keep its whole output in gpurun_out/<tag>.log, fail loudly on a runtime fault
# usage: bash tools/gpu_step.sh <tag> <command ...>
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT not set}" || exit 2; mkdir -p gpurun_out; tag=$1; shift
timeout -k 10 1000 "$@" > "gpurun_out/$tag.log" 2>&1; rc=$?
if grep -q "HSA_STATUS_ERROR\|Memory access fault" "gpurun_out/$tag.log"; then echo "$tag: GPU runtime fault"; grep -m3 "HSA_STATUS_ERROR\|Memory access fault\|Kernel Name" "gpurun_out/$tag.log" | cut -c1-220; exit 99; fi
echo "$tag rc=$rc"; tail -6 "gpurun_out/$tag.log"; exit $rc
