cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_suite.log 2>&1
rc=$?; echo suite rc=$rc; tail -6 gpurun_out/r04_gpu_suite.log
exit $rc
