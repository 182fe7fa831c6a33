# round 6: A/B of the builds named in AB_VARIANTS + fine stage profile of ab_libs/librgstep_fine.so + GPU parity tier on the in-tree build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_stage_ab.sh ab_libs/librgstep_fine.so > /dev/null 2>&1; tail -32 gpurun_out/stage_ab_librgstep_fine.txt
bash tools/ab_hot.sh 2>&1 | tee gpurun_out/r06_ab.txt
if [ -n "$RUN_TESTS" ]; then timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_kernel_emul.py tests/test_env_parity.py tests/test_reach.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r06_tests.txt; fi
