# round 6, call a: register Newton factorisation (creg) vs base: stage profile, bench A/B, parity tier on the new build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/gpu_stage_ab.sh ab_libs/librgstep_base.so ab_libs/librgstep_creg.so 2>&1 | tee gpurun_out/r06a_stage.txt
AB_VARIANTS="base creg base creg" bash tools/ab_hot.sh 2>&1 | tee gpurun_out/r06a_ab.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_kernel_emul.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06a_tests.txt
