#!/bin/bash
# Round 5: the large configuration (dactyl/full_perpendicular): lane-per-column Hessian assembly across four waves (block rows owned by waves), limit rows / row addresses by
# ballot in wave 0, direct pyramid-row reads in J'f.  head | hessoff (everything but the assembly) | tree, then the large-model GPU tests (incl. run-to-run bit identity) and the stage profile
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --workload full_perpendicular --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab9_$name.json 2> gpurun_out/ab9_$name.err
  tail -1 gpurun_out/ab9_$name.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', round(b['value']), round(b['ms_per_step'],2), 'iters', round(b['config']['mean_newton_iters'],3), 'status', b['config']['status_bits'])" 2>&1 | tail -1
}
run head RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_head.so
run hessoff RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_hessoff.so
run tree X=1
run head2 RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_head.so
run tree2 X=1
timeout 900 python -m pytest tests/test_large_model.py tests/test_full_cube_env.py tests/test_full_perpendicular.py -q -m gpu -p no:cacheprovider > gpurun_out/large_tests_r05i.txt 2>&1
tail -6 gpurun_out/large_tests_r05i.txt | cut -c1-300
python tools/large_stage_profile.py 512 2>&1 | grep -v amdgpu > gpurun_out/large_stage_r05i.txt
cat gpurun_out/large_stage_r05i.txt | cut -c1-300
