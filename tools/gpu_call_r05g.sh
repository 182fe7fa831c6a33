#!/bin/bash
# Round 5, A/B of the working tree against HEAD (generic; see the commit it was run for)
# model loads hoisted (all bit-identical): head (the 88 k build) against the working tree, twice; then the rearrange GPU tests
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab7_${name}_$w.json 2> gpurun_out/ab7_${name}_$w.err
  tail -1 gpurun_out/ab7_${name}_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', '$w', round(b['value']), b['config'].get('launch_ms'), 'status', b['config'].get('status_bits'), 'iters', round(b['config']['main']['mean_newton_iters'],3))" 2>&1 | tail -1
}
for w in rearrange_blocks ycb; do
  run head $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_head.so
  run tree $w X=1
  run head2 $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_head.so
  run tree2 $w X=1
done
timeout 900 python -m pytest tests/test_rearrange_kernel.py tests/test_rearrange_env.py tests/test_rearrange_ycb.py tests/test_rearrange_env_params.py tests/test_zz_round4_late_gpu.py tests/test_large_model.py -q -m gpu -p no:cacheprovider > gpurun_out/rb_tests_r05g.txt 2>&1
tail -8 gpurun_out/rb_tests_r05g.txt | cut -c1-300
python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_r05g.txt 2>&1
tail -24 gpurun_out/rearrange_stage_r05g.txt
timeout 300 python bench.py --workload full_perpendicular --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab7_full_perp.json 2> gpurun_out/ab7_full_perp.err
tail -1 gpurun_out/ab7_full_perp.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('full_perpendicular', round(b['value']), b['ms_per_step'], b['config'].get('mean_newton_iters'))"
RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_head.so timeout 300 python bench.py --workload full_perpendicular --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab7_full_perp_head.json 2> gpurun_out/ab7_full_perp_head.err
tail -1 gpurun_out/ab7_full_perp_head.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('full_perpendicular head', round(b['value']), b['ms_per_step'], b['config'].get('mean_newton_iters'))"
