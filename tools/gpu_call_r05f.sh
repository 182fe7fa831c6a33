#!/bin/bash
# Round 5, sixth GPU call: contact-Jacobian build through LDS dof lists + prefetched subtree sums (A/B against the same build with -DRB_ROWS_LEGACY and against 9dabb66),
# and the two env-step parity tests that failed in call 5, with their full output
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rearrange_env.py -q -m gpu -p no:cacheprovider -s -k "env_step_matches_oracle_gpu or observation_row_matches_oracle_gpu" > gpurun_out/env_step_tests_r05f.txt 2>&1
grep -E "^E |env.step vs|passed|failed" gpurun_out/env_step_tests_r05f.txt | cut -c1-700 | head -30
run() {
  local name=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab6_${name}_$w.json 2> gpurun_out/ab6_${name}_$w.err
  tail -1 gpurun_out/ab6_${name}_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', '$w', round(b['value']), b['config'].get('launch_ms'), 'status', b['config'].get('status_bits'), 'iters', round(b['config']['main']['mean_newton_iters'],3))" 2>&1 | tail -1
}
for w in rearrange_blocks ycb; do
  run prev $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_prev.so
  run rowslegacy $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_rowslegacy.so
  run full $w X=1
done
python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_r05f.txt 2>&1
tail -24 gpurun_out/rearrange_stage_r05f.txt
