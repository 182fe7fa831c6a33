#!/bin/bash
# Round 5, fifth GPU call: rb_step_kernel Newton changes (evaluation order + incremental advance) and the flattened Hessian assembly of the one-wave configurations:
# prev (9dabb66) | newton (-DRB_HESS_SERIAL: Newton changes only) | full (default build); rearrange parity tests + stage profile on the full build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab5_${name}_$w.json 2> gpurun_out/ab5_${name}_$w.err
  tail -1 gpurun_out/ab5_${name}_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', '$w', round(b['value']), b['config'].get('launch_ms'), 'status', b['config'].get('status_bits'), 'iters', round(b['config']['main']['mean_newton_iters'],3))" 2>&1 | tail -1
}
for w in rearrange_blocks ycb; do
  run prev $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_prev.so
  run newton $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_newton.so
  run full $w X=1
done
timeout 900 python -m pytest tests/test_rearrange_kernel.py tests/test_rearrange_env.py tests/test_rearrange_ycb.py tests/test_rearrange_env_params.py tests/test_zz_round4_late_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_r05e.txt 2>&1
tail -24 gpurun_out/rearrange_stage_r05e.txt
python tests/tools/rearrange_parity_report.py 60 20 > gpurun_out/parity_rearrange_r05e.txt 2>&1
head -12 gpurun_out/parity_rearrange_r05e.txt | cut -c1-200
