#!/bin/bash
# Round 5, fourth GPU call: the -m gpu tier on the build with per-env parameter rows and the reference's reset-recipe steps; cost of the rows (model values) on
# rearrange blocks / ycb; the free-running divergence statement
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests_r05d.txt 2>&1
tail -30 gpurun_out/gpu_tests_r05d.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_rearrange_env_params.py tests/test_reach.py -q -m gpu -p no:cacheprovider -s -k "free_running or per_env_parameters" 2>&1 | grep -v amdgpu.ids > gpurun_out/parity_free_running_r05.txt
grep -E "free-running|first step|env [0-9]|passed|failed" gpurun_out/parity_free_running_r05.txt | cut -c1-300
run() {
  local name=$1 w=$2; shift 2
  timeout 300 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/ab4_${name}_$w.json 2> gpurun_out/ab4_${name}_$w.err
  tail -1 gpurun_out/ab4_${name}_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', '$w', round(b['value']), b['config'].get('launch_ms'), 'status', b['config'].get('status_bits'), 'rows', b['config'].get('per_env_parameters'))" 2>&1 | tail -1
}
for w in rearrange_blocks ycb; do
  run base $w
  run rows $w --per-env-params
  run base2 $w
done
