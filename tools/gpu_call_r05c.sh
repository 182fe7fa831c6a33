#!/bin/bash
# Round 5, third GPU call: occupancy A/B of the small configuration (LDS now 8320 B = 7 granules: 18 workgroups per CU when compiled for 5 waves per SIMD / 96 VGPRs,
# 16 at the default 128 VGPRs), the medium configuration at 12 per CU (LDS 12768 B), and the -m gpu tier on this build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {
  local name=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab3_${name}_$w.json 2> gpurun_out/ab3_${name}_$w.err
  tail -1 gpurun_out/ab3_${name}_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', '$w', round(b['value']), b['config'].get('launch_ms'), 'status', b['config'].get('status_bits'), 'lds', b['config'].get('lds_bytes_per_workgroup'))" 2>&1 | tail -1
}
run base rearrange_blocks X=1
run w5 rearrange_blocks RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_w5.so
run base ycb X=1
run w5 ycb RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_w5.so
run base2 rearrange_blocks X=1
run w5b rearrange_blocks RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_w5.so
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests_r05c.txt 2>&1
tail -25 gpurun_out/gpu_tests_r05c.txt | cut -c1-400
timeout 300 python -m pytest tests/test_rearrange_env.py -q -m gpu -p no:cacheprovider -s -k "observation_row or env_step_matches" 2>&1 | grep -E "env.step vs|passed|failed" | cut -c1-1200
