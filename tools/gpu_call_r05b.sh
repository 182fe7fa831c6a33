#!/bin/bash
# Round 5, second GPU call: A/B of rb_step_kernel placements / capacities on rearrange blocks + ycb (env-var variants of one build + two builds under ab_libs/).
# (The RB_LDS_PLACE variants need a library built with -DRB_LDS_ARENA: at the time of the call that was the default build; results: profiles/r05_ab_rb_placement.txt)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() {   # name, workload, extra env...
  local name=$1 w=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_${name}_$w.json 2> gpurun_out/ab_${name}_$w.err
  tail -1 gpurun_out/ab_${name}_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$name', '$w', round(b['value']), b['config'].get('launch_ms'), 'status', b['config'].get('status_bits'), 'lds', b['config'].get('lds_bytes_per_workgroup'))" 2>&1 | tail -1
}
for w in rearrange_blocks ycb; do
  run base $w X=1
  run cw24 $w RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_cw24.so
  run compact $w RB_SCRATCH_MAXCON=96 RB_SCRATCH_MAXROW=640
  run frames $w RB_LDS_PLACE=frames
  run dyn $w RB_LDS_PLACE=dyn
  run kin $w RB_LDS_PLACE=kin
  run kin_compact $w RB_LDS_PLACE=kin RB_SCRATCH_MAXCON=96 RB_SCRATCH_MAXROW=640
  run frames_compact $w RB_LDS_PLACE=frames RB_SCRATCH_MAXCON=96 RB_SCRATCH_MAXROW=640
  run kin_w2 $w RB_LDS_PLACE=kin RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_w2.so
  run kin_compact_w2 $w RB_LDS_PLACE=kin RB_SCRATCH_MAXCON=96 RB_SCRATCH_MAXROW=640 RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_w2.so
done
# correctness of the placements on the GPU: the rearrange parity tests under the kin placement, and the failing test's detail from call 1
RB_LDS_PLACE=kin timeout 600 python -m pytest tests/test_rearrange_kernel.py tests/test_rearrange_env.py tests/test_rearrange_ycb.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5
timeout 600 python -m pytest tests/test_rearrange_env.py -q -m gpu -p no:cacheprovider -s -k "ycb_env_step_observation_row" > gpurun_out/ycb_obs_row.txt 2>&1
grep -E "Error|assert|env.step vs" gpurun_out/ycb_obs_row.txt | head -12 | cut -c1-600
RB_LDS_PLACE=kin python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_kin.txt 2>&1
tail -24 gpurun_out/rearrange_stage_kin.txt
