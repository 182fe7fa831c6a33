#!/bin/bash
# Round 6: Newton factor reuse in rb_solve (large configuration), A/B of the in-tree library (before) against ab_libs/librgstep_reuse.so on configs[2], plus its identity test
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in base reuse; do
  [ $v = reuse ] && export RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_reuse.so
  timeout 900 python bench.py --workload full_perpendicular --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab_full_$v.json 2> gpurun_out/ab_full_$v.err
  tail -1 gpurun_out/ab_full_$v.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$v full cube', round(b['value']), round(b['ms_per_step'],2), b['config'].get('status_bits'), b['config'].get('mean_ncon'))" || tail -3 gpurun_out/ab_full_$v.err
  timeout 300 python tools/large_stage_profile.py 512 2>&1 | tail -13
done 2>&1 | tee gpurun_out/ab_full_reuse.txt
timeout 900 python -m pytest tests/test_large_model.py tests/test_full_cube_env.py tests/test_full_perpendicular.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4
