# A/B of rb_step_kernel builds / configurations on the rearrange workload (gpurun helper)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_rearrange_kernel.py tests/test_rearrange_env.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
for v in ${AB_VARIANTS:-small}; do
  if [ $v = large ]; then export RB_CONFIG=large; else unset RB_CONFIG; fi
  if [ -f ab_libs/librgstep_$v.so ]; then export RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_$v.so; else unset RGSTEP_LIB; fi
  timeout 600 python bench.py --workload rearrange_blocks --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_rearr_$v.json 2> gpurun_out/ab_rearr_$v.err
  tail -1 gpurun_out/ab_rearr_$v.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$v', round(b['value']), b['config'].get('launch_ms'), b['config'].get('status_bits'))"
done
unset RB_CONFIG RGSTEP_LIB
timeout 300 python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_small.txt 2>&1
cat gpurun_out/rearrange_stage_small.txt | head -30
