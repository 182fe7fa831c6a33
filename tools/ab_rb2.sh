# rb_step_kernel A/B of the builds named in AB_VARIANTS (ab_libs/librgstep_<v>.so) on rearrange/blocks and ycb, quick reset protocol
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in ${AB_VARIANTS:-base}; do
  export RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_$v.so
  for w in rearrange_blocks ycb; do
    timeout 900 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_${w}_$v.json 2> gpurun_out/ab_${w}_$v.err
    tail -1 gpurun_out/ab_${w}_$v.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$v', '$w', round(b['value']), b['config'].get('launch_ms'), b['config'].get('status_bits'))" || tail -3 gpurun_out/ab_${w}_$v.err
  done
done
