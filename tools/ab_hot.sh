# A/B of librgstep builds on the headline workload (gpurun helper): AB_VARIANTS="base w4 ..." -> ab_libs/librgstep_<v>.so ("base" = the in-tree build)
cd $GRAFT_REPO_ROOT
for v in ${AB_VARIANTS:-base}; do
  if [ -f ab_libs/librgstep_$v.so ]; then export RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_$v.so; else unset RGSTEP_LIB; fi
  for rep in 1 2; do
    timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline ${AB_EXTRA} > gpurun_out/ab_hot_$v.json 2> gpurun_out/ab_hot_$v.err
    tail -1 gpurun_out/ab_hot_$v.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$v', round(b['value']), round(b['ms_per_step'],3), b['roofline']['frac'], b['config'].get('status_bits'))"
  done
done
