# rb_step_kernel quick A/B on the three rb-based workloads (gpurun helper)
cd $GRAFT_REPO_ROOT
for w in rearrange_blocks ycb; do
  timeout 900 python bench.py --workload $w --quick-reset --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$w.json 2> gpurun_out/ab_$w.err
  tail -1 gpurun_out/ab_$w.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$w', round(b['value']), b['config'].get('launch_ms'), b['config'].get('status_bits'))"
done
if [ -n "$AB_FULL" ]; then
timeout 900 python bench.py --workload full_perpendicular --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab_full.json 2> gpurun_out/ab_full.err
tail -1 gpurun_out/ab_full.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('full', round(b['value']), b['ms_per_step'], b['config'].get('status_bits'))"
fi
