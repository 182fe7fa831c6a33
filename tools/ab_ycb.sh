# rearrange/ycb on the GPU: parity tests, env test, bench line, stage profile (gpurun helper)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_rearrange_ycb.py tests/test_rearrange_kernel.py tests/test_rearrange_env.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8
timeout 900 python bench.py --workload ycb --quick-reset --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ab_ycb.json 2> gpurun_out/ab_ycb.err
tail -3 gpurun_out/ab_ycb.err
tail -1 gpurun_out/ab_ycb.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('ycb', round(b['value']), b['config'].get('launch_ms'), b['config'].get('status_bits'), b['config']['main'])"
