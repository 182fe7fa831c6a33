# rb_step_kernel, large configuration: parity tests + the configs[2] bench line (gpurun helper)
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_large_model.py tests/test_full_cube_env.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
timeout 900 python bench.py --workload full_perpendicular --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab_full.json 2> gpurun_out/ab_full.err
tail -1 gpurun_out/ab_full.json | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('full', round(b['value']), b['ms_per_step'], b['config'].get('status_bits'), b['config'].get('mean_ncon'))"
timeout 300 python tools/large_stage_profile.py 512 2>&1 | tail -14
