#!/bin/bash
# Round-end GPU cycle (rounds 5, 6): the -m gpu tier, the default bench line (headline + every secondary), kernel-trace stats of the headline command, HBM-traffic PMC
# passes of the headline and of the three rb_step_kernel workloads (FETCH_SIZE / WRITE_SIZE in separate passes), per-world kernel stats of the rearrange workloads,
# stage profiles, parity reports.  python tools/summarize_profile.py r05 afterwards stamps profiles/.
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests_$R.txt 2>&1
tail -4 gpurun_out/gpu_tests_$R.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_$R.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_$R.log 2>&1
for W in full_perpendicular rearrange_blocks ycb; do
  X="--quick-reset"; [ $W = full_perpendicular ] && X=""
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${W}_$R -o ${W}_$R --output-format csv -- python bench.py --workload $W $X --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_${W}_$R.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_${W}_$R -o fetch --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_${W}_$R.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_${W}_$R -o write --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_${W}_$R.log 2>&1
done
# the PMC figures of THIS build are stamped (profiles/hbm_traffic.json, on this box's copy) before the bench line runs, so that the line carries them (`roofline.traffic`)
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/bench_$R.json 2> /dev/null      # (a short line for the summariser to read the batch size from)
python tools/summarize_profile.py $R > /dev/null 2>&1
python tests/tools/parity_json.py $R > gpurun_out/parity_json_$R.log 2>&1; cp profiles/parity.json gpurun_out/parity.json
( time python bench.py --steps 20 --warmup 5 --verbose-secondary ) > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err      # (profiles/ keeps the secondary entries' text; the default line carries their numbers only)
tail -3 gpurun_out/bench_$R.err; tail -1 gpurun_out/bench_$R.json | cut -c1-200
python tests/tools/rearrange_parity_report.py 150 60 > gpurun_out/parity_rearrange_$R.txt 2>&1
python tools/stage_profile.py 8192 > gpurun_out/stage_$R.txt 2>&1
# (round 6) the parity block of the bench line, measured on this build; the Newton sub-stage profile of the analysis build (-DRG_FINE_PROF, ab_libs/librgstep_fine.so); the FETCH / WRITE calibration
[ -f ab_libs/librgstep_fine.so ] && RGSTEP_LIB=$GRAFT_REPO_ROOT/ab_libs/librgstep_fine.so python tools/stage_profile.py 8192 > gpurun_out/stage_fine_$R.txt 2>&1
bash tools/gpu_call_fetch_calib.sh > /dev/null 2>&1
timeout 500 python tools/wrapped_breakdown.py 8192 20 2>&1 | grep -v amdgpu.ids > gpurun_out/wrapped_breakdown_$R.txt      # the default make_env(): physics launches vs the wrapper stack's tensor kernels
RG_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29712 bench.py --workload ycb --quick-reset --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_ycb_rccl1_$R.json 2> gpurun_out/bench_ycb_rccl1_$R.err
python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_$R.txt 2>&1
python tools/rearrange_stage_profile.py 4096 ycb > gpurun_out/ycb_stage_$R.txt 2>&1
python tools/large_stage_profile.py 512 > gpurun_out/large_stage_$R.txt 2>&1
bash tools/prof_pmc_rearrange.sh > gpurun_out/pmc_rearrange_$R.txt 2>&1
# steady state with episode ends, the recipe on the device (ra_recipe_kernel): 2 M env-steps of blocks, 1.2 M of ycb; heterogeneous ycb batches
python tools/soak_rearrange.py 4096 500 blocks 30 device > gpurun_out/soak_rearrange_$R.txt 2>&1
python tools/soak_rearrange.py 4096 300 ycb 30 device >> gpurun_out/soak_rearrange_$R.txt 2>&1
python tools/bench_ycb_sets.py 4096 8 multi singles > gpurun_out/ycb_object_sets_$R.txt 2>&1
python tools/bench_ycb_sets.py 4096 8 streams 2>&1 | grep -v "object sets (0,)" >> gpurun_out/ycb_object_sets_$R.txt
du -sh gpurun_out
