#!/bin/bash
# Round profile: bench line (with cpu_baseline), rocprofv3 kernel-trace stats of the same command, the HBM-traffic PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate passes, per MI355X_MICROARCH.md), SQ counters and the per-stage cycle profile;
# A/B of the dispatch modes; configs[2] (large-model kernel) bench + stage profile; parity reports.
#   gpurun -- 'bash tools/profile_round.sh r03'   then   python tools/summarize_profile.py r03
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
tail -1 gpurun_out/bench_$R.json | cut -c1-400
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/bench100_$R.json 2> gpurun_out/bench100_$R.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --pipelined-reset > gpurun_out/bench_pipe_$R.json 2> gpurun_out/bench_pipe_$R.err
RG_SUBSTEP_ITEMS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/bench_classic_$R.json 2> gpurun_out/bench_classic_$R.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sort-dispatch 0 > gpurun_out/bench_nosort_$R.json 2> gpurun_out/bench_nosort_$R.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_$R.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_$R.log 2>&1
# the same two PMC passes without the per-pair collision cache (launch flag bit 2): what is left is state rows + register spills
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_nocache_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --launch-flags 4 > gpurun_out/pmc_fetch_nocache_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_nocache_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --launch-flags 4 > gpurun_out/pmc_write_nocache_$R.log 2>&1
python tools/stage_profile.py 8192 > gpurun_out/stage_$R.txt 2>&1
python tests/tools/parity_report.py 4 1000 > gpurun_out/parity_$R.txt 2>&1
python bench.py --workload full_perpendicular --steps 5 --warmup 1 --no-secondary > gpurun_out/bench_full_$R.json 2> gpurun_out/bench_full_$R.err
tail -1 gpurun_out/bench_full_$R.json | cut -c1-300
python tools/large_stage_profile.py 512 > gpurun_out/large_stage_$R.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_full_$R -o full_$R --output-format csv -- python bench.py --workload full_perpendicular --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_full_prof_$R.log 2>&1
python tools/bench_wrapped.py > gpurun_out/wrapped_$R.txt 2>&1
bash tools/prof_pmc_large.sh > gpurun_out/pmc_large_$R.txt 2>&1
# configs[3] rearrange/blocks: bench line through env.step, kernel trace of the same command, stage cycle profile of both worlds
python bench.py --workload rearrange_blocks --steps 10 --warmup 3 > gpurun_out/bench_rearrange_$R.json 2> gpurun_out/bench_rearrange_$R.err
tail -1 gpurun_out/bench_rearrange_$R.json | cut -c1-300
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rearrange_$R -o rearrange_$R --output-format csv -- python bench.py --workload rearrange_blocks --quick-reset --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_rearrange_prof_$R.log 2>&1
python tools/rearrange_stage_profile.py 4096 > gpurun_out/rearrange_stage_$R.txt 2>&1
bash tools/prof_pmc_rearrange.sh > gpurun_out/pmc_rearrange_$R.txt 2>&1
# configs[4] rearrange/ycb (fixed object set), one GPU's share of the 8-GPU line
python bench.py --workload ycb --steps 10 --warmup 3 > gpurun_out/bench_ycb_$R.json 2> gpurun_out/bench_ycb_$R.err
tail -1 gpurun_out/bench_ycb_$R.json | cut -c1-300
python tools/rearrange_stage_profile.py 4096 ycb > gpurun_out/ycb_stage_$R.txt 2>&1
# steady state with episode ends (pipelined resets)
python tools/soak_rearrange.py 4096 500 blocks 20 > gpurun_out/soak_rearrange_$R.txt 2>&1
python tools/soak_rearrange.py 4096 300 ycb 10 >> gpurun_out/soak_rearrange_$R.txt 2>&1
ls gpurun_out/prof_$R gpurun_out/pmc_fetch_$R gpurun_out/pmc_write_$R gpurun_out/prof_rearrange_$R | head -30
# HBM traffic of rb_step_kernel on configs[2] / [3] / [4] (separate FETCH_SIZE / WRITE_SIZE passes, as for the headline): the dominant launch of each
for W in full_perpendicular rearrange_blocks ycb; do
  X="--quick-reset"; [ $W = full_perpendicular ] && X=""
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_${W}_$R -o fetch --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_${W}_$R.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_${W}_$R -o write --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_${W}_$R.log 2>&1
done
# the N > 1 code path of the ycb workload over RCCL with one rank (process group, barriers, all-gather of the packed rows, max-reduce) on this 1-GPU box
RG_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29712 bench.py --workload ycb --quick-reset --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_ycb_rccl1_$R.json 2> gpurun_out/bench_ycb_rccl1_$R.err
tail -1 gpurun_out/bench_ycb_rccl1_$R.json | cut -c1-200

