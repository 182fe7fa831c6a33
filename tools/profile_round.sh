#!/bin/bash
# Round profile: bench line (with cpu_baseline), rocprofv3 kernel-trace stats of the same command, the HBM-traffic PMC
# passes (FETCH_SIZE and WRITE_SIZE in separate passes, per MI355X_MICROARCH.md), SQ counters and the per-stage cycle profile.
#   gpurun -- 'bash tools/profile_round.sh r02'   then   python tools/summarize_profile.py r02
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
tail -1 gpurun_out/bench_$R.json | cut -c1-600
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench100_$R.json 2> gpurun_out/bench100_$R.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipelined-reset > gpurun_out/bench_pipe_$R.json 2> gpurun_out/bench_pipe_$R.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof_$R.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_write_$R.log 2>&1
python tools/stage_profile.py 8192 > gpurun_out/stage_$R.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d gpurun_out/pmc1 -o pmc1 --output-format csv -- python tools/stage_profile.py 8192 > gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/pmc2 -o pmc2 --output-format csv -- python tools/stage_profile.py 8192 > gpurun_out/pmc2.log 2>&1
python tests/tools/parity_report.py 4 1000 > gpurun_out/parity_$R.txt 2>&1
ls gpurun_out/prof_$R gpurun_out/pmc_fetch_$R gpurun_out/pmc_write_$R | head -20
