#!/bin/bash
# Round profile: bench line (with cpu_baseline), rocprofv3 kernel-trace stats of the same command, and the
# HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, per MI355X_MICROARCH.md).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
tail -1 gpurun_out/bench_$R.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof_$R.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_write_$R.log 2>&1
ls gpurun_out/prof_$R gpurun_out/pmc_fetch_$R gpurun_out/pmc_write_$R
