#!/bin/bash
# The measured part of the round profile that is stamped with the kernel build (bench line, kernel stats of the headline, HBM-traffic PMC passes of the headline and
# of the rb_step_kernel workloads) -- for a late kernel change that does not move the numbers; the full run is tools/profile_round.sh.
R=${1:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
tail -1 gpurun_out/bench_$R.json | cut -c1-200
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/bench_prof_$R.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_$R.log 2>&1
for W in full_perpendicular rearrange_blocks ycb; do
  X="--quick-reset"; [ $W = full_perpendicular ] && X=""
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_${W}_$R -o fetch --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_${W}_$R.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_${W}_$R -o write --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_${W}_$R.log 2>&1
done
python tests/tools/rearrange_parity_report.py 150 60 > gpurun_out/parity_rearrange_$R.txt 2>&1
tail -14 gpurun_out/parity_rearrange_$R.txt
python tools/soak_rearrange.py 4096 500 blocks 20 > gpurun_out/soak_rearrange_$R.txt 2>&1
tail -2 gpurun_out/soak_rearrange_$R.txt
