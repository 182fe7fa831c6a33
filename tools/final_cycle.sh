#!/bin/bash
# last GPU call of the round: stamped PMC passes + bench line on the final build, the GPU tests, the rearrange parity report
R=r04
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -4 > gpurun_out/final_gpu_tests.txt
cat gpurun_out/final_gpu_tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
tail -1 gpurun_out/bench_$R.json | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$R -o fetch --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_$R.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_$R -o write --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_$R.log 2>&1
for W in full_perpendicular rearrange_blocks ycb; do
  X="--quick-reset"; [ $W = full_perpendicular ] && X=""
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_${W}_$R -o fetch --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_fetch_${W}_$R.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write_${W}_$R -o write --output-format csv -- python bench.py --workload $W $X --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/pmc_write_${W}_$R.log 2>&1
done
python tests/tools/rearrange_parity_report.py 150 60 > gpurun_out/parity_rearrange_$R.txt 2>&1
tail -13 gpurun_out/parity_rearrange_$R.txt | head -9
