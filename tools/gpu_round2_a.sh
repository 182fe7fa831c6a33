cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -120) > gpurun_out/r02_gputest1.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sort-dispatch 0 > gpurun_out/r02_bench_nosort.json 2> gpurun_out/r02_bench_nosort.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sort-dispatch 1 > gpurun_out/r02_bench_sort.json 2> gpurun_out/r02_bench_sort.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sort-dispatch 1 > gpurun_out/r02_bench_sort100.json 2> gpurun_out/r02_bench_sort100.err
tail -3 gpurun_out/r02_gputest1.log; cat gpurun_out/r02_bench_nosort.json | cut -c1-400; cat gpurun_out/r02_bench_sort.json | cut -c1-400
