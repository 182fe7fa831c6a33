# round 6: parity.json on this build, the whole GPU tier, the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$DO_PARITY" ]; then python tests/tools/parity_json.py r06 > gpurun_out/r06_parity_json.log 2>&1; tail -2 gpurun_out/r06_parity_json.log | cut -c1-600; fi
cp profiles/parity.json gpurun_out/parity.json 2>/dev/null
timeout 3000 python -m pytest tests -m gpu -q --durations=15 2>&1 | tail -40 > gpurun_out/r06_gpu_tests.txt; tail -25 gpurun_out/r06_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; tail -1 gpurun_out/r06_bench.json | cut -c1-1500
