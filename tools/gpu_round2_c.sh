cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
./tools/lds_occupancy_probe > gpurun_out/r02_lds_probe.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_A.json 2> gpurun_out/r02_bench_A.err
RGSTEP_LIB=$GRAFT_REPO_ROOT/robogym_amd/csrc/librgstep_lb3.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_B.json 2> gpurun_out/r02_bench_B.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02_bench_A100.json 2> gpurun_out/r02_bench_A100.err
(timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 | tail -150) > gpurun_out/r02_gputest3.log 2>&1
cat gpurun_out/r02_lds_probe.txt
for f in A B A100; do python -c "
import json;r=json.loads(open('gpurun_out/r02_bench_$f.json').read().strip().split('\n')[-1]);print('$f',round(r['value']),r['ms_per_step'],r['roofline']['kernel_ms'],r['config']['status_bits'],r['config']['status_bits_before_timed_region'])"; done
tail -4 gpurun_out/r02_gputest3.log
