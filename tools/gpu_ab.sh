# A/B bench of alternative builds of librgstep (RGSTEP_LIB): usage: bash tools/gpu_ab.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  RGSTEP_LIB=$GRAFT_REPO_ROOT/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python -c "
import json;r=json.loads(open('gpurun_out/ab_$name.json').read().strip().split('\n')[-1]);print('$name',round(r['value']),round(r['ms_per_step'],3),round(r['roofline']['kernel_ms'],3),r['config']['status_bits'])" || tail -3 gpurun_out/ab_$name.err
done
