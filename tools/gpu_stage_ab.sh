# per-stage cycle profile of alternative builds of librgstep (RGSTEP_LIB): bash tools/gpu_stage_ab.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  RGSTEP_LIB=$GRAFT_REPO_ROOT/$lib python tools/stage_profile.py 8192 > gpurun_out/stage_ab_$name.txt 2>&1
  echo "$name $(grep 'kernel ms' gpurun_out/stage_ab_$name.txt | cut -c1-40) $(grep total gpurun_out/stage_ab_$name.txt)"
done
