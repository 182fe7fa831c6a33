cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/stage_profile.py 8192 > gpurun_out/stage_r02a.txt 2>&1
cat gpurun_out/stage_r02a.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02a -o r02a --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof_r02a.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_r02a/*kernel_stats.csv')[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < 25: print("%-70s calls %5s avg %9.1f us tot %8.2f ms %s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
