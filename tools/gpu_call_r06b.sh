# round 6, call b: SQ counters of the headline launch (bench.py's own launches), three passes.  usage: gpu_call_r06b.sh <lib.so> <tag>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
LIB=${1:-robogym_amd/csrc/librgstep.so}
TAG=${2:-cur}
export RGSTEP_LIB=$GRAFT_REPO_ROOT/$LIB
CMD="python bench.py --steps 10 --warmup 3 --no-secondary --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d gpurun_out/pmc_${TAG}_1 -o p --output-format csv -- $CMD > gpurun_out/pmc_${TAG}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES -d gpurun_out/pmc_${TAG}_2 -o p --output-format csv -- $CMD > gpurun_out/pmc_${TAG}_2.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS -d gpurun_out/pmc_${TAG}_3 -o p --output-format csv -- $CMD > gpurun_out/pmc_${TAG}_3.log 2>&1
python - <<PY
import csv, glob, collections
for k in (1, 2, 3):
    fs = glob.glob("gpurun_out/pmc_${TAG}_%d/**/*counter_collection.csv" % k, recursive=True)
    if not fs: print("pass", k, "no file", glob.glob("gpurun_out/pmc_${TAG}_%d/**/*" % k, recursive=True)[:5]); continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if "rg_step_items_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("pass", k, {a: "%.4g" % (acc[a] / max(n[a], 1)) for a in sorted(acc)}, "launches", max(n.values()) if n else 0)
PY
tail -2 gpurun_out/pmc_${TAG}_3.log | cut -c1-300
