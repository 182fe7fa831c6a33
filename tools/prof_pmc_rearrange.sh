# SQ counters of rb_step_kernel (small configuration) on the rearrange/blocks bench command: where the wave cycles go
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
W=${1:-rearrange_blocks}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d gpurun_out/pmcR1 -o pmcR1 --output-format csv -- python bench.py --workload $W --quick-reset --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcR1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/pmcR2 -o pmcR2 --output-format csv -- python bench.py --workload $W --quick-reset --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcR2.log 2>&1
python - <<'PY'
import csv, glob
for tag in ("pmcR1", "pmcR2"):
    f = glob.glob("gpurun_out/%s/*counter_collection.csv" % tag)
    if not f:
        print(tag, "no counters"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if "rb_step_kernel" in r["Kernel_Name"] and d > 20e6:      # the main world's full launches (round 5: ~26 ms; the solver world's are ~13 ms)
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(tag, {k: "%.3g" % (sum(v) / len(v)) for k, v in sorted(acc.items())}, "launches", len(next(iter(acc.values()), [])))
PY
