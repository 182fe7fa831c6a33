# SQ counters of the large-model stepper (rb_step_kernel) on the configs[2] bench command: where the wave cycles go
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d gpurun_out/pmcL1 -o pmcL1 --output-format csv -- python bench.py --workload full_perpendicular --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcL1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d gpurun_out/pmcL2 -o pmcL2 --output-format csv -- python bench.py --workload full_perpendicular --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmcL2.log 2>&1
python - <<'PY'
import csv, glob
for tag in ("pmcL1", "pmcL2"):
    f = glob.glob("gpurun_out/%s/*counter_collection.csv" % tag)
    if not f:
        print(tag, "no counters"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "rb_step_kernel" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50e6:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(tag, {k: "%.3g" % (sum(v) / len(v)) for k, v in sorted(acc.items())}, "launches", {k: len(v) for k, v in acc.items()}.get("SQ_WAVES", len(next(iter(acc.values()), []))))
PY
