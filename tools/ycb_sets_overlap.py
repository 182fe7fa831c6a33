"""How much the per-group launch chains of GroupedYcbRearrangeEnv overlap on the GPU: from a rocprofv3 kernel trace of tools/bench_ycb_sets.py, the rb_step_kernel
launches' total busy time against the length of the union of their intervals (1.0 = serialised, K = K chains fully concurrent).

    rocprofv3 --kernel-trace -d gpurun_out/prof_sets -o sets --output-format csv -- python tools/bench_ycb_sets.py 4096 6
    python tools/ycb_sets_overlap.py gpurun_out/prof_sets"""
import csv
import glob
import os
import sys

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:6]) for r in csv.DictReader(open(f)) if "rb_step_kernel" in r["Kernel_Name"])
# split the trace into the runs of the script (gaps > 1 s between launches: model creation of the next configuration)
runs, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - cur[-1][1] > 1_000_000_000:
        runs.append(cur); cur = []
    cur.append(r)
runs.append(cur)
for k, run in enumerate(runs):
    tail = run[-(len(run) // 3):]                      # the timed steps at the end of the run (the reset recipe comes first)
    busy = sum(e - s for s, e, _ in tail)
    ivs, union, end = sorted((s, e) for s, e, _ in tail), 0, 0
    for s, e in ivs:
        if s > end:
            union += e - s; end = e
        elif e > end:
            union += e - end; end = e
    print("configuration %d: %d launches in the tail, mean %.2f ms, busy %.1f ms over %.1f ms of wall = concurrency %.2f" % (
        k, len(tail), busy / len(tail) / 1e6, busy / 1e6, union / 1e6, busy / max(union, 1)))
