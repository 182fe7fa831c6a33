# FETCH_SIZE / WRITE_SIZE calibration by access pattern (tools/ubench/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
./tools/ubench/fetch_calib > gpurun_out/fetch_calib_bytes.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/fc_fetch -o f --output-format csv -- ./tools/ubench/fetch_calib > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/fc_write -o w --output-format csv -- ./tools/ubench/fetch_calib > /dev/null 2>&1
python - <<PY
import csv, glob
known = {}
for l in open("gpurun_out/fetch_calib_bytes.txt"):
    w = l.split()
    if len(w) == 3 and w[1].isdigit(): known[w[0]] = (int(w[1]), int(w[2]))
out = ["FETCH_SIZE / WRITE_SIZE calibration by access pattern (tools/ubench/fetch_calib.hip, MI355X): counter in KB as rocprofv3 reports it; factor = known bytes / (KB x 1024)", ""]
for tag, ctr in (("fc_fetch", "FETCH_SIZE"), ("fc_write", "WRITE_SIZE")):
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if r["Counter_Name"] == ctr and k in known and k.startswith("read" if ctr == "FETCH_SIZE" else "write"):
                kb = float(r["Counter_Value"]); req, lines = known[k]
                out.append("%-10s %-18s counter %12.0f KB | requested %12d B  factor %.3f | 64-B lines touched %12d B  factor %.3f" % (ctr, k, kb, req, req / max(kb * 1024, 1), lines, lines / max(kb * 1024, 1)))
open("gpurun_out/fetch_calib.txt", "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
