mkdir -p gpurun_out/prof_real
export SPASM_HIP_EXPERIMENT=1
export TMPDIR=/tmp
python3 tools/probe_dense_real_only.py mk13.b5 > gpurun_out/prof_real/warm.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_real/trace -- python3 tools/probe_dense_real_only.py mk13.b5 > gpurun_out/prof_real/run.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
k = defaultdict(list)
for f in glob.glob("gpurun_out/prof_real/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sh::", "")[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(k.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print("%-60s %6d %10.1f %8.1f" % (n, len(v), sum(v), sum(v) / len(v)))
PY
tail -2 gpurun_out/prof_real/run.log | cut -c1-300
