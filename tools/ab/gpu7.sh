mkdir -p gpurun_out/pmc_pivots
export SPASM_HIP_EXPERIMENT=1
export TMPDIR=/tmp
for w in mk15.b4 mk14.b4; do
wp=8; [ $w = mk14.b4 ] && wp=4
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_pivots/$w -- python3 tools/probe_pivot_waves.py $w $wp > gpurun_out/pmc_pivots/$w.log 2>&1
python3 - $w <<'PY'
import csv, glob, sys, collections
w = sys.argv[1]
acc = collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_pivots/%s/**/*counter_collection.csv" % w, recursive=True):
    for r in csv.DictReader(open(f)):
        if "pivot_search_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(w, dict(acc))
PY
grep -E "device:" gpurun_out/pmc_pivots/$w.log | cut -c1-300
done
