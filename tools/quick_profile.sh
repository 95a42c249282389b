export TMPDIR=/tmp
mkdir -p gpurun_out/qp
rm -rf gpurun_out/qp/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qp/trace -- python3 tools/probe_backsolve.py --variants 12 --no-check --steps 5 > gpurun_out/qp/out.txt 2> gpurun_out/qp/err.txt
f=$(find gpurun_out/qp/trace -name "*kernel_stats.csv" | head -1)
head -6 "$f" | cut -d, -f1-4 | cut -c1-150
