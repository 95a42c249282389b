#!/bin/bash
# round 6, final tree: the whole GPU suite, the randomised stress run, the bench line
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1
tail -3 gpurun_out/r06_gpu_tests.log
timeout 900 python tools/stress_gpu.py > gpurun_out/r06_stress.log 2>&1
tail -2 gpurun_out/r06_stress.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line.json 2> gpurun_out/r06_bench_stderr.txt
cp bench_full.json gpurun_out/r06_bench_full.json
wc -c gpurun_out/r06_bench_line.json; wc -l gpurun_out/r06_bench_line.json
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_bench_line.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','unit','ms_per_step','n_gpus','steps')}); print(d['roofline']); print(d['cpu_baseline'])
"
