ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -3 >> gpurun_out/run.log
python tools/bench_dense.py --n 4096 --m 32768 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['mfma_i8'], d['valu_u64'], d['same_matrix'])" >> gpurun_out/run.log
python tools/bench_dense.py --n 16384 --m 20000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['mfma_i8'], d['same_matrix'])" >> gpurun_out/run.log
python tools/bench_dense.py --n 1000 --m 100000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['mfma_i8'], d['same_matrix'])" >> gpurun_out/run.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dense2 -- python3 $GRAFT_REPO_ROOT/tools/bench_dense.py --n 4096 --m 32768 > /dev/null 2>&1
