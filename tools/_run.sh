ulimit -c 0
timeout 1500 python -m pytest tests/test_gpu_dense.py tests/test_gpu_echelonize.py -x -q 2>&1 | tail -2 >> gpurun_out/run.log
for sh in "4096 32768" "16384 20000"; do set -- $sh
python tools/bench_dense.py --n $1 --m $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['shape'], d['ms_total_untimed'], d['mfma_i8'], d['valu_u64']['ms_total'], d['same_matrix'])" >> gpurun_out/run.log
done
