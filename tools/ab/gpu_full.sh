mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r04_gputests_full2.log 2>&1
tail -5 gpurun_out/r04_gputests_full2.log
