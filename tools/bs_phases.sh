for v in 12 13; do
  for dbg in 0 1 2 4 3 7; do
    echo -n "variant $v dbg $dbg: "
    SPASM_HIP_EXPERIMENT=1 SPASM_HIP_BS_DEBUG=$dbg timeout 200 python tools/probe_backsolve.py --variants $v --steps 5 --no-check 2>&1 | grep variant | sed 's/.*backsolve \([0-9.]*\) ms.*/\1 ms/'
  done
done
