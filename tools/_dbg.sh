for d in 0 1 2 4 7; do echo "dbg $d"; SPASM_HIP_BS_DEBUG=$d timeout 120 python tools/probe_backsolve.py --variants 12 --no-check 2>&1 | grep variant; done
SPASM_HIP_BS_FAT=0 timeout 200 python tools/probe_backsolve.py --variants 12 2>&1 | grep "variant"
