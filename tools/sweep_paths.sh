#!/bin/bash
# both elimination paths on the round-0 Schur complement of a family of generated matrices (tools/probe_paths.py):
# the data behind the path rule of DESIGN.md section 3
cd "$(dirname "$0")/.." || exit 1
for w in mk13.b3 ch7-8.b3 ch8-8.b3 mk12.b3 mk13.b5 ch7-7.b4 ch7-8.b4 mk12.b4 ch8-8.b4 mk13.b4 mk12.b5 ch7-7.b3 ch6-7.b4 mk11.b4; do
	timeout 300 python tools/probe_paths.py --workload $w --steps 3 --no-e2e 2>&1 | grep -v amdgpu
done
