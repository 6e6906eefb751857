#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dp_gpu.py -m gpu -x -q -s > $O/dp.log 2>&1; echo "dp rc=$?"; grep "rank\|passed\|failed\|mismatch" $O/dp.log
for d in 0 1 2 3 4; do
CX_PW_BWD_DBG=$d timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --no-graph --roofline-kernel "pw_bwd2_kernel<2, true>" > $O/dbg$d.json 2> $O/dbg$d.err; echo "dbg $d rc=$?"
python - <<PY
import json
d=json.load(open("$O/dbg$d.json")); print("DBG=$d", d["ms_per_step"], d["roofline"]["avg_launch_ms"])
PY
done
