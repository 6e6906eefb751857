#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2d; rm -rf $O; mkdir -p $O
for d in 0 7 127; do
echo "== DBG=$d"; CX_PW_BWD_DBG=$d timeout -k 10 200 python scratch/bench_pw.py fused 2>&1 | tee $O/fused$d.txt
done
echo "== v1"; CX_PW_BWD_V1=1 timeout -k 10 200 python scratch/bench_pw.py fused 2>&1 | tee $O/fused_v1.txt
