#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2h; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fp32_gpu.py -m gpu -q -s -x > $O/fp32.log 2>&1; echo "fp32 rc=$?"; tail -25 $O/fp32.log
