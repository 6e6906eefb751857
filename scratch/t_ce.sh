#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ce; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "softmax or harness or cifar" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log | cut -c1-220
exit $rc
