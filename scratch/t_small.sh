#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/small; rm -rf $O; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_aaconv_gpu.py tests/test_kernels_gpu.py -q -x -v -k "aa_densenet_matches_oracle or conv3x3 or ring or strip or dense_side" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -15 $O/tests.log | cut -c1-200
exit $rc
