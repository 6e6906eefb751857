#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f32e; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_efficientnet_gpu.py tests/test_dwconv_gpu.py tests/test_fp32_gpu.py -q -s -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -h "passed\|failed\|FAILED\|Error\|fp32 eff" $O/tests.log | cut -c1-260 | tail -12
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
exit $rc
