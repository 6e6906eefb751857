#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f32a; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_fp32_gpu.py tests/test_aaconv_gpu.py tests/test_determinism_gpu.py -q -s -x -k "attention or aa or AA" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -h "passed\|failed\|FAILED\|Error\|fp32 aa" $O/tests.log | cut -c1-260 | tail -12
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
exit $rc
