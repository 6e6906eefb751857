#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/aa8; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_aaconv_gpu.py tests/test_resnet_gpu.py -q -s -k "attention_forward_backward or head_size_8 or aa" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -h "passed\|failed\|FAILED\|Error\|aawrn10_10\|cos " $O/tests.log | cut -c1-200 | tail -20
