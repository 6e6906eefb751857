#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tn; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_golden_smooth_gpu.py "tests/test_resnet_gpu.py::test_basic_block_networks_smooth_regime_match_fp32_oracle" -q -s > $O/tests.log 2>&1; echo "tests rc=$?"
grep -h "x1:\|x8:\|x16:\|x32:\|smooth:\|passed\|failed\|Error\|assert" $O/tests.log | cut -c1-400
