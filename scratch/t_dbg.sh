#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dbg; rm -rf $O; mkdir -p $O
HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 python -m pytest "tests/test_efficientnet_gpu.py::test_efficientnet_matches_oracle" -q -s -x > $O/tests.log 2>&1; echo "tests rc=$?"
grep -n "Memory access\|File \"/tmp/code\|passed\|failed" $O/tests.log | head -12
