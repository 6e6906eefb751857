#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/all; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$?"
grep -h "passed\|failed\|FAILED\|rank 0" $O/tests.log | cut -c1-300 | tail -30
