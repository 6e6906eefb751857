#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/seg2; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -q -x -k "cli" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && { tail -40 $O/tests.log; exit $rc; }
exit 0
