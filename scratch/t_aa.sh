#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/aa; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_aaconv_gpu.py -q -x > $O/tests.log 2>&1; rc=$?; echo "aa tests rc=$rc"; tail -4 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { grep -n "Error\|^E " $O/tests.log | head -20; }
exit $rc
