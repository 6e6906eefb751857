#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fwdk; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py -q -x -k "not aa and not efficientnet" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
for pipe in 0 2 3 4 0 3; do
  echo "== CX_FWDK_PIPE=$pipe"; CX_FWDK_PIPE=$pipe timeout -k 10 200 python scratch/bench_pw.py fwd 2>&1 | grep "hw=20\|hw=10"
done > $O/pw.txt 2>&1
cat $O/pw.txt
exit $rc
