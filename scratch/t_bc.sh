#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bc; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_golden_smooth_gpu.py -q -x -s -k "densenetbc or densenet_bc" > $O/tests.log 2>&1; rc=$?; echo "bc tests rc=$rc"; grep -h "rel\|passed\|failed\|Error" $O/tests.log | tail -12
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { tail -40 $O/tests.log; exit $rc; }
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_aaconv_gpu.py -q -x -k "cifar_harness or aadensenet or transition" > $O/tests2.log 2>&1; rc=$?; echo "regression tests rc=$rc"; tail -4 $O/tests2.log
if grep -q "Memory access fault" $O/tests2.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { grep -n "Error\|^E " $O/tests2.log | head -20; }
exit $rc
