#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/eff; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dwconv_gpu.py tests/test_efficientnet_gpu.py tests/test_determinism_gpu.py tests/test_golden_smooth_gpu.py tests/test_fp32_gpu.py tests/test_dp_gpu.py -q -s -x -k "dwconv or se_backward or efficientnet or Efficient" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -h "passed\|failed\|FAILED\|Error\|efficientnet-b. .* x\|rank 0" $O/tests.log | cut -c1-260 | tail -14
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
timeout -k 10 300 python bench.py --model efficientnet-b4 --batch 64 --size 380 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('efficientnet-b4', d['value'], d['ms_per_step'], d['config']['launch'])"
exit $rc
