#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/f32; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fp32_gpu.py tests/test_resnet_gpu.py -q -s -k "fp32_resnet or resnet_smooth or resnet152_matches" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -h "passed\|failed\|FAILED\|Error\|fp32 res" $O/tests.log | cut -c1-260 | tail -20
timeout -k 10 300 python bench.py --model resnet152 --dtype fp32 --batch 32 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
