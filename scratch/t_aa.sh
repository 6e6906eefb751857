#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/aa; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_determinism_gpu.py tests/test_aaconv_gpu.py tests/test_resnet_gpu.py tests/test_golden_smooth_gpu.py -q -s -k "aa or AA" > $O/tests.log 2>&1; echo "tests rc=$?"
grep -h "passed\|failed\|FAILED\|Error" $O/tests.log | cut -c1-300 | tail -20
grep -h "x1:\|x16:" $O/tests.log | cut -c1-250
timeout -k 10 300 python bench.py --model aadensenet121 --batch 128 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_aa.json 2> $O/bench_aa.err; python -c "
import json; d=json.load(open('$O/bench_aa.json')); print('aadensenet121', d['value'], d['ms_per_step'])"
