#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b; rm -rf $O; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py tests/test_model_gpu.py -m gpu -q -x > $O/t.log 2>&1; echo "tests rc=$?"; tail -2 $O/t.log
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench$i.json 2> $O/bench$i.err; python -c "
import json; d=json.load(open('$O/bench$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done
grep "launches" $O/bench1.err | head -12
