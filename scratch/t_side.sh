#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/side; rm -rf $O; mkdir -p $O
for f in 0 1 0 1; do
CHEXPERT_W2_SIDE=$f timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_s$f.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 5; }
python -c "
import json; d=json.load(open('$O/bench_s$f.json')); print('w2 side=$f', d['value'], d['ms_per_step'], d['config']['loss'])"
done
CHEXPERT_W2_SIDE=1 timeout -k 10 600 python -m pytest tests/test_determinism_gpu.py tests/test_golden_smooth_gpu.py -q -x -k "densenet121 and not aa" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
