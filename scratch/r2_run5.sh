#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2e; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_determinism_gpu.py -m gpu -q -s > $O/det.log 2>&1; echo "det rc=$?"; tail -15 $O/det.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s --deselect tests/test_determinism_gpu.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json; grep "launches" $O/bench.err | head -20
CHEXPERT_DET=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_legacy.json 2> $O/bench_legacy.err; echo "bench legacy rc=$?"; python -c "
import json; d=json.load(open('$O/bench_legacy.json')); print('legacy', d['value'], d['ms_per_step'])"
