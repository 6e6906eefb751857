#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2g; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$O/bench.json')); print('det', d['value'], d['ms_per_step'])"
CHEXPERT_DET=0 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_legacy.json 2> $O/bench_legacy.err; python -c "
import json; d=json.load(open('$O/bench_legacy.json')); print('legacy', d['value'], d['ms_per_step'])"
