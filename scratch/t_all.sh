#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/all; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -h "passed\|failed\|FAILED" $O/tests.log | cut -c1-300 | tail -12
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
timeout -k 10 300 python scratch/host_rate.py > $O/host_enqueue.txt 2>> $O/bench.err; cat $O/host_enqueue.txt
exit $rc
