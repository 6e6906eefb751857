#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/t; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
