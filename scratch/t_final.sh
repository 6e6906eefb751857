#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { grep -h "FAILED\|Error" $O/tests.log | head -10; exit $rc; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
