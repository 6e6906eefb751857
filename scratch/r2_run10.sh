#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2j; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log
