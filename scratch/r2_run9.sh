#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2i; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fp32_gpu.py -m gpu -q -s > $O/fp32.log 2>&1; echo "fp32 rc=$?"; tail -4 $O/fp32.log; grep "fp32 golden" $O/fp32.log
timeout -k 10 400 python bench.py --dtype fp32 --steps 3 --warmup 1 > $O/bench_fp32.json 2> $O/bench_fp32.err; echo "bench fp32 rc=$?"; cat $O/bench_fp32.json; grep "launches\|timed" $O/bench_fp32.err | head
