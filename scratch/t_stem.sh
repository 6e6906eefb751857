#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/stem; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "stem or maxpool" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scratch/bench_stembwd.py > $O/bench_stembwd.txt 2>&1 && cat $O/bench_stembwd.txt || exit 4
CX_STEM_STRIP=0 timeout -k 10 300 python scratch/bench_stembwd.py > $O/bench_stembwd_nostrip.txt 2>&1 && cat $O/bench_stembwd_nostrip.txt || exit 4
timeout -k 10 600 python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"; cut -c1-700 $O/bench.json
exit $rc
