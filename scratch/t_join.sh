#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/join; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_conv_mm_gpu.py -q -x -k "join" > $O/tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -5 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_resnet_gpu.py tests/test_determinism_gpu.py -q -x -s -k "resnet or join" > $O/tests2.log 2>&1; rc=$?; echo "model tests rc=$rc"; grep -h "join fuse\|passed\|failed" $O/tests2.log | tail -6
if grep -q "Memory access fault" $O/tests2.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { tail -30 $O/tests2.log; exit $rc; }
for f in 1 0 1 0; do
CHEXPERT_JOIN_FUSE=$f timeout -k 10 300 python bench.py --model resnet152 --batch 128 --size 320 --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_fuse$f.json 2>> $O/bench.err || exit 4
python -c "
import json; d=json.load(open('$O/bench_fuse$f.json')); print('fuse=$f', d['value'], d['ms_per_step'])"
done
