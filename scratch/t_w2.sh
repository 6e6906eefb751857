#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/w2; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad_batch" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
for sp in 8 16 32 64; do
CX_SW_BATCH_SPLITS=$sp timeout -k 10 300 python scratch/bench_w2batch.py > $O/bench_$sp.txt 2>&1 || { tail -5 $O/bench_$sp.txt; exit 4; }
cat $O/bench_$sp.txt | grep -v amdgpu.ids
done
for f in 1 0 1 0; do
CHEXPERT_W2_BATCH=$f timeout -k 10 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench_b$f.json 2>> $O/bench.err || { tail -5 $O/bench.err; exit 5; }
python -c "
import json; d=json.load(open('$O/bench_b$f.json')); print('batch=$f', d['value'], d['ms_per_step'])"
done
