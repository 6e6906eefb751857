#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/quart; rm -rf $O; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_aaconv_gpu.py tests/test_kernels_gpu.py -q -x -k "aa_densenet_matches_oracle or conv3x3 or ring or strip or dense_side or wgrad" > $O/small.log 2>&1; rc=$?; echo "small rc=$rc"; tail -3 $O/small.log
[ $rc -ne 0 ] && { grep -n "^E \|FAILED" $O/small.log | head; exit $rc; }
for Q in 0 1 0 1; do echo "CX_RING_DGRAD_QUART=$Q"
  CX_RING_DGRAD_QUART=$Q timeout -k 10 120 python scratch/bench_ring.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/bench.log || exit 1
done
for r in 1 2 3; do for Q in 0 1; do
  CX_RING_DGRAD_QUART=$Q timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('densenet121 QUART=$Q', d['value'], d['ms_per_step'])" | tee -a $O/ab.log
done; done
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
timeout -k 10 200 python scratch/stamps_ring.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
