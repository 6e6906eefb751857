#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/abc; rm -rf $O; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_conv_mm_gpu.py tests/test_resnet_gpu.py -q -x > $O/small.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/small.log
[ $rc -ne 0 ] && { grep -n "^E \|FAILED" $O/small.log | head; exit $rc; }
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
for L in scratch/libA.so scratch/libC.so; do
  cp $L chexpert_amd/libchexpert_hip.so; echo $L
  timeout -k 10 120 python scratch/bench_ring.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/bench.log || exit 1
done
for r in 1 2; do for L in scratch/libA.so scratch/libB.so scratch/libC.so; do
  cp $L chexpert_amd/libchexpert_hip.so
  timeout -k 10 300 python bench.py --model resnet152 --batch 128 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('resnet152 $L', d['value'], d['ms_per_step'])" | tee -a $O/ab.log
done; done
for r in 1 2 3; do for L in scratch/libA.so scratch/libC.so; do
  cp $L chexpert_amd/libchexpert_hip.so
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('densenet121 $L', d['value'], d['ms_per_step'])" | tee -a $O/ab.log
done; done
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
