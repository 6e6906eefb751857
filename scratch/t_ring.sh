#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ring; rm -rf $O; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_aaconv_gpu.py tests/test_kernels_gpu.py -q -x -k "aa_densenet_matches_oracle or conv3x3 or ring or strip or dense_side or wgrad" > $O/small.log 2>&1; rc=$?; echo "small rc=$rc"; tail -3 $O/small.log
[ $rc -ne 0 ] && exit $rc
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
for L in scratch/libA.so scratch/libB.so scratch/libA.so scratch/libB.so; do
  cp $L chexpert_amd/libchexpert_hip.so; echo $L
  timeout -k 10 120 python scratch/bench_ring.py 2>&1 | grep -v "amdgpu.ids" | tee -a $O/bench.log || exit 1
done
timeout -k 10 200 python scratch/stamps_ring.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
