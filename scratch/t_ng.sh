#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ng; rm -rf $O; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_aaconv_gpu.py tests/test_kernels_gpu.py -q -x -k "aa_densenet_matches_oracle or conv3x3 or ring or strip or dense_side or wgrad" > $O/small.log 2>&1; rc=$?; echo "small rc=$rc"; tail -3 $O/small.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do timeout -k 10 120 python scratch/bench_w2batch.py 2>&1 | grep -v "^CX_SW\|amdgpu.ids" | tee -a $O/ng.log || exit 1; done
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
timeout -k 10 200 python scratch/stamps_strip.py 2>&1 | grep -v amdgpu.ids | tee $O/strip.txt
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
