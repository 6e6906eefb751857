#!/bin/bash
# small-map kernel tests first (loop changes can hang), then the A/B of two library builds
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_aaconv_gpu.py tests/test_kernels_gpu.py -q -x -k "aa_densenet_matches_oracle or conv3x3 or ring or strip or dense_side or wgrad" > $O/small.log 2>&1; rc=$?; echo "small rc=$rc"; tail -3 $O/small.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -m pytest tests/test_golden_smooth_gpu.py tests/test_model_gpu.py -q -x > $O/model.log 2>&1; rc=$?; echo "model rc=$rc"; tail -3 $O/model.log
[ $rc -ne 0 ] && exit $rc
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
bash scratch/ab.sh scratch/libA.so scratch/libB.so
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
