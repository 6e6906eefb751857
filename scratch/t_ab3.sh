#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_determinism_gpu.py tests/test_model_gpu.py tests/test_golden_smooth_gpu.py tests/test_resnet_gpu.py -q -x -k "not cifar" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { grep -n "Error\|^E \|FAILED" $O/tests.log | head; exit $rc; }
cp chexpert_amd/libchexpert_hip.so /tmp/cur.so
for L in scratch/libA.so scratch/libB.so scratch/libA.so scratch/libB.so; do
  cp $L chexpert_amd/libchexpert_hip.so
  echo "== $L"; timeout -k 10 300 python scratch/bench_ring.py 2>&1 | grep -v amdgpu
done
bash scratch/ab.sh scratch/libA.so scratch/libB.so
cp /tmp/cur.so chexpert_amd/libchexpert_hip.so
