#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/aak; rm -rf $O; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_aaconv_gpu.py -q -x -k "attention_forward_backward" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
echo "== mfma key side"; timeout -k 10 120 python scratch/bench_aa.py 128 2>&1 | grep -v amdgpu.ids
echo "== row key side"; CX_AA_K_ROW=1 timeout -k 10 120 python scratch/bench_aa.py 128 2>&1 | grep -v amdgpu.ids
