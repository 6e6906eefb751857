#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dy; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_golden_smooth_gpu.py tests/test_determinism_gpu.py tests/test_dp_gpu.py tests/test_aaconv_gpu.py tests/test_fp32_gpu.py -q -x -k "not resnet and not efficientnet" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
for r in 1 2; do for E in "CHEXPERT_DENSE_DY=0 CHEXPERT_SERIAL_WGRAD=0" "CHEXPERT_DENSE_DY=1 CHEXPERT_SERIAL_WGRAD=1"; do
  env $E timeout -k 10 300 python bench.py --model aadensenet121 --batch 128 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('aadensenet121 $E', d['value'], d['ms_per_step'])"
done; done
