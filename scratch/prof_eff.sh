#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/peff; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_dwconv_gpu.py tests/test_efficientnet_gpu.py tests/test_determinism_gpu.py tests/test_golden_smooth_gpu.py tests/test_dp_gpu.py -q -s -k "efficientnet or se_backward or dwconv" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -h "passed\|failed\|FAILED\|efficientnet-b. .* x\|rank 0" $O/tests.log | cut -c1-230 | tail -10
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
for det in 1; do
CHEXPERT_DET=$det timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$det -- python bench.py --model efficientnet-b4 --batch 64 --size 380 --no-cpu-baseline --steps 6 --warmup 2 > $O/bench$det.json 2> $O/bench$det.err
cp $(ls $O/st$det/*/*_kernel_stats.csv) $O/kernel_stats_det$det.csv; rm -rf $O/st$det
python -c "
import json; d=json.load(open('$O/bench$det.json')); print('det=$det efficientnet-b4', d['value'], d['ms_per_step'])"
done
exit $rc
