#!/bin/bash
# ResNet152: the negative control of the direction check, the fixture at both lo-plane policies, and what the lo plane on every identity join costs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5_resnet_margin.txt; : > $O
timeout -k 10 280 python -m pytest tests/test_golden_smooth_gpu.py -q -s -k "transposed_tile_in_resnet152 or (baseline_batch and resnet152)" 2>&1 | grep -E "x16:|corrupted|sign flip|passed|failed|Error|assert" | cut -c1-400 >> $O
for v in 6 1; do
  echo "## CHEXPERT_STREAM_LO_MIN=$v" >> $O
  CHEXPERT_STREAM_LO_MIN=$v timeout -k 10 280 python -m pytest tests/test_golden_smooth_gpu.py -q -s -k "baseline_batch and resnet152" 2>&1 | grep -E "x16: train|passed|failed" | cut -c1-200 >> $O
  for r in 1 2; do
    CHEXPERT_STREAM_LO_MIN=$v timeout -k 10 200 python bench.py --model resnet152 --batch 128 --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench resnet152 bs128: %.1f img/s %.2f ms/step' % (d['value'], d['ms_per_step']))" >> $O
  done
done
cat $O
