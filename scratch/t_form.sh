#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/form
for r in 1 2; do for F in 0 3; do
  CX_MM_FORM=$F timeout -k 10 300 python bench.py --model resnet152 --batch 128 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('resnet152 CX_MM_FORM=$F', d['value'], d['ms_per_step'])" | tee -a gpurun_out/form/ab.log
done; done
