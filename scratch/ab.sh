#!/bin/bash
# A/B of two builds of the library on one box: bash scratch/ab.sh libA libB [bench args]   (alternating, 3 rounds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for r in 1 2 3; do
  for L in $A $B; do
    cp $L chexpert_amd/libchexpert_hip.so
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$L', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'])"
  done
done
