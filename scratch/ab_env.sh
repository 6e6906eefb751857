#!/bin/bash
# A/B of one environment variable on one box: bash scratch/ab_env.sh VAR=valueA VAR=valueB [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for r in 1 2; do
  for E in $A $B; do
    env $E timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; }
    python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$E', d['value'], d['ms_per_step'])"
  done
done
