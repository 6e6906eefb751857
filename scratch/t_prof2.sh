#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash scratch/prof_model.sh resnet152 128 320 > /dev/null 2>&1
bash scratch/prof_model.sh aadensenet121 128 320 > /dev/null 2>&1
for m in resnet152 aadensenet121; do
timeout -k 10 300 python bench.py --model $m --batch 128 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/prof_$m/bench_graph.json 2>> gpurun_out/prof_$m/bench.err
python -c "
import json; d=json.load(open('gpurun_out/prof_$m/bench_graph.json')); print('$m', d['value'], d['ms_per_step'], d['roofline']['kernel'])"
done
head -24 gpurun_out/prof_resnet152/kstats.txt | cut -c1-150
