#!/bin/bash
# kernel statistics of one model configuration: bash scratch/prof_model.sh resnet152 128 320
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$1; rm -rf $O; mkdir -p $O
CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --model $1 --batch $2 --size $3 --no-cpu-baseline --no-graph --steps 4 --warmup 1 > $O/bench.json 2> $O/bench.err
python scratch/kstats.py $O/stats 5 30 > $O/kstats.txt; cat $O/kstats.txt; rm -rf $O/stats
