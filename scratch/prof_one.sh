#!/bin/bash
# per-kernel times of one model's eager step: bash scratch/prof_one.sh <model> <batch> <size> <out-name> [top]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
M=$1; B=$2; S=$3; N=$4; TOP=${5:-45}
O=gpurun_out/prof_$N; rm -rf $O; mkdir -p $O
CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --model $M --batch $B --size $S --no-cpu-baseline --no-other-configs --no-graph --steps 4 --warmup 1 > /dev/null 2> $O/err.txt
python scratch/kstats.py $O/stats 5 $TOP > gpurun_out/kstats_$N.txt; rm -rf $O/stats
cat gpurun_out/kstats_$N.txt
