#!/bin/bash
# per-kernel times of the attention micro-benchmark: bash scratch/prof_aa.sh <batch> <out-name>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$2; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python scratch/bench_aa.py $1 > $O/out.txt 2> $O/err.txt
python scratch/kstats.py $O/stats 1 20 > gpurun_out/kstats_$2.txt; rm -rf $O/stats
cat $O/out.txt gpurun_out/kstats_$2.txt
