#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tl; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/b.json 2> $O/b.err
python scratch/timeline.py $(ls $O/tr/*/*kernel_trace.csv)
rm -rf $O/tr
