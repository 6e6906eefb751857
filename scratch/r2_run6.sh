#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2f; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_det -- python bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 2 > $O/p_det.json 2> $O/p_det.err
CHEXPERT_DET=0 CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_leg -- python bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 2 > $O/p_leg.json 2> $O/p_leg.err
cp $(ls $O/stats_det/*/*_kernel_stats.csv) $O/kernel_stats_det.csv; cp $(ls $O/stats_leg/*/*_kernel_stats.csv) $O/kernel_stats_leg.csv
rm -rf $O/stats_det $O/stats_leg
head -30 $O/kernel_stats_det.csv
