#!/bin/bash
# one gpurun call that refreshes every committed measurement: tests, bench line, kernel stats, PMC traffic
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --no-cpu-baseline > $O/bench_prof.json 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>> $O/bench.err
python scratch/pmc_summary.py $(ls $O/fetch/*/*counter_collection.csv) $(ls $O/write/*/*counter_collection.csv) $O/pmc_traffic.json 4 > $O/pmc.txt
cp $(ls $O/stats/*/*_kernel_stats.csv) $O/kernel_stats.csv
rm -rf $O/stats/*/*kernel_trace.csv $O/fetch $O/write
tail -2 $O/tests.log; cat $O/bench.json
