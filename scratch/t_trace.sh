#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > $O/bench.json 2> $O/bench.err
f=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python scratch/trace_step.py $f > $O/step_graph.txt
python scratch/timeline.py $f > $O/timeline.txt
CHEXPERT_SERIAL_WGRAD=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kts -- python bench.py --no-cpu-baseline --no-graph --steps 4 --warmup 2 > $O/bench_serial.json 2>> $O/bench.err
f=$(ls $O/kts/*/*kernel_trace.csv | head -1)
python scratch/trace_step.py $f > $O/step_serial.txt
rm -rf $O/kt $O/kts
head -3 $O/step_graph.txt; cat $O/timeline.txt | head -12; cat $O/bench.json | cut -c1-300
