#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/dn; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_determinism_gpu.py tests/test_model_gpu.py tests/test_golden_smooth_gpu.py -q -x -k "not resnet and not efficientnet" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
CHEXPERT_WGRAD_DEFER=0 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_nodefer.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_nodefer.json')); print('bench no-defer', d['value'], d['ms_per_step'])"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python bench.py --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>> $O/bench.err
f=$(ls $O/kt/*/*kernel_trace.csv | head -1)
python scratch/trace_step.py $f > $O/step_graph.txt
python scratch/timeline.py $f > $O/timeline.txt
rm -rf $O/kt
head -1 $O/step_graph.txt; head -8 $O/timeline.txt
