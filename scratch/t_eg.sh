#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/eg; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_efficientnet_gpu.py tests/test_determinism_gpu.py tests/test_model_gpu.py -q -x -k "efficientnet or Efficient or graph" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && exit $rc
for g in "" "--no-graph"; do
timeout -k 10 300 python bench.py --model efficientnet-b4 --batch 64 --size 380 --no-cpu-baseline --steps 20 --warmup 5 $g > $O/bench.json 2> $O/bench.err; grep -i "graph" $O/bench.err | tail -2; python -c "
import json; d=json.load(open('$O/bench.json')); print('efficientnet-b4 $g', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['loss'])"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-graph > $O/bench_dn_eager.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_dn_eager.json')); print('densenet121 eager', d['value'], d['ms_per_step'], d['config']['launch'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_dn.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_dn.json')); print('densenet121 graph', d['value'], d['ms_per_step'], d['config']['launch'])"
