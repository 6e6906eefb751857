#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/t; rm -rf $O; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -s ${TESTSEL:-} > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
grep -h "rank 0\|aaresnet152 golden\|resnet152 golden" $O/tests.log | cut -c1-200
if [ -z "$TESTSEL" ]; then
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['mfma_util'])"
timeout -k 10 300 python bench.py --model resnet152 --batch 128 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_rn.json 2> $O/bench_rn.err; python -c "
import json; d=json.load(open('$O/bench_rn.json')); print('resnet152', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['traffic'])"
CHEXPERT_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 64 --no-cpu-baseline > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc=$?"; tail -1 $O/bench_n2.json | cut -c1-300
fi
