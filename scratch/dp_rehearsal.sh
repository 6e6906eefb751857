#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/seg; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -q -x -s > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; grep -h "rank .:\|passed\|failed\|Error" $O/tests.log | tail -20
if grep -q "Memory access fault" $O/tests.log; then echo "GPU FAULT"; exit 3; fi
[ $rc -ne 0 ] && { tail -30 $O/tests.log; exit $rc; }
for G in 0 1; do
CHEXPERT_BENCH_GRAPH=$G CHEXPERT_BENCH_FORCE_DP=1 CHEXPERT_FORCE_COLLECTIVES=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_dp1_$G.json 2> $O/bench_dp1_$G.err; tail -1 $O/bench_dp1_$G.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one-rank DP rehearsal graph=$G', d['value'], d['ms_per_step'], d['config']['launch'])"
grep -h "segments\|failed\|probe\|Error" $O/bench_dp1_$G.err | head -6
done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('densenet121 N=1 graph', d['value'], d['ms_per_step'])"
for M in "resnet152 128 320" "efficientnet-b4 64 380" "aadensenet121 128 320"; do set -- $M
CHEXPERT_BENCH_FORCE_DP=1 CHEXPERT_FORCE_COLLECTIVES=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29618 bench.py --model $1 --batch $2 --size $3 --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/b_dp_$1.json 2> $O/b_dp_$1.err; tail -1 $O/b_dp_$1.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1 one-rank DP', d['value'], d['ms_per_step'], d['config']['launch'])"
grep -h "probe\|failed" $O/b_dp_$1.err | cut -c1-200
done
