#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/seg2; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_efficientnet_gpu.py tests/test_determinism_gpu.py tests/test_dp_gpu.py -q -x -k "efficientnet or Efficient" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $O/tests.log
[ $rc -ne 0 ] && { tail -30 $O/tests.log; exit $rc; }
timeout -k 10 300 python bench.py --model efficientnet-b4 --batch 64 --size 380 --no-cpu-baseline --steps 20 --warmup 5 > $O/b.json 2> $O/b.err; echo "rc=$?"; grep -i "graph" $O/b.err | tail -2 | cut -c1-200; python -c "
import json; d=json.load(open('$O/b.json')); print('efficientnet-b4 N=1', d['value'], d['ms_per_step'], d['config']['launch'], d['config']['loss'])"
CHEXPERT_BENCH_FORCE_DP=1 CHEXPERT_FORCE_COLLECTIVES=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29618 bench.py --model efficientnet-b4 --batch 64 --size 380 --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/b_dp.json 2> $O/b_dp.err; echo "rc=$?"; tail -1 $O/b_dp.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('efficientnet-b4 one-rank DP', d['value'], d['ms_per_step'], d['config']['launch'])"
grep -h "probe\|failed" $O/b_dp.err | cut -c1-200
