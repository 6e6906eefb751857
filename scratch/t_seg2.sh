#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/seg2; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -q -x -k "segmented or cli" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && { tail -40 $O/tests.log; exit $rc; }
CHEXPERT_BENCH_FORCE_DP=1 CHEXPERT_FORCE_COLLECTIVES=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 1 --no-cpu-baseline > $O/bench_dp1.json 2> $O/bench_dp1.err; echo "rc=$?"; tail -1 $O/bench_dp1.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('one-rank DP rehearsal (default steps)', d['value'], d['ms_per_step'], d['config']['launch'])"
grep -h "segments\|failed\|probe\|Error" $O/bench_dp1.err | head -6
