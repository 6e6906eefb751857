#!/bin/bash
# round 2, call 1: new tests + A/B of the pipelined 1x1 backward and of graph replay
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" | tee $O/tests.rc
tail -5 $O/tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-graph > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc=$?"
CX_PW_BWD_V1=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-graph > $O/bench_v1.json 2> $O/bench_v1.err; echo "bench v1 rc=$?"
cat $O/bench_graph.json $O/bench_eager.json $O/bench_v1.json
grep -h "pw_bwd\|timed region\|eager replica\|graph" $O/bench_graph.err $O/bench_eager.err $O/bench_v1.err
